#!/usr/bin/env bash
# Kernel trace of the MD-loop path (one aspirin molecule, skin list): per-kernel durations AND the gaps between consecutive
# kernels of one step (tools/rocpd_timeline.py).   usage (through gpurun): tools/profile_md.sh <tag>
set -uo pipefail
tag="${1:-md}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/${tag}_mdtrace -o t -- python3 tools/md_profile.py > $out/${tag}_mdtrace.log 2>&1
{ echo "# cmd python3 tools/md_profile.py (60 calculator steps, one aspirin molecule)"; python3 tools/rocpd_stats.py $out/${tag}_mdtrace/t_results.db --by-grid; } > $out/${tag}_md_kernel_stats.txt
python3 tools/rocpd_timeline.py $out/${tag}_mdtrace/t_results.db embed_kernel 3 > $out/${tag}_md_timeline.txt 2>&1
rm -rf $out/${tag}_mdtrace
head -60 $out/${tag}_md_kernel_stats.txt
