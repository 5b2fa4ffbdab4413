#!/usr/bin/env bash
# Kernel timeline of one steady-state step of B aspirin conformers (default forms): tools/trace_small.sh B [tag]   (through gpurun)
B="${1:-128}"; tag="${2:-small}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
rocprofv3 --kernel-trace --stats -d gpurun_out/tr_$tag -o t -- python3 tools/steps.py $B 30 > gpurun_out/tr_$tag.log 2>&1
python3 tools/rocpd_timeline.py gpurun_out/tr_$tag/t_results.db param_check_kernel 3 > gpurun_out/r05/timeline_${tag}_B$B.txt
python3 tools/rocpd_stats.py gpurun_out/tr_$tag/t_results.db > gpurun_out/r05/kstats_${tag}_B$B.txt
rm -rf gpurun_out/tr_$tag
tail -45 gpurun_out/r05/timeline_${tag}_B$B.txt
