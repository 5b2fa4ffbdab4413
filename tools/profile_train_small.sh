#!/usr/bin/env bash
# Kernel trace of the small-batch all-HIP training step (ethanol x 32, HIP-graph replay): per-kernel durations + one step in
# dispatch order.   usage (through gpurun): tools/profile_train_small.sh <tag>
set -uo pipefail
tag="${1:-tr}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/${tag}_trs -o t -- python3 tools/profile_train.py ethanol 32 --fused --steps 30 > $out/${tag}_trs.log 2>&1
{ echo "# cmd python3 tools/profile_train.py ethanol 32 --fused --steps 30"; python3 tools/rocpd_stats.py $out/${tag}_trs/t_results.db --by-grid; } > $out/${tag}_train_ethanol32_kernel_stats.txt
for b in 3 8 15; do python3 tools/rocpd_timeline.py $out/${tag}_trs/t_results.db embed_rows_kernel $b > $out/${tag}_train_ethanol32_timeline_$b.txt 2>&1; done
rm -rf $out/${tag}_trs
