#!/usr/bin/env bash
# Collect the rocprofv3 evidence for one round on the GPU box (run through gpurun):
#   1. kernel trace + stats of the default bench command
#   2. PMC pass: FETCH_SIZE (3 TCC slots)        3. PMC pass: WRITE_SIZE (2 TCC slots)
# Counters are collected in their own runs with --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 PMC slots).
# usage: tools/profile_round.sh <tag>      -> gpurun_out/<tag>_{trace,fetch,write}/ + text summaries
set -uo pipefail
tag="${1:-r01}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
mkdir -p $out
rocprofv3 --kernel-trace --stats -d $out/${tag}_trace -o t -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-leg --no-strong-leg --no-box-leg --streams 1 --warm-seconds 0.1 --regions 1 > $out/${tag}_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/${tag}_fetch -o f -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train-leg --no-strong-leg --no-box-leg --streams 1 --warm-seconds 0.1 --regions 1 > $out/${tag}_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $out/${tag}_write -o w -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train-leg --no-strong-leg --no-box-leg --streams 1 --warm-seconds 0.1 --regions 1 > $out/${tag}_write.log 2>&1
# the same step with the bench's default steps in flight (two lanes on two streams): how much of the step's ~45 launches overlap
rocprofv3 --kernel-trace --stats -d $out/${tag}_trace2 -o t -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-leg --no-strong-leg --no-box-leg --warm-seconds 0.1 --regions 1 > $out/${tag}_trace2.log 2>&1
# the tree these counters were collected on (written by the caller before the snapshot travels: the GPU box has no .git)
sha="$(cat profiles/.tree_sha 2>/dev/null || echo unknown)"
{ echo "# tree $sha"; echo "# cmd python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-leg --no-strong-leg --no-box-leg --streams 1 (every row belongs to the inference step; one step at a time on the GPU)"; python3 tools/rocpd_stats.py $out/${tag}_trace/t_results.db --by-grid; } > $out/${tag}_kernel_stats.txt
{ echo "# tree $sha"; echo "# cmd bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train-leg --no-strong-leg --no-box-leg --streams 1"; python3 tools/rocpd_pmc.py $out/${tag}_fetch/f_results.db; } > $out/${tag}_pmc_fetch_size.txt
{ echo "# tree $sha"; echo "# cmd bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train-leg --no-strong-leg --no-box-leg --streams 1"; python3 tools/rocpd_pmc.py $out/${tag}_write/w_results.db; } > $out/${tag}_pmc_write_size.txt
{ echo "# tree $sha"; echo "# cmd python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-leg --no-strong-leg --no-box-leg (default --streams: two steps in flight on two lanes; a kernel's duration here includes what it shares the chip with)"; python3 tools/rocpd_stats.py $out/${tag}_trace2/t_results.db --by-grid; } > $out/${tag}_kernel_stats_2_in_flight.txt
python3 -c "import bench; print(bench.csrc_digest())" > $out/${tag}_csrc_sha.txt   # -> profiles/.csrc_sha when these summaries are committed
rm -rf $out/${tag}_trace $out/${tag}_trace2 $out/${tag}_fetch $out/${tag}_write
head -25 $out/${tag}_kernel_stats.txt; head -12 $out/${tag}_pmc_fetch_size.txt; head -12 $out/${tag}_pmc_write_size.txt
