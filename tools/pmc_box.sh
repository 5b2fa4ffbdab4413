#!/usr/bin/env bash
# FETCH_SIZE / WRITE_SIZE of every kernel of the 100k-atom box step (BASELINE configs[4]).  usage (through gpurun): tools/pmc_box.sh <tag>
set -uo pipefail
tag="${1:-box}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
mkdir -p $out $(dirname $out/$tag)
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c -d $out/${tag}_$c -o p -- python3 bench.py --workload box100k --steps 2 --warmup 1 --no-cpu-baseline --no-train-leg --no-strong-leg --no-box-leg --warm-seconds 0.1 --regions 1 > $out/${tag}_$c.log 2>&1
  { echo "# box100k  $c (KiB; FETCH_SIZE x2 on gfx950)"; python3 tools/rocpd_pmc.py $out/${tag}_$c/p_results.db | head -16; } > $out/${tag}_$c.txt 2>&1
  rm -rf $out/${tag}_$c
done
cat $out/${tag}_FETCH_SIZE.txt $out/${tag}_WRITE_SIZE.txt
