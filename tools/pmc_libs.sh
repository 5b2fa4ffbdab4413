#!/usr/bin/env bash
# FETCH_SIZE (fabric read traffic) and kernel times of the edge kernels for several builds of the library, same box.
# usage (through gpurun): tools/pmc_libs.sh <tag> libA.so libB.so ...
set -uo pipefail
tag="$1"; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
mkdir -p $out $(dirname $out/$tag)
for v in "$@"; do
  NNHIP_ALLOW_TOOLING_LIB=1 NNHIP_LIB_NAME=$v rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $out/${tag}_$v -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train-leg --no-strong-leg --no-box-leg --warm-seconds 0.1 --regions 1 > $out/${tag}_$v.log 2>&1
  { echo "# $v  FETCH_SIZE (KiB; x2 on gfx950 for wide reads)"; python3 tools/rocpd_pmc.py $out/${tag}_$v/p_results.db | grep -E "kernel|msg_fwd|msg_bwd|force_fwd|force_bwd"; } > $out/${tag}_$v.txt 2>&1
  rm -rf $out/${tag}_$v
done
cat $out/${tag}_*.txt
