#!/usr/bin/env bash
# Does the excess fabric traffic of the edge kernels (second reads of pair rows) come from the in-flight working set of an XCD
# exceeding its 4 MB L2?  Cap the occupancy of the four edge kernels with unused dynamic LDS (NNHIP_EDGE_LDS=<bytes>: 4-wave
# workgroups, so 160 KiB / bytes workgroups per CU) and read FETCH_SIZE + TCC hit / miss and the kernel times at each cap.
# usage (through gpurun): tools/pmc_occupancy.sh <tag>
set -uo pipefail
tag="${1:-occ}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
mkdir -p $out $(dirname $out/$tag)
for lds in 0 40960 65536; do
  for set in "FETCH_SIZE" "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
    name=$(echo $set | cut -d' ' -f1)
    NNHIP_EDGE_LDS=$lds rocprofv3 --kernel-trace --pmc $set -d $out/${tag}_${lds}_$name -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train-leg --warm-seconds 0.1 --regions 1 > $out/${tag}_${lds}_$name.log 2>&1
    { echo "# NNHIP_EDGE_LDS=$lds  counters: $set"; python3 tools/rocpd_pmc.py $out/${tag}_${lds}_$name/p_results.db | grep -E "kernel|msg_fwd|msg_bwd|force_fwd|force_bwd"; } > $out/${tag}_${lds}_$name.txt 2>&1
    rm -rf $out/${tag}_${lds}_$name
  done
done
cat $out/${tag}_*_FETCH_SIZE.txt $out/${tag}_*_TCC_HIT_sum.txt
