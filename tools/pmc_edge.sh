#!/usr/bin/env bash
# PMC passes for the edge kernels (each pass = its own run with --kernel-trace only; MI355X_MICROARCH.md rocprofv3 rules).
# usage (through gpurun): tools/pmc_edge.sh <tag>
set -uo pipefail
tag="${1:-pmc}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
mkdir -p $out
i=0
for set in "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_ANY SQ_INST_CYCLES_VMEM" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TA_TA_BUSY_sum TCC_BUSY_sum TCC_TAG_STALL_sum" "MemUnitStalled MemUnitBusy"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $out/${tag}_p$i -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/${tag}_p$i.log 2>&1
  python3 tools/rocpd_pmc.py $out/${tag}_p$i/p_results.db > $out/${tag}_p$i.txt 2>&1
  rm -rf $out/${tag}_p$i
done
head -16 $out/${tag}_p*.txt
