#!/usr/bin/env bash
# MFMA-utilisation counters for the dense kernels (mlp128_kernel<*>, node_fwd/bwd_kernel): each set is its own run with
# --kernel-trace only (MI355X_MICROARCH.md, rocprofv3 rules).  usage (through gpurun): tools/pmc_mfma.sh <tag>
set -uo pipefail
tag="${1:-pmc_mfma}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
mkdir -p $out
i=0
for set in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES" "SQ_INSTS_VALU_MFMA_MOPS_F32 SQ_INSTS_MFMA SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU_MFMA_F32 SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "GRBM_GUI_ACTIVE SQ_WAVES SQ_INST_CYCLES_VMEM"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $out/${tag}_m$i -o p -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline > $out/${tag}_m$i.log 2>&1
  python3 tools/rocpd_pmc.py $out/${tag}_m$i/p_results.db > $out/${tag}_m$i.txt 2>&1
  rm -rf $out/${tag}_m$i
done
head -14 $out/${tag}_m*.txt
