#!/usr/bin/env bash
# Vector-memory-path counters of force_fwd / msg_bwd in their molecule-resident and row forms (own runs with --kernel-trace only).
# usage (through gpurun): tools/pmc_mol_vs_row.sh <tag>
set -uo pipefail
tag="${1:-molrow}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
mkdir -p $out
rm -f $out/${tag}.txt
for form in mol row; do
  if [ $form = row ]; then export NNHIP_FORCE_FWD_MOL=0 NNHIP_MSG_BWD_MOL=0; else unset NNHIP_FORCE_FWD_MOL NNHIP_MSG_BWD_MOL; fi
  i=0
  for set in "TA_TA_BUSY_sum TCC_BUSY_sum GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "MemUnitBusy MemUnitStalled" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VMEM"; do
    i=$((i+1))
    rocprofv3 --kernel-trace --pmc $set -d $out/${tag}_${form}_p$i -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train-leg --no-strong-leg --no-box-leg --warm-seconds 0.1 --regions 1 > $out/${tag}_${form}_p$i.log 2>&1
    { echo "## $form form: $set"; python3 tools/rocpd_pmc.py $out/${tag}_${form}_p$i/p_results.db | grep -i "kernel \|force_fwd\|msg_bwd"; } >> $out/${tag}.txt 2>&1
    rm -rf $out/${tag}_${form}_p$i
  done
done
cat $out/${tag}.txt
