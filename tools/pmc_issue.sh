#!/usr/bin/env bash
# Where the waves of each kernel spend their time: parked at a wait (SQ_WAIT_ANY), stalled at issue (SQ_WAIT_INST_ANY) or
# issuing (SQ_ACTIVE_INST_*), in quad-cycles; + instruction counts.  Two passes, --kernel-trace only.
# usage (through gpurun): tools/pmc_issue.sh <tag> [lib.so]
set -uo pipefail
tag="${1:-issue}"
lib="${2:-}"
[ -n "$lib" ] && export NNHIP_LIB_NAME="$lib" NNHIP_ALLOW_TOOLING_LIB=1
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_VMEM SQ_BUSY_CYCLES" \
           "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES SQ_INST_CYCLES_VMEM SQ_ACTIVE_INST_MISC"; do
  i=$((i+1))
  rocprofv3 --kernel-trace --pmc $set -d $out/${tag}_i$i -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train-leg --no-strong-leg --no-box-leg > $out/${tag}_i$i.log 2>&1
  python3 tools/rocpd_pmc.py $out/${tag}_i$i/p_results.db > $out/${tag}_pmc_issue_p$i.txt 2>&1
  rm -rf $out/${tag}_i$i
done
head -14 $out/${tag}_pmc_issue_p*.txt
