#!/usr/bin/env bash
# Kernel stats of the steady-state step of B aspirin conformers under NNHIP_MOL_FUSED=<mode>: tools/trace_mode.sh B mode   (through gpurun)
B="${1:-1024}"; mode="${2:-6}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/r05
export NNHIP_MOL_FUSED=$mode
rocprofv3 --kernel-trace --stats -d gpurun_out/tr_m$mode -o t -- python3 tools/steps.py $B 30 > gpurun_out/tr_m$mode.log 2>&1
python3 tools/rocpd_stats.py gpurun_out/tr_m$mode/t_results.db > gpurun_out/r05/kstats_mode${mode}_B$B.txt
rm -rf gpurun_out/tr_m$mode
head -16 gpurun_out/r05/kstats_mode${mode}_B$B.txt
