#!/usr/bin/env bash
# TCC request / hit counters of the edge kernels (own pass, --kernel-trace only).  usage: tools/pmc_edge3.sh <tag>
set -uo pipefail
tag="${1:-r03}"
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
out=gpurun_out
rocprofv3 --kernel-trace --pmc TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum -d $out/${tag}_p3 -o p -- python3 bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-train-leg --no-strong-leg --no-box-leg > $out/${tag}_p3.log 2>&1
{ echo "# tree $(cat profiles/.tree_sha 2>/dev/null || echo unknown)"; python3 tools/rocpd_pmc.py $out/${tag}_p3/p_results.db; } > $out/${tag}_pmc_edge_p3.txt
rm -rf $out/${tag}_p3
head -16 $out/${tag}_pmc_edge_p3.txt
