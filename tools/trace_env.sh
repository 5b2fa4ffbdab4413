#!/usr/bin/env bash
# rocprofv3 kernel trace of the config-2 bench under each value of an environment switch: tools/trace_env.sh VAR a b ...
# -> gpurun_out/trace_<VAR>_<value>.txt   (through gpurun; rocprofv3 needs the program right after --, so the variable is exported)
var="$1"; shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for v in "$@"; do
  export $var=$v
  rocprofv3 --kernel-trace --stats -d gpurun_out/te_trace -o t -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-leg --no-strong-leg --no-box-leg --no-train-roofline > gpurun_out/te_trace.log 2>&1
  python3 tools/rocpd_stats.py gpurun_out/te_trace/t_results.db --by-grid > gpurun_out/trace_${var}_$v.txt
  rm -rf gpurun_out/te_trace
done
