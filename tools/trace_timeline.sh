cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats -d gpurun_out/q_trace -o t -- python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-leg --no-strong-leg --no-box-leg --no-train-roofline --warm-seconds 0.1 --regions 1 > gpurun_out/q_trace.log 2>&1
python3 tools/rocpd_timeline.py gpurun_out/q_trace/t_results.db "param_check_kernel<false>" 90 > gpurun_out/q_timeline.txt 2>&1
python3 tools/rocpd_stats.py gpurun_out/q_trace/t_results.db --by-grid > gpurun_out/q_kernel_stats.txt
rm -rf gpurun_out/q_trace
