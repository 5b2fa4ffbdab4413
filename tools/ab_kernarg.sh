for v in 0 1; do
  echo "== HIP_FORCE_DEV_KERNARG=$v" >> gpurun_out/kernarg.txt
  HIP_FORCE_DEV_KERNARG=$v NNHIP_ALLOW_TOOLING_LIB=1 NNHIP_LIB_NAME=libnewtonnet_hip_nsclk.so python tools/dbg_node_clock.py 2>&1 | grep node_fwd | tail -2 >> gpurun_out/kernarg.txt
  HIP_FORCE_DEV_KERNARG=$v python tools/bench_latency.py 2>&1 | grep -v amdgpu.ids >> gpurun_out/kernarg.txt
done
