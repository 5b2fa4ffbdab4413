# compile the probe with 0..15 dwords of padding ahead of the kernel body (every alignment of its code mod 64 bytes) and run each
# CHAIN=6 bash tools/probes/run_pad_sweep.sh: the chain without op_sel modifiers
for k in 0 1 2 3 4 5 6 7 8 9 10 11 12 13 14 15; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -DPAD=$k -DCHAIN=${CHAIN:-0} ${PROBE_FLAGS:-} tools/probes/pk_chain_probe.hip -o /tmp/pk_pad_$k 2>/dev/null && timeout 120 /tmp/pk_pad_$k ${1:-20000} 2>&1 | grep "^pad" | grep "2 workgroup"
done
