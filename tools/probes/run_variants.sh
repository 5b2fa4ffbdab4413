# pad sweeps of the probe's variants: which instruction of the chain has to carry op_sel, and whether the MFMA phase is needed
for v in "CHAIN=0" "CHAIN=0 PROBE_FLAGS=-DNO_MFMA" "CHAIN=20" "CHAIN=21" "CHAIN=22" "CHAIN=23"; do
  echo "== $v"
  env $v bash tools/probes/run_pad_sweep.sh ${1:-10000} | sed -E 's/^pad +([0-9]+) dwords.*chain differ ([0-9]+) .*evaluated twice differs ([0-9]+); the v_fma_f32 chain evaluated twice differs ([0-9]+)/pad \1: packed vs v_fma_f32 \2, packed twice \3, v_fma_f32 twice \4/' | tr '\n' ';'; echo
done
