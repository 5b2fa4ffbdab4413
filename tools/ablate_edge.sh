#!/usr/bin/env bash
# Build the EDGE_ABL_* variants of the library and time the edge-kernel classes with each (results are WRONG by design;
# this only attributes time to the gather / pair-row / table / store streams).  Run through gpurun from the repo root.
set -uo pipefail
for v in NONE SELF PAIR TABLE STORE; do
  NNHIP_LIB_NAME=libabl_$v.so bash newtonnet_amd/csrc/build.sh -DEDGE_ABL_$v > /dev/null
  NNHIP_ALLOW_TOOLING_LIB=1 NNHIP_LIB_NAME=libabl_$v.so python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-leg --no-strong-leg --no-box-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); c=d['kernel_classes'] if 'kernel_classes' in d else d.get('classes', {})
print('$v', d['ms_per_step'], {k: round(v['ms_per_step'],3) for k,v in c.items() if k.startswith('edge')})"
  rm -f newtonnet_amd/lib/libabl_$v.so
done
