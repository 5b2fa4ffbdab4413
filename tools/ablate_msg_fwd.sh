#!/usr/bin/env bash
# What msg_fwd_kernel waits for: builds that REMOVE one stream each (wrong results by design).
# Build here (CPU), run through gpurun: tools/ablate_msg_fwd.sh build | run
set -uo pipefail
variants=(NONE NOTAB NOMJ STORE TABLE)
if [ "${1:-run}" = build ]; then
  for v in "${variants[@]}"; do
    NNHIP_LIB_NAME=libabl_$v.so bash newtonnet_amd/csrc/build.sh -DEDGE_ABL_$v > /dev/null && echo built $v
  done
  NNHIP_LIB_NAME=libabl_NOTAB_NOMJ.so bash newtonnet_amd/csrc/build.sh -DEDGE_ABL_NOTAB -DEDGE_ABL_NOMJ > /dev/null && echo built NOTAB_NOMJ
  NNHIP_LIB_NAME=libabl_ALL.so bash newtonnet_amd/csrc/build.sh -DEDGE_ABL_NOTAB -DEDGE_ABL_NOMJ -DEDGE_ABL_STORE > /dev/null && echo built ALL
else
  for v in "${variants[@]}" NOTAB_NOMJ ALL; do
    NNHIP_ALLOW_TOOLING_LIB=1 NNHIP_LIB_NAME=libabl_$v.so python3 bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-train-leg --no-strong-leg --no-box-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); c=d['kernel_classes']
print('$v', d['ms_per_step'], {k: round(v['ms_per_step'],3) for k,v in c.items() if k.startswith('edge')})"
  done
fi
