#!/usr/bin/env bash
# A/B two builds of the library on the same box: tools/ab_lib.sh libA.so libB.so [repeats]   (run through gpurun)
a="$1"; b="$2"; n="${3:-3}"
for r in $(seq $n); do
  for v in "$a" "$b"; do
    echo -n "$v: "
    NNHIP_ALLOW_TOOLING_LIB=1 NNHIP_LIB_NAME=$v python bench.py --no-cpu-baseline --steps 30 --no-train-leg --no-strong-leg --no-box-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); c=d['kernel_classes']
print(d['value'], d['ms_per_step'], {k: round(v['ms_per_step'],3) for k,v in c.items() if k.startswith('edge_') or k in ('mlp128','lin128')})"
  done
done
