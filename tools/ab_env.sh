#!/usr/bin/env bash
# A/B an environment switch of the library on the config-2 bench: tools/ab_env.sh <repeats> VAR valA valB ...   (through gpurun)
n="$1"; var="$2"; shift 2
for r in $(seq $n); do
  for v in "$@"; do
    echo -n "$var=$v: "
    env $var=$v python bench.py --no-cpu-baseline --steps 30 --no-train-leg --no-strong-leg --no-box-leg --no-train-roofline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); c=d['kernel_classes']
print(d['value'], d['ms_per_step'], {k: round(v['ms_per_step'],3) for k,v in c.items() if k.startswith('edge_') or k in ('mlp128','lin128')})"
  done
done
