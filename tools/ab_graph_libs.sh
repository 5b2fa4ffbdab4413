#!/usr/bin/env bash
# A/B tooling builds of the library on the neighbor-list class: tools/ab_graph_libs.sh <repeats> libA.so libB.so ...  (through gpurun)
n="$1"; shift
for r in $(seq $n); do
  for v in "$@"; do
    echo -n "$v: "
    NNHIP_ALLOW_TOOLING_LIB=1 NNHIP_LIB_NAME=$v python bench.py --no-cpu-baseline --steps 30 --no-train-leg --no-strong-leg --no-box-leg 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); c=d['kernel_classes']
print(d['value'], d['ms_per_step'], 'graph', round(c['graph']['ms_per_step'],4), c['graph']['launches_per_step'])"
  done
done
