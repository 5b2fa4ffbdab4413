#!/usr/bin/env bash
# A/B builds of the library on the small-system paths: tools/ab_latency.sh <repeats> libA.so libB.so ...   (through gpurun)
# prints energy_forces of one molecule, the calculator's skin-list step and the mixed-32 training step for each build
n="$1"; shift
for r in $(seq $n); do
  for v in "$@"; do
    echo "== $v"
    NNHIP_ALLOW_TOOLING_LIB=1 NNHIP_LIB_NAME=$v python tools/bench_latency.py 2>&1 | grep "B=  1\|skin list, direct"
    NNHIP_ALLOW_TOOLING_LIB=1 NNHIP_LIB_NAME=$v python bench.py --no-cpu-baseline --no-train-roofline --steps 5 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print('train ms', d['train']['ms_per_step'], ' config2 ms', d['ms_per_step'])"
  done
done
