#!/usr/bin/env bash
# Throughput vs batch size (number of aspirin conformers) -- run through gpurun
for c in 16 64 128 256 512 1024 2048 4096 8192; do
  python bench.py --no-cpu-baseline --no-train-leg --no-strong-leg --no-box-leg --no-train-roofline --conformers $c --steps 20 --warmup 5 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); c=d['kernel_classes']
print('conformers %5d: %8.3f ms/step %7.2f M atom-steps/s' % ($c, d['ms_per_step'], d['value']/1e6), {k: round(v['ms_per_step'],3) for k,v in c.items() if k in ('edge_all','mlp128','lin128','graph','other')})"
done
