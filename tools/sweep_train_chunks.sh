#!/usr/bin/env bash
# A/B of the split-K chunk size of the weight-gradient launch (rows per chunk; needs the NNHIP_WGRAD_ROWS hook that
# train_fused.TrainWorkspace had while this was measured -- kept as the record of the sweep):
#   mixed-32 train leg (bench.py --mode train): 32 rows 1.155 ms, 64 1.045, 128 1.053, 256 1.053-1.061
#   ethanol-32 all-HIP graph: 32 rows 0.873 ms, 128 0.879, 256 0.900;  aspirin-128 all-HIP eager: 2.05 / 1.99 / 1.93 ms
#   aspirin-1024: 256 chunks in every case
for rows in 32 64 128 256; do
  echo -n "rows=$rows: "
  NNHIP_WGRAD_ROWS=$rows python bench.py --mode train --steps 40 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readlines()[-1]); print(d['ms_per_step'], d['value'])"
done
