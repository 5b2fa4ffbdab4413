#!/usr/bin/env bash
# The 100k-atom box (BASELINE configs[4]) with 1 / 2 / 4 waves per receiver row in every edge kernel (rows hold ~54 edges there,
# 14 at config 2, for which the per-kernel defaults were tuned).  usage (through gpurun): tools/box_wpr_ab.sh
for w in default 1 2 4; do
  if [ $w = default ]; then unset NNHIP_EDGE_WPR; else export NNHIP_EDGE_WPR=$w; fi
  python - <<PY
import sys
sys.path.insert(0, ".")
import bench
r = bench.box_leg("cuda", steps=5)
print("NNHIP_EDGE_WPR=$w", r["ms_per_step"], r["edge_ms_per_step"])
PY
done
