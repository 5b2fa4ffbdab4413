#!/usr/bin/env python3
"""Per-kernel averages of rocprofv3 PMC counters from a rocpd .db (one row per kernel name x grid)."""
import sqlite3
import sys
from collections import defaultdict

db = sqlite3.connect(sys.argv[1])
cur = db.cursor()
rows = cur.execute("select kernel_name, grid_size_x, grid_size_y, workgroup_size_x, counter_name, value, duration, dispatch_id "
                   "from counters_collection").fetchall()
agg = defaultdict(lambda: defaultdict(list))
dur = defaultdict(dict)
for name, gx, gy, wx, cname, val, d, did in rows:
    key = (name.split('(')[0][:48], gx // max(wx, 1), gy)
    agg[key][cname].append(val)
    dur[key][did] = d
names = sorted({c for k in agg for c in agg[k]})
print(f'{"kernel":60s} {"n":>4s} {"avg_us":>8s} ' + ' '.join(f'{n[:18]:>18s}' for n in names))
for key in sorted(agg, key=lambda k: -sum(dur[k].values())):
    n = len(dur[key])
    line = f'{" ".join(map(str, key)):60s} {n:4d} {sum(dur[key].values()) / n / 1e3:8.1f} '
    line += ' '.join(f'{sum(agg[key][c]) / max(len(agg[key][c]), 1):18.0f}' for c in names)
    print(line)
