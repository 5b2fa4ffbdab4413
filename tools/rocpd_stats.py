#!/usr/bin/env python3
"""Summarise a rocprofv3 rocpd (.db) kernel trace: per-kernel count / total / avg / min / max, optionally per
grid size.   usage: python tools/rocpd_stats.py results.db [--by-grid] [--skip N_first_dispatches_fraction]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    by_grid = '--by-grid' in sys.argv
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = 'name' if 'name' in cols else 'kernel_name'
    rows = cur.execute(f"select {name_col}, start, end, grid_x, grid_y, workgroup_x from kernels order by start").fetchall()
    agg = {}
    for name, s, e, gx, gy, wx in rows:
        key = (name.split('(')[0][:70], gx // max(wx, 1), gy) if by_grid else (name.split('(')[0][:70],)
        a = agg.setdefault(key, [0, 0, 1 << 62, 0])
        d = e - s
        a[0] += 1
        a[1] += d
        a[2] = min(a[2], d)
        a[3] = max(a[3], d)
    tot = sum(a[1] for a in agg.values())
    print(f'{"kernel":72s} {"calls":>6s} {"total_ms":>9s} {"avg_us":>9s} {"min_us":>9s} {"max_us":>9s} {"%":>6s}')
    for key, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        label = ' '.join(str(k) for k in key)
        print(f'{label:72s} {a[0]:6d} {a[1] / 1e6:9.3f} {a[1] / a[0] / 1e3:9.2f} {a[2] / 1e3:9.2f} {a[3] / 1e3:9.2f} {100 * a[1] / tot:6.2f}')
    print(f'total kernel time {tot / 1e6:.3f} ms over {sum(a[0] for a in agg.values())} dispatches')


if __name__ == '__main__':
    main()
