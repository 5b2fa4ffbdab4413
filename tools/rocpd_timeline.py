#!/usr/bin/env python3
"""Print the kernel sequence of the LAST step in a rocprofv3 rocpd (.db) kernel trace: start offset, duration and the idle
gap before each dispatch.  A step is delimited by the first kernel name given (default mol_ptr_kernel / cells kernel).
usage: python tools/rocpd_timeline.py results.db [first_kernel_substring] [steps_back]   (steps_back: 0 = the last complete
step; bench.py's last steps are its event-instrumented pass, so pass e.g. 12 to look at a step of the timed region)"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    first = sys.argv[2] if len(sys.argv) > 2 else 'mol_ptr_kernel'
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    name_col = 'name' if 'name' in cols else 'kernel_name'
    rows = cur.execute(f"select {name_col}, start, end, grid_x, workgroup_x from kernels order by start").fetchall()
    starts = [k for k, r in enumerate(rows) if first in r[0]]
    if len(starts) < 2:
        sys.exit('need two steps in the trace')
    back = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    if len(starts) < back + 2:
        sys.exit('not that many steps in the trace')
    a, b = starts[-2 - back], starts[-1 - back]
    t0, prev_end = rows[a][1], rows[a][1]
    busy = 0
    for name, s, e, gx, wx in rows[a:b]:
        print(f'{(s - t0) / 1e3:9.1f} us  +{(s - prev_end) / 1e3:6.1f} gap  {(e - s) / 1e3:8.1f} us  {gx // max(wx, 1):6d}  {name.split("(")[0][:60]}')
        busy += e - s
        prev_end = max(prev_end, e)
    span = rows[b][1] - t0
    print(f'step span {span / 1e3:.1f} us, kernel busy {busy / 1e3:.1f} us, idle {(span - busy) / 1e3:.1f} us, {b - a} dispatches')


if __name__ == '__main__':
    main()
