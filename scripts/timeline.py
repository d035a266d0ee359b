#!/usr/bin/env python3
"""Start/end of every kernel of the last full step of a rocprofv3 --kernel-trace run, relative to the step's first kernel (us):
shows which kernels ran side by side.     python scripts/timeline.py <results.db>"""
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select name, start, end from kernels order by start").fetchall()
names = [r[0].split('(')[0][:44] for r in rows]
last = max(i for i, n in enumerate(names) if 'jx_tail_row' in n)
first = max(i for i, n in enumerate(names[:last]) if 'jx_tail_row' in n) + 1
t0 = rows[first][1]
for i in range(first, last + 1):
    print('%-46s %9.1f -> %9.1f  (%7.1f us)' % (names[i], (rows[i][1] - t0) / 1e3, (rows[i][2] - t0) / 1e3, (rows[i][2] - rows[i][1]) / 1e3))
