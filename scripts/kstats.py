#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace run (rocpd sqlite output): python scripts/kstats.py <results.db> [csv-out]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    rows = db.execute("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3 "
                      "from kernels group by name order by sum(end-start) desc").fetchall()
    tot = sum(r[2] for r in rows)
    lines = ['"Name","Calls","TotalDurationUs","AverageUs","Percentage","MinUs","MaxUs"']
    for r in rows:
        lines.append('"%s",%d,%.1f,%.2f,%.2f,%.2f,%.2f' % (r[0], r[1], r[2], r[3], 100.0 * r[2] / tot, r[4], r[5]))
    out = '\n'.join(lines) + '\n'
    if len(sys.argv) > 2:
        open(sys.argv[2], 'w').write(out)
    for r in rows[:12]:
        print('%-72s n=%4d avg=%9.1f us  %5.1f%%' % (r[0][:72], r[1], r[3], 100.0 * r[2] / tot))


if __name__ == '__main__':
    main()
