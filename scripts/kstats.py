#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --kernel-trace run (rocpd sqlite output), FULL-SIZE launches only: of every kernel
name only the dispatches with that name's largest grid are kept (set-up, probe and tap launches of the same kernel on a
handful of walkers would otherwise pull the averages down).    python scripts/kstats.py <results.db> [csv-out]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    cols = [r[1] for r in db.execute("pragma table_info(kernels)")]
    gcols = [c for c in ('grid_x', 'grid_size_x', 'grid_size', 'workgroup_count_x') if c in cols]
    if gcols:
        g = gcols[0]
        q = ("select k.name, count(*), sum(k.end - k.start) / 1e3, avg(k.end - k.start) / 1e3, min(k.end - k.start) / 1e3, max(k.end - k.start) / 1e3, m.g "
             "from kernels k join (select name, max(%s) as g from kernels group by name) m on k.name = m.name and k.%s = m.g "
             "group by k.name order by sum(k.end - k.start) desc" % (g, g))
    else:
        q = ("select name, count(*), sum(end-start)/1e3, avg(end-start)/1e3, min(end-start)/1e3, max(end-start)/1e3, 0 "
             "from kernels group by name order by sum(end-start) desc")
    rows = db.execute(q).fetchall()
    tot = sum(r[2] for r in rows)
    lines = ['"Name","FullSizeCalls","TotalDurationUs","AverageUs","Percentage","MinUs","MaxUs","Grid"']
    for r in rows:
        lines.append('"%s",%d,%.1f,%.2f,%.2f,%.2f,%.2f,%s' % (r[0], r[1], r[2], r[3], 100.0 * r[2] / tot, r[4], r[5], r[6]))
    out = '\n'.join(lines) + '\n'
    if len(sys.argv) > 2:
        open(sys.argv[2], 'w').write(out)
    print('(full-size launches only; grid column: %s)' % (gcols[0] if gcols else 'none found, all launches'))
    for r in rows[:14]:
        print('%-72s n=%4d avg=%9.1f us  min=%8.1f max=%8.1f  %5.1f%%' % (r[0][:72], r[1], r[3], r[4], r[5], 100.0 * r[2] / tot))


if __name__ == '__main__':
    main()
