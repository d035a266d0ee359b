#!/usr/bin/env python3
"""Idle time between the kernels of a step: rocprofv3 --kernel-trace of bench.py, then start/end timestamps of one
steady-state step.   python scripts/step_gaps.py   (on the GPU box, from the repo root)"""
import csv, glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
out = os.path.join(ROOT, 'gpurun_out', 'gaps')
subprocess.run(['rocprofv3', '--kernel-trace', '--output-format', 'csv', '-d', out, '--', sys.executable, os.path.join(ROOT, 'bench.py'),
                '--steps', '20', '--warmup', '5', '--no-cpu', '--no-full-map', '--no-f32', '--no-other-configs', '--no-other-routes', '--no-host-pointer', '--regions', '2'], check=True, env=dict(os.environ, TMPDIR='/tmp'),
               stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
f = sorted(glob.glob(out + '/*/*kernel_trace.csv'))[-1]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r['Start_Timestamp']))
names = [r['Kernel_Name'].split('(')[0][:40] for r in rows]
# steady state: the last 10 occurrences of the tail kernel delimit 10 steps
tails = [i for i, n in enumerate(names) if 'jx_rowsum_tail_kernel' in n or 'jx_tail_row_kernel' in n]
lo, hi = tails[-11], tails[-1]
busy = gap = 0
per = {}
for i in range(lo + 1, hi + 1):
    s, e = int(rows[i]['Start_Timestamp']), int(rows[i]['End_Timestamp'])
    pe = int(rows[i - 1]['End_Timestamp'])
    busy += e - s; gap += s - pe
    per.setdefault(names[i], [0, 0]); per[names[i]][0] += e - s; per[names[i]][1] += s - pe
print('10 steps: busy %.1f us/step, idle between kernels %.1f us/step' % (busy / 1e4, gap / 1e4))
for n, (b, g) in per.items():
    print('  %-42s busy %7.1f us/step   idle in front of it %6.1f us/step' % (n, b / 1e4, g / 1e4))
