#!/usr/bin/env python3
"""Average of a set of rocprofv3 counters per kernel (full-size launches only: the largest grid of each kernel name).
    python scripts/pmc.py <tag> CNT1,CNT2,... -- <program and args>       (GPU box; program directly after --)
Writes gpurun_out/<tag>_pmc.csv."""
import collections, csv, glob, os, subprocess, sys, shutil
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag, cnts = sys.argv[1], sys.argv[2].split(',')
cmd = sys.argv[sys.argv.index('--') + 1:]
out = os.path.join(ROOT, 'gpurun_out', 'pmc_' + tag)
shutil.rmtree(out, ignore_errors=True)
subprocess.run(['rocprofv3', '--kernel-trace', '--pmc'] + cnts + ['--output-format', 'csv', '-d', out, '--'] + cmd, check=True,
               env=dict(os.environ, TMPDIR='/tmp'), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
f = glob.glob(out + '/*/*counter_collection.csv')[0]
rows = list(csv.DictReader(open(f)))
big = collections.defaultdict(int)
for r in rows:
    big[r['Kernel_Name']] = max(big[r['Kernel_Name']], int(r['Grid_Size']))
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in rows:
    if int(r['Grid_Size']) == big[r['Kernel_Name']] and not r['Kernel_Name'].startswith('__amd'):
        acc[r['Kernel_Name'].split('(')[0][:48]][r['Counter_Name']].append(float(r['Counter_Value']))
lines = ['kernel,launches,' + ','.join(cnts)]
for k, d in acc.items():
    v = [sum(d[c]) / len(d[c]) if d[c] else 0.0 for c in cnts]
    lines.append('%s,%d,' % (k, len(d[cnts[0]])) + ','.join('%.6g' % x for x in v))
txt = '\n'.join(lines) + '\n'
open(os.path.join(ROOT, 'gpurun_out', tag + '_pmc.csv'), 'w').write(txt)
print(txt)
shutil.rmtree(out, ignore_errors=True)
