#!/usr/bin/env python3
"""Where the wave cycles of each kernel of a bench step go (rocprofv3 SQ counters, one pass):
    python scripts/sq_counters.py          (on the GPU box; prints a table, writes gpurun_out/sq_counters.csv)
    python scripts/sq_counters.py literal  (the literal route instead, scripts/literal_prof.py; gpurun_out/sq_counters_literal.csv)"""
import collections
import csv
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CNT = ['SQ_WAVE_CYCLES', 'SQ_WAIT_ANY', 'SQ_WAIT_INST_ANY', 'SQ_ACTIVE_INST_ANY', 'SQ_ACTIVE_INST_LDS', 'SQ_LDS_BANK_CONFLICT',
       'SQ_LDS_IDX_ACTIVE', 'SQ_ACTIVE_INST_VALU']


def main():
    literal = len(sys.argv) > 1 and sys.argv[1] == 'literal'
    out = os.path.join(ROOT, 'gpurun_out', 'sq_pmc')
    subprocess.run(['rm', '-rf', out])
    cmd = ['rocprofv3', '--kernel-trace', '--pmc'] + CNT + ['--output-format', 'csv', '-d', out, '--']
    if literal: cmd += [sys.executable, os.path.join(ROOT, 'scripts', 'literal_prof.py')] + sys.argv[2:]
    else: cmd += [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '3', '--warmup', '1', '--no-cpu', '--no-full-map', '--no-f32', '--no-other-configs', '--no-other-routes', '--no-host-pointer', '--regions', '2']
    subprocess.run(cmd, check=True, env=dict(os.environ, TMPDIR='/tmp'), stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    f = glob.glob(out + '/*/*counter_collection.csv')[0]
    rows = list(csv.DictReader(open(f)))
    big = collections.defaultdict(int)
    for r in rows:
        big[r['Kernel_Name']] = max(big[r['Kernel_Name']], int(r['Grid_Size']))
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in rows:
        if int(r['Grid_Size']) == big[r['Kernel_Name']] and not r['Kernel_Name'].startswith('__amd'):
            acc[r['Kernel_Name'].split('(')[0][:40]][r['Counter_Name']].append(float(r['Counter_Value']))
    lines = ['kernel,' + ','.join(CNT)]
    print('%-42s %12s %8s %8s %8s %8s %8s %8s' % ('kernel', 'wave_cycles', 'wait', 'stall', 'active', 'lds', 'valu', 'bankcf/lds'))
    for k, d in acc.items():
        v = {c: (sum(d[c]) / len(d[c]) if d[c] else 0.0) for c in CNT}
        wc = max(v['SQ_WAVE_CYCLES'], 1.0)
        print('%-42s %12.3e %7.1f%% %7.1f%% %7.1f%% %7.1f%% %7.1f%% %8.2f' % (k, wc, 100 * v['SQ_WAIT_ANY'] / wc, 100 * v['SQ_WAIT_INST_ANY'] / wc,
              100 * v['SQ_ACTIVE_INST_ANY'] / wc, 100 * v['SQ_ACTIVE_INST_LDS'] / wc, 100 * v['SQ_ACTIVE_INST_VALU'] / wc,
              v['SQ_LDS_BANK_CONFLICT'] / max(v['SQ_LDS_IDX_ACTIVE'], 1.0)))
        lines.append(k + ',' + ','.join('%.6g' % v[c] for c in CNT))
    open(os.path.join(ROOT, 'gpurun_out', 'sq_counters_literal.csv' if literal else 'sq_counters.csv'), 'w').write('\n'.join(lines) + '\n')


if __name__ == '__main__':
    main()
