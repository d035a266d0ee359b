#!/usr/bin/env python3
"""HBM traffic of every kernel of one bench step from rocprofv3 PMC counters.

Runs bench.py twice under rocprofv3 (FETCH_SIZE and WRITE_SIZE need separate passes: the TCC
block has 4 slots, FETCH_SIZE takes 3 and WRITE_SIZE 2 -- MI355X_MICROARCH.md "rocprofv3 PMC
slots"), averages the counters over the full-size dispatches of each kernel, applies the gfx950
correction of the same guide (FETCH_SIZE reports half the bytes of a wide coalesced read stream;
WRITE_SIZE is exact for 16-byte-per-lane streaming stores; both are in KiB) and writes
profiles/<tag>_pmc_traffic.json, which bench.py quotes in roofline.traffic.

    python scripts/measure_traffic.py r01            (on the GPU box, from the repo root)
    python scripts/measure_traffic.py r01 literal    (the literal route, scripts/literal_prof.py at 512^2 / 500 / 1024 walkers -> <tag>_pmc_traffic_literal.json)
"""
import collections
import csv
import glob
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


LITERAL = len(sys.argv) > 2 and sys.argv[2] == 'literal'


def collect(counter, outdir):
    subprocess.run(['rm', '-rf', outdir])
    cmd = ['rocprofv3', '--kernel-trace', '--pmc', counter, '--output-format', 'csv', '-d', outdir, '--']
    if LITERAL: cmd += [sys.executable, os.path.join(ROOT, 'scripts', 'literal_prof.py'), '512', '500', '1024', '3']
    else: cmd += [sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '3', '--warmup', '1', '--no-cpu', '--no-f32', '--no-other-configs', '--no-other-routes', '--no-host-pointer', '--regions', '2']
    env = dict(os.environ, TMPDIR='/tmp')
    res = subprocess.run(cmd, check=True, env=env, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    collect.bench = {'config': {'S': 512, 'N': 500, 'chunk': 1024, 'conv': 'rocfft'}} if LITERAL else json.loads(res.stdout.strip().splitlines()[-1])
    f = glob.glob(outdir + '/*/*counter_collection.csv')[0]
    rows = list(csv.DictReader(open(f)))
    big = collections.defaultdict(int)
    for r in rows:
        big[r['Kernel_Name']] = max(big[r['Kernel_Name']], int(r['Grid_Size']))
    acc = collections.defaultdict(list)
    for r in rows:
        if r['Counter_Name'] == counter and int(r['Grid_Size']) == big[r['Kernel_Name']]:
            acc[r['Kernel_Name']].append(float(r['Counter_Value']))
    collect.grid = dict(big)
    return {k: sum(v) / len(v) for k, v in acc.items() if not k.startswith('__amd')}


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else 'r03'
    scratch = os.path.join(ROOT, 'gpurun_out', 'traffic_' + tag)
    fetch = collect('FETCH_SIZE', scratch + '_fetch')
    write = collect('WRITE_SIZE', scratch + '_write')
    cfg = collect.bench['config']
    out = {'unit': 'bytes per launch', 'S': cfg['S'], 'N': cfg['N'], 'walkers_per_launch': cfg['chunk'], 'conv': cfg['conv'],
           'command': 'rocprofv3 --kernel-trace --pmc <FETCH_SIZE|WRITE_SIZE> -- python3 ' + ('scripts/literal_prof.py 512 500 1024 3' if LITERAL else 'bench.py --steps 3 --warmup 1 --no-cpu --no-f32 --no-other-configs'), 'note': 'FETCH_SIZE KiB x 1024 x 2 (gfx950 wide-read correction), WRITE_SIZE KiB x 1024',
           'kernels': {}}
    for k in sorted(set(fetch) | set(write)):
        rd = fetch.get(k, 0.0) * 1024 * 2
        wr = write.get(k, 0.0) * 1024
        out['kernels'][k.split('(')[0].strip()] = {'read_bytes': rd, 'write_bytes': wr, 'total_bytes': rd + wr, 'grid_size': collect.grid.get(k, 0),
                                                     'fetch_size_kib_raw': fetch.get(k, 0.0), 'write_size_kib_raw': write.get(k, 0.0)}
    if LITERAL:
        out['kernels'] = {k: v for k, v in out['kernels'].items() if not k.startswith('twiddle_gen')}
        out['total_bytes'] = sum(v['total_bytes'] for v in out['kernels'].values())
        out['survey_8d_bytes'] = 16.0 * 512 * 512 * 1024
        out['total_over_survey_8d_bytes'] = out['total_bytes'] / out['survey_8d_bytes']
    path = os.path.join(ROOT, 'gpurun_out', '%s_pmc_traffic%s.json' % (tag, '_literal' if LITERAL else ''))
    json.dump(out, open(path, 'w'), indent=1)
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
