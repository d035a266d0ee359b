#!/usr/bin/env python3
"""Sweep library knobs (environment variables) and print per-stage GPU time.

    python scripts/sweep.py "JOXSZ_MAP_THREADS=256,512,1024" "JOXSZ_MAP_SPLIT=1,2,4" [--S 512 --N 500 --walkers 1024]

Each combination runs in a fresh subprocess (the knobs are read at jx_finalize)."""
import itertools
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    knobs, extra = [], []
    for a in sys.argv[1:]:
        if '=' in a and not a.startswith('--'):
            k, v = a.split('=', 1)
            knobs.append((k, v.split(',')))
        else:
            extra.append(a)
    names = [k for k, _ in knobs]
    for combo in itertools.product(*[v for _, v in knobs]) if knobs else [()]:
        env = dict(os.environ)
        env.update(dict(zip(names, combo)))
        cmd = [sys.executable, os.path.join(ROOT, 'bench.py'), '--no-cpu', '--steps', '10', '--warmup', '2'] + extra
        r = subprocess.run(cmd, env=env, capture_output=True, text=True)
        tag = ' '.join('%s=%s' % kv for kv in zip(names, combo))
        if r.returncode != 0:
            print(tag, 'FAILED', r.stderr.strip().splitlines()[-1:] if r.stderr else '')
            continue
        j = json.loads(r.stdout.strip().splitlines()[-1])
        st = j['stage_ms_per_step']
        print('%-50s value=%9.0f/s  map=%.3f ms/launch (%.0f GB/s, frac %.3f) | per step: prep %.3f map %.3f beam %.3f tf %.3f tail %.3f total %.3f'
              % (tag, j['value'], j['roofline']['launch_ms'], j['roofline']['achieved'], j['roofline']['frac'],
                 st['prep_ms'], st['abel_map_ms'], st['beam_fft_ms'], st['tf_fft_ms'], st['tail_ms'], st['total_ms']), flush=True)


if __name__ == '__main__':
    main()
