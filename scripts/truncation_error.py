#!/usr/bin/env python3
"""Effect of the default operator truncation (singular-value cut 1e-10, band limit 1e-11) on the results:
default route against JOXSZ_LOWRANK=0 (every job, every column) and against the tight cut, on a wide walker ball.

    python scripts/truncation_error.py [S N walkers]        (on the GPU box)"""
import os
import subprocess
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

CHILD = r"""
import sys, numpy as np
sys.path.insert(0, sys.argv[1])
from joxsz_amd import datasets
from joxsz_amd.posterior import JoxszPosterior
S, N, W = int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
pb = datasets.synthetic_problem(S=S, N=N, seed=0)
th = datasets.walker_ball(pb, W, spread=0.10, seed=5)
post = JoxszPosterior(pb, device=0, conv='custom')
lay = post.ctx.conv_layout
np.savez(sys.argv[5], logp=post.log_prob(th), row=post.stage(th[:64], 'map_row'), rank=lay['rank'], kact=lay.get('kact', 0))
post.close()
"""


def run(env, out, S, N, W):
    e = dict(os.environ)
    e.update(env)
    subprocess.run([sys.executable, '-c', CHILD, ROOT, str(S), str(N), str(W), out], check=True, env=e)
    return np.load(out)


def main():
    S, N, W = (int(a) for a in sys.argv[1:4]) if len(sys.argv) > 3 else (512, 500, 1024)
    tmp = os.environ.get('TMPDIR', '/tmp')
    full = run({'JOXSZ_LOWRANK': '0'}, os.path.join(tmp, 'te_full.npz'), S, N, W)
    for name, env in (('default cut', {}), ('tight cut (JOXSZ_LOWRANK_TOL=1e-13)', {'JOXSZ_LOWRANK_TOL': '1e-13'})):
        r = run(env, os.path.join(tmp, 'te_x.npz'), S, N, W)
        fin = np.isfinite(full['logp'])
        assert np.array_equal(np.isfinite(r['logp']), fin)
        dl = np.abs(r['logp'][fin] - full['logp'][fin])
        rel = dl / np.abs(full['logp'][fin])
        dr = np.abs(r['row'] - full['row']).max(axis=1) / np.abs(full['row']).max(axis=1)
        print('%-38s rank %3d, columns %3d | log-posterior: max abs %.2e, max rel %.2e, median rel %.2e | extracted row: max rel %.2e'
              % (name, int(r['rank']), int(r['kact']), dl.max(), rel.max(), np.median(rel), dr.max()))
    print('(%d walkers, %d finite, S=%d, N=%d, theta0 * (1 + 0.10 N(0,1)))' % (W, int(fin.sum()), S, N))


if __name__ == '__main__':
    main()
