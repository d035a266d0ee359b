#!/usr/bin/env python3
"""End-to-end MCMC on the GPU callable: synthetic CL J1226.9+3332-shaped problem, the built-in
stretch-move sampler (emcee's red/blue structure), all walkers of a half step in one launch.

    python examples/run_chain.py --walkers 256 --steps 200 --S 512 --N 500
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from joxsz_amd import datasets                                  # noqa: E402
from joxsz_amd.posterior import JoxszPosterior                  # noqa: E402
from joxsz_amd.sampler import StretchMoveSampler, initial_ball  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--walkers', type=int, default=256)
    ap.add_argument('--steps', type=int, default=100)
    ap.add_argument('--S', type=int, default=512)
    ap.add_argument('--N', type=int, default=500)
    ap.add_argument('--seed', type=int, default=0)
    a = ap.parse_args()
    pb = datasets.synthetic_problem(S=a.S, N=a.N, seed=a.seed)
    post = JoxszPosterior(pb)
    t0 = datasets.fiducial_theta(pb)
    datasets.fill_data(pb, post.stage(t0, 'bright')[0], post.stage(t0, 'xprofs')[0], seed=a.seed)
    post.close()
    post = JoxszPosterior(pb)                                    # constants are uploaded once per context
    rng = np.random.default_rng(a.seed)
    p0 = initial_ball(post.log_prob, t0, a.walkers, spread=0.01, rng=rng)
    s = StretchMoveSampler(a.walkers, post.ndim, post.log_prob, seed=a.seed + 1)
    t = time.perf_counter()
    chain, lp = s.run(p0, a.steps)
    dt = time.perf_counter() - t
    print('%d walkers x %d steps on %s: %.2f s, %.0f walker-likelihoods/s, acceptance %.2f'
          % (a.walkers, a.steps, post.ctx.device_name, dt, a.walkers * a.steps / dt, s.acceptance_fraction.mean()))
    med = np.median(chain[a.steps // 2:].reshape(-1, post.ndim), axis=0)
    for name, m, tr in zip(post.thawed, med, t0):
        print('  %-18s median %10.4f   (truth %10.4f)' % (name, m, tr))
    post.close()


if __name__ == '__main__':
    main()
