#!/usr/bin/env python3
"""Ensemble-sampler step rate at the headline shape: host loop (proposals in numpy, one batched log_prob call per half
step) against the device-resident loop (jx_sample).  python scripts/sampler_rate.py [walkers steps]   (on the GPU box)"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from joxsz_amd import datasets                                  # noqa: E402
from joxsz_amd.posterior import JoxszPosterior                  # noqa: E402
from joxsz_amd.sampler import StretchMoveSampler, initial_ball  # noqa: E402


def main():
    W = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 50
    pb = datasets.synthetic_problem(S=512, N=500, seed=0)
    post = JoxszPosterior(pb, device=0)
    t0w = np.repeat(datasets.fiducial_theta(pb)[None, :], 8, axis=0)
    bright = post.ctx.eval_stage(t0w, 'bright')[0]
    xprofs = post.ctx.eval_stage(t0w, 'xprofs')[0]
    post.close()
    datasets.fill_data(pb, bright, xprofs, seed=0)
    post = JoxszPosterior(pb, device=0)
    p0 = initial_ball(post.log_prob, datasets.fiducial_theta(pb), W, spread=0.01, rng=np.random.default_rng(1))
    for route in ('map', 'operator'):
        post.ctx.set_route(route)
        run(post, pb, p0, W, steps, route)
    post.close()


def run(post, pb, p0, W, steps, route):
    post.sample(p0, 2)                                           # warm-up
    td = th = float('inf')
    sm = StretchMoveSampler(W, pb.ndim, post.log_prob, seed=5)
    sm.run(p0, 2)
    for rep in range(3):                                         # best of three: the host side (fresh numpy pages for the
        t = time.perf_counter()                                  # returned chain) adds tens of milliseconds now and then
        chain, lps, nacc = post.sample(p0, steps, seed=5)
        td = min(td, time.perf_counter() - t)
        t = time.perf_counter(); sm.run(p0, steps); th = min(th, time.perf_counter() - t)
    print('%d walkers, %d steps (2 half steps each) at 512^2 / 500, route %s' % (W, steps, route))
    print('device-resident loop (jx_sample): %.2f ms per step, %.0f walker-updates/s, acceptance %.2f'
          % (1e3 * td / steps, W * steps / td, nacc.mean() / steps))
    print('host loop (numpy proposals + log_prob per half step): %.2f ms per step, %.0f walker-updates/s, acceptance %.2f'
          % (1e3 * th / steps, W * steps / th, sm.acceptance_fraction.mean()))


if __name__ == '__main__':
    main()
