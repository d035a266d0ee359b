#!/usr/bin/env python3
"""A/B of the timed step inside ONE process on one box: every variant (a comma-separated list of OPTION=value, '-' = the defaults) builds its own
context, is timed over 15 regions of 300 steps, and the variants alternate three times (box-to-box and run-to-run spread is +-0.5 us; inside one
process 0.1 us).      python3 scripts/ab_step.py - X_FOLD=0 PREP_LEAN=0,X_FOLD=0      (GPU box)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from joxsz_amd import datasets
from joxsz_amd.posterior import JoxszPosterior
variants = sys.argv[1:] or ['-']
pb = datasets.synthetic_problem(S=512, N=500, seed=0)
big = np.ascontiguousarray(datasets.walker_ball(pb, 1024, spread=0.03, seed=1))


def run(spec):
    opts = dict(kv.split('=', 1) for kv in spec.split(',')) if spec != '-' else {}
    post = JoxszPosterior(pb, device=0, max_batch=1024, options=opts)
    c = post.ctx
    tp, lp = c.dev_alloc(big.nbytes), c.dev_alloc(8 * 1024)
    c.h2d(tp, big)
    for _ in range(20):
        c.eval_device(tp, 1024, lp)
    c.sync()
    us = []
    for rep in range(15):
        t = time.perf_counter()
        for _ in range(300):
            c.eval_device(tp, 1024, lp)
        c.sync()
        us.append((time.perf_counter() - t) / 300 * 1e6)
    post.close()
    return float(np.median(us)), min(us)


for rnd in range(3):
    for spec in variants:
        print('%-40s median %.2f us  min %.2f us' % ((spec,) + run(spec)), flush=True)
