#!/usr/bin/env python3
"""A few launches of the literal route (conv = rocfft) at one shape, for rocprofv3 --kernel-trace --stats.   python3 scripts/literal_prof.py [S N W reps]"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from joxsz_amd import datasets
from joxsz_amd.posterior import JoxszPosterior
S, N, W, reps = [int(a) for a in (sys.argv[1:5] + ['512', '500', '1024', '5'][len(sys.argv) - 1:])]
pb = datasets.synthetic_problem(S=S, N=N, seed=0)
th = np.ascontiguousarray(datasets.walker_ball(pb, W, spread=0.02, seed=1))
post = JoxszPosterior(pb, device=0, conv='rocfft', max_batch=W)
c = post.ctx
tp, lp = c.dev_alloc(th.nbytes), c.dev_alloc(8 * W)
c.h2d(tp, th)
for _ in range(reps):
    c.eval_device(tp, W, lp)
c.sync()
post.close()
