#!/usr/bin/env python3
"""Time of the per-walker kernel in front of the SZ chain (jx_prep_kernel) with and without its X-ray part and at two grid
lengths: where its time goes.   python scripts/prep_time.py    (GPU box)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from joxsz_amd import datasets
from joxsz_amd.posterior import JoxszPosterior
W = 1024
for S, N, sz_only in ((512, 500, False), (512, 500, True), (128, 100, False), (128, 100, True)):
    pb = datasets.synthetic_problem(S=S, N=N, seed=0, sz_only=sz_only)
    post = JoxszPosterior(pb, device=0, conv='mix')
    c = post.ctx
    big = np.ascontiguousarray(datasets.walker_ball(pb, W, spread=0.03, seed=1))
    tp, lp = c.dev_alloc(big.nbytes), c.dev_alloc(8 * W)
    c.h2d(tp, big)
    for _ in range(3):
        c.eval_device(tp, W, lp)
    c.sync()
    c.timing_enable(True); c.timing_reset()
    for _ in range(20):
        c.eval_device(tp, W, lp)
    tm = c.timing()
    print('S=%d N=%d sz_only=%s: %s' % (S, N, sz_only, {k[:-3]: round(v / 20, 4) for k, v in tm.items() if k.endswith('_ms')}), flush=True)
    post.close()
