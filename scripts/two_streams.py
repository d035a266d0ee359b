#!/usr/bin/env python3
"""One batch of 1024 walkers as one launch sequence against two halves on two contexts (each its own stream), at the headline
shape: do the latency-bound kernels of one half hide behind the other's?   python scripts/two_streams.py   (GPU box)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from joxsz_amd import datasets
from joxsz_amd.posterior import JoxszPosterior
pb = datasets.synthetic_problem(S=512, N=500, seed=0)
W = 1024
big = np.ascontiguousarray(datasets.walker_ball(pb, W, spread=0.03, seed=1))
for parts in (1, 2, 4):
    posts = [JoxszPosterior(pb, device=0, max_batch=W // parts) for _ in range(parts)]
    bufs = []
    for k, p in enumerate(posts):
        c = p.ctx
        th = np.ascontiguousarray(big[k * (W // parts):(k + 1) * (W // parts)])
        tp, lp = c.dev_alloc(th.nbytes), c.dev_alloc(8 * len(th))
        c.h2d(tp, th)
        bufs.append((c, tp, lp, len(th)))
    def step():
        for c, tp, lp, n in bufs:
            c.eval_device(tp, n, lp)
    for _ in range(5):
        step()
    for c, *_ in bufs:
        c.sync()
    t = time.perf_counter()
    for _ in range(100):
        step()
    for c, *_ in bufs:
        c.sync()
    dt = (time.perf_counter() - t) / 100
    print('%d context(s) x %d walkers: %.4f ms per 1024 walkers = %.0f /s' % (parts, W // parts, dt * 1e3, W / dt), flush=True)
    for p in posts:
        p.close()
