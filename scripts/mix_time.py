#!/usr/bin/env python3
"""Stage times of one route at one shape.  python scripts/mix_time.py [conv] [S] [N] [W] [dtype]    (GPU box)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from joxsz_amd import datasets
from joxsz_amd.posterior import JoxszPosterior
conv = sys.argv[1] if len(sys.argv) > 1 else 'mix'
S = int(sys.argv[2]) if len(sys.argv) > 2 else 512
N = int(sys.argv[3]) if len(sys.argv) > 3 else 500
W = int(sys.argv[4]) if len(sys.argv) > 4 else 1024
dtype = sys.argv[5] if len(sys.argv) > 5 else 'f64'
pb = datasets.synthetic_problem(S=S, N=N, seed=0)
t0 = time.time()
post = JoxszPosterior(pb, device=0, conv=conv, dtype=dtype)
dt = time.time() - t0
c = post.ctx
big = np.ascontiguousarray(datasets.walker_ball(pb, W, spread=0.03, seed=1))
tp, lp = c.dev_alloc(big.nbytes), c.dev_alloc(8 * W)
c.h2d(tp, big)
for _ in range(3):
    c.eval_device(tp, W, lp)
c.sync()
t1 = time.perf_counter()
for _ in range(20):
    c.eval_device(tp, W, lp)
c.sync()
ms = (time.perf_counter() - t1) / 20 * 1e3
c.timing_enable(True); c.timing_reset()
for _ in range(20):
    c.eval_device(tp, W, lp)
tm = c.timing()
print('%s %s %d^2/%d x %d walkers (context %.2f s, %s, %s): %.3f ms/step = %.0f /s | stages (ms): %s | env %s'
      % (conv, dtype, S, N, W, dt, c.conv_layout, c.truncation, ms, W / ms * 1e3, {k[:-3]: round(v / 20, 4) for k, v in tm.items() if k.endswith('_ms')},
         {k: v for k, v in os.environ.items() if k.startswith('JOXSZ_')}), flush=True)
post.close()
