#!/usr/bin/env python3
"""The timed step alone (no taps, no side measurements): what scripts/prof_*.sh put under rocprofv3.
    python3 scripts/step_only.py [S] [N] [W] [steps]     (GPU box; JOXSZ_* select forms and variants)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from joxsz_amd import datasets
from joxsz_amd.posterior import JoxszPosterior
S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else 500
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 200
pb = datasets.synthetic_problem(S=S, N=N, seed=0)
post = JoxszPosterior(pb, device=0, max_batch=W)
c = post.ctx
big = np.ascontiguousarray(datasets.walker_ball(pb, W, spread=0.03, seed=1))
tp, lp = c.dev_alloc(big.nbytes), c.dev_alloc(8 * W)
c.h2d(tp, big)
for _ in range(5):
    c.eval_device(tp, W, lp)
c.sync()
t1 = time.perf_counter()
for _ in range(steps):
    c.eval_device(tp, W, lp)
c.sync()
ms = (time.perf_counter() - t1) / steps * 1e3
c.timing_enable(1); c.timing_reset()
for _ in range(50):
    c.eval_device(tp, W, lp)
tm = c.timing()
c.timing_enable(0)
print('%d^2/%d x %d walkers, form %s: %.4f ms/step = %.2f M/s | stages (us): %s | env %s'
      % (S, N, W, (c.conv_layout or {}).get('form'), ms, W / ms / 1e3, {k[:-3]: round(1e3 * v / 50, 2) for k, v in tm.items() if k.endswith('_ms') and v},
         {k: v for k, v in os.environ.items() if k.startswith('JOXSZ_')}), flush=True)
post.close()
