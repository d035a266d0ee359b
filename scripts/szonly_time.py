import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from joxsz_amd import datasets
from joxsz_amd.posterior import JoxszPosterior
for so in (False, True):
    pb = datasets.synthetic_problem(S=512, N=500, seed=0, sz_only=so)
    post = JoxszPosterior(pb, device=0)
    c = post.ctx
    W = 1024
    big = np.ascontiguousarray(datasets.walker_ball(pb, W, spread=0.03, seed=1))
    tp, lp = c.dev_alloc(big.nbytes), c.dev_alloc(8 * W)
    c.h2d(tp, big)
    for _ in range(3): c.eval_device(tp, W, lp)
    c.sync()
    t1 = time.perf_counter()
    for _ in range(50): c.eval_device(tp, W, lp)
    c.sync()
    ms = (time.perf_counter() - t1) / 50 * 1e3
    c.timing_enable(True); c.timing_reset()
    for _ in range(20): c.eval_device(tp, W, lp)
    tm = c.timing()
    print('sz_only', so, '%.4f ms' % ms, {k[:-3]: round(v / 20, 4) for k, v in tm.items() if k.endswith('_ms')})
    post.close()
