"""The spline-array product with one column tile per wave (JOXSZ_AG_SINGLE=1) against the paired form: same bits, stage times."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from joxsz_amd import datasets
from joxsz_amd.posterior import JoxszPosterior
for S, N, W in ((512, 500, 1024), (1024, 1000, 1024), (171, 313, 1024), (512, 500, 256)):
    pb = datasets.synthetic_problem(S=S, N=N, seed=0)
    th = np.ascontiguousarray(datasets.walker_ball(pb, W, spread=0.03, seed=1))
    res = {}
    for single in ('0', '1'):
        os.environ['JOXSZ_AG_SINGLE'] = single
        post = JoxszPosterior(pb, device=0)
        c = post.ctx
        tp, lp = c.dev_alloc(th.nbytes), c.dev_alloc(8 * W)
        c.h2d(tp, th)
        for _ in range(3): c.eval_device(tp, W, lp)
        c.sync()
        t1 = time.perf_counter()
        for _ in range(50): c.eval_device(tp, W, lp)
        c.sync()
        ms = (time.perf_counter() - t1) / 50 * 1e3
        c.timing_enable(True); c.timing_reset()
        for _ in range(20): c.eval_device(tp, W, lp)
        tm = c.timing()
        out = np.empty(W); c.d2h(out, lp)
        res[single] = out
        print(S, N, W, 'single', single, '%.4f ms' % ms, {k[:-3]: round(v / 20, 4) for k, v in tm.items() if k.endswith('_ms')}, flush=True)
        post.close()
    print('   bitwise equal:', np.array_equal(res['0'], res['1'], equal_nan=True))
