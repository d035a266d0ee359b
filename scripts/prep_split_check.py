"""The per-walker kernel as two blocks per walker (the X-ray side beside the rest; default) against one block per walker
(JOXSZ_PREP_SPLIT=0): same bits for every walker (rejected ones included), stage times."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from joxsz_amd import datasets
from joxsz_amd.posterior import JoxszPosterior
for S, N, W, kw in ((512, 500, 1024, {}), (512, 500, 1000, dict(ne_mode='double')), (1024, 1000, 1024, {}), (171, 313, 1024, {}), (512, 500, 4096, {})):
    pb = datasets.synthetic_problem(S=S, N=N, seed=0, **kw)
    th = np.ascontiguousarray(datasets.walker_ball(pb, W, spread=0.08, seed=1))      # wide: some walkers leave the box, some fail the vetoes
    th[5, 0] = np.nan
    res = {}
    for sp in ('0', '1'):
        os.environ['JOXSZ_PREP_SPLIT'] = sp
        post = JoxszPosterior(pb, device=0, max_batch=W)
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
        c.timing_enable(False)
        res[sp] = (out, post.log_prob(th[:64]), post.stage(th[:64], 'parts'))
        print(S, N, W, kw, 'two blocks per walker', sp, '%.4f ms' % ms, {k[:-3]: round(v / 20, 4) for k, v in tm.items() if k.endswith('_ms')}, flush=True)
        post.close()
    a, b = res['0'][0], res['1'][0]
    parts = res['1'][2]
    fin = np.isfinite(a[:64])
    print('   finite %d of %d; same rejections: %s; bitwise equal: %s; host-pointer call equal: %s; timed log-posterior = taps prior + X-ray + SZ to %.1e'
          % (np.isfinite(a).sum(), W, np.array_equal(np.isfinite(a), np.isfinite(b)), np.array_equal(a, b), np.array_equal(res['0'][1], res['1'][1]),
             np.max(np.abs(b[:64][fin] - (parts[fin, 0] + parts[fin, 1] + parts[fin, 2])) / np.abs(b[:64][fin]))), flush=True)
