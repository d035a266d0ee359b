"""The per-walker kernel with the table-driven exp / log of jx_fastmath.hpp (default) against the device library's (JOXSZ_PREP_FASTMATH=0):
log-posteriors, rejection sets, the stages in between, and the stage times."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from joxsz_amd import datasets
from joxsz_amd.posterior import JoxszPosterior
for S, N, W, kw in ((512, 500, 1024, {}), (512, 500, 1000, dict(ne_mode='double')), (1024, 1000, 1024, {}), (256, 300, 1024, dict(sz_only=True))):
    pb = datasets.synthetic_problem(S=S, N=N, seed=0, **kw)
    th = np.ascontiguousarray(datasets.walker_ball(pb, W, spread=0.08, seed=1))      # wide: some walkers leave the box, some fail the vetoes
    res = {}
    for fm in ('0', '1'):
        os.environ['JOXSZ_PREP_FASTMATH'] = fm
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
        c.timing_enable(False)
        parts = post.stage(th[:256], 'parts')
        tprof = post.stage(th[:256], 'tprof')
        res[fm] = (out, parts, tprof)
        print(S, N, W, kw, 'fastmath', fm, '%.4f ms' % ms, {k[:-3]: round(v / 20, 4) for k, v in tm.items() if k.endswith('_ms')}, flush=True)
        post.close()
    a, b = res['0'][0], res['1'][0]
    fin = np.isfinite(a)
    print('   finite %d of %d; same set of rejections: %s; log-posterior max rel diff %.2e; X-ray log-likelihood max rel diff %.2e; T_SZ profile max rel diff %.2e'
          % (fin.sum(), W, np.array_equal(fin, np.isfinite(b)), np.max(np.abs(a - b)[fin] / np.abs(a[fin])),
             np.nanmax(np.abs(res['0'][1][:, 0] - res['1'][1][:, 0]) / np.maximum(1.0, np.abs(res['0'][1][:, 0]))) if not kw.get('sz_only') else 0.0,
             np.nanmax(np.abs(res['0'][2] - res['1'][2]) / np.abs(res['0'][2]))), flush=True)
