"""Experiment: do two contexts (two HIP streams) evaluating half the walkers each overlap on one GPU?"""
import sys, time, threading, numpy as np
sys.path.insert(0, '.')
from joxsz_amd import datasets
from joxsz_amd.posterior import JoxszPosterior

W = 1024
pb = datasets.synthetic_problem(S=512, N=500, seed=0)
th = datasets.walker_ball(pb, W, spread=0.02, seed=1)

def mk(n, mb):
    post = JoxszPosterior(pb, device=0, conv='custom', max_batch=mb)
    ctx = post.ctx
    tp = ctx.dev_alloc(th[:n].nbytes); lp = ctx.dev_alloc(8 * n)
    ctx.h2d(tp, th[:n])
    return post, ctx, tp, lp

def run(ctxs, steps):
    def work(c):
        ctx, tp, lp, n = c
        for _ in range(steps):
            ctx.eval_device(tp, n, lp)
        ctx.sync()
    ts = [threading.Thread(target=work, args=(c,)) for c in ctxs]
    t0 = time.perf_counter()
    for t in ts: t.start()
    for t in ts: t.join()
    return time.perf_counter() - t0

for nctx in (1, 2, 4):
    n = W // nctx
    objs = [mk(n, n) for _ in range(nctx)]
    cs = [(o[1], o[2], o[3], n) for o in objs]
    run(cs, 3)
    steps = 30
    el = run(cs, steps)
    print('contexts %d x %d walkers: %.3f ms per %d walkers -> %.0f /s' % (nctx, n, 1e3 * el / steps, W, W * steps / el), flush=True)
    for o in objs: o[0].close()
