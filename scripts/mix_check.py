#!/usr/bin/env python3
"""Contracted route (conv = 'mix') against the rocFFT sequence and the oracle at several map sides, then its rate and
stage times at the headline shape.   python scripts/mix_check.py [quick]     (GPU box)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from joxsz_amd import datasets
from joxsz_amd.posterior import JoxszPosterior
from oracle import joxsz_oracle as orc

quick = len(sys.argv) > 1 and sys.argv[1] == 'quick'
shapes = [(64, 80), (65, 80), (128, 100), (171, 313)] + ([] if quick else [(256, 300), (257, 300), (512, 500), (513, 500)])
for S, N in shapes:
    pb = datasets.synthetic_problem(S=S, N=N, seed=S)
    th = datasets.walker_ball(pb, 70, spread=0.05, seed=S)
    t0 = time.time()
    post = JoxszPosterior(pb, device=0, conv='custom')
    dt = time.time() - t0
    a = post.log_prob(th)
    row_a = post.stage(th[:6], 'map_row')
    chi_a = post.stage(th, 'chisq')
    tr = post.ctx.truncation
    post.close()
    ref = JoxszPosterior(pb, device=0, conv='rocfft')
    b = ref.log_prob(th)
    row_b = ref.stage(th[:6], 'map_row')
    chi_b = ref.stage(th, 'chisq')
    ref.close()
    want = orc.log_posterior_batch(pb, th[:3])
    fin = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), fin)
    print('S=%4d N=%4d: context %.2f s, rank %d | vs rocFFT: logp rel %.2e, row %.2e of max, |d chi2/2| %.2e | vs oracle %.2e'
          % (S, N, dt, tr['rank'], np.max(np.abs(a[fin] - b[fin]) / np.abs(b[fin])), np.max(np.abs(row_a - row_b)) / np.max(np.abs(row_b)),
             np.max(np.abs(chi_a[fin] - chi_b[fin])) / 2, np.max(np.abs(a[:3] - want) / np.abs(want))), flush=True)

for conv in ('custom',):
    pb = datasets.synthetic_problem(S=512, N=500, seed=0)
    post = JoxszPosterior(pb, device=0, conv=conv)
    c = post.ctx
    W = 1024
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
    print('%s 512^2/500 x %d walkers: %.3f ms/step = %.0f walker-likelihoods/s | stages (ms): %s'
          % (conv, W, ms, W / ms * 1e3, {k: round(v / 20, 4) for k, v in tm.items() if k.endswith('_ms')}), flush=True)
    post.close()
