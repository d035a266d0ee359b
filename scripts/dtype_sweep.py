#!/usr/bin/env python3
"""The three arithmetic variants of the contracted route side by side (BASELINE configs[4]: fp32 variant, fp32-vs-fp64
tolerance sweep): dtype f64 (the reference's), f32 (fp32 spline arrays, every sum in fp64), f32c (fp32 arithmetic in
stage 1 -- packed fp32 FMAs -- and stage 2 -- fp32 matrix cores --, K slices added in fp64): speed on W walkers and accuracy
against f64 on the same walkers, with model-generated data (chi^2 of order the number of data points).
    python scripts/dtype_sweep.py [S N W]      (GPU box)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from joxsz_amd import datasets
from joxsz_amd.posterior import JoxszPosterior

S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else 500
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
pb = datasets.synthetic_problem(S=S, N=N, seed=0)
post = JoxszPosterior(pb, device=0)
t0 = np.repeat(datasets.fiducial_theta(pb)[None, :], 2, axis=0)
datasets.fill_data(pb, post.stage(t0, 'bright')[0], post.stage(t0, 'xprofs')[0], seed=0)
post.close()
res = {}
th = None
for dt in ('f64', 'f32', 'f32c'):
    post = JoxszPosterior(pb, device=0, dtype=dt, max_batch=W)
    c = post.ctx
    if th is None:
        cand = datasets.walker_ball(pb, 4 * W, spread=0.02, seed=1)
        th = np.ascontiguousarray(cand[np.isfinite(post.log_prob(cand))][:W])
    lp = post.log_prob(th)
    chi = post.stage(th[:256], 'chisq')
    row = post.stage(th[:64], 'map_row')
    tp, lpp = c.dev_alloc(th.nbytes), c.dev_alloc(8 * len(th))
    c.h2d(tp, th)
    for _ in range(3):
        c.eval_device(tp, len(th), lpp)
    c.sync()
    t = time.perf_counter()
    for _ in range(20):
        c.eval_device(tp, len(th), lpp)
    c.sync()
    ms = (time.perf_counter() - t) / 20 * 1e3
    c.timing_enable(True); c.timing_reset()
    for _ in range(10):
        c.eval_device(tp, len(th), lpp)
    tm = c.timing()
    res[dt] = (lp, chi, row, ms, {k[:-3]: round(v / 10, 4) for k, v in tm.items() if k.endswith('_ms')}, c.conv_layout, c.truncation)
    post.close()
lp64, chi64, row64 = res['f64'][:3]
print('# %d^2 map, %d-pt grid, %d walkers (3 %% ball around the fiducial vector), log-posterior ~ %.1f, chi^2 ~ %.1f, rank %d'
      % (S, N, len(th), np.median(lp64), np.median(chi64), res['f64'][5]['rank']))
for dt in ('f64', 'f32', 'f32c'):
    lp, chi, row, ms, st, lay, tr = res[dt]
    rel = np.abs(lp - lp64) / np.abs(lp64)
    print('%-5s %.4f ms/step = %9.0f /s (x%.2f) | rel dlogp max %.2e median %.2e | |d chi^2/2| max %.2e median %.2e | row max %.2e of its max | stages %s'
          % (dt, ms, len(th) / ms * 1e3, res['f64'][3] / ms, rel.max(), np.median(rel), np.abs(chi - chi64).max() / 2, np.median(np.abs(chi - chi64)) / 2,
             (np.abs(row - row64).max(axis=1) / np.abs(row64).max(axis=1)).max(), st), flush=True)
