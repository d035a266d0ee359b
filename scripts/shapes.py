#!/usr/bin/env python3
"""Rates and parity of the default route at the other BASELINE shapes and at odd sides, with the synthetic (reference-default,
joxsz_funcs.py:90-102, 69-71) and the MEASURED (bundled files) beam and transfer function: per shape the form the cost model
picked, rank, what the guard measured, context build time, walker-likelihoods/s, and the log-posterior against the rocFFT
sequence and the CPU oracle.     python scripts/shapes.py [quick]      (GPU box; reads tests/golden/bundled_inputs.npz)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from joxsz_amd import setup_host as sh, datasets
from joxsz_amd.posterior import JoxszPosterior
from oracle import joxsz_oracle as orc

quick = len(sys.argv) > 1 and sys.argv[1] == 'quick'
z = np.load(os.path.join(ROOT, 'tests', 'golden', 'bundled_inputs.npz'))
prof = sh.clip_beam_profile(z['beam_r'], z['beam_prof'])


def problem(S, N, measured, **kw):
    pb = datasets.synthetic_problem(S=S, N=N, seed=S, **kw)           # geometry, X-ray side and parameters: synthetic
    if measured:
        beam_2d, _ = sh.beam_image(2., 116.0, approx=False, profile=prof)             # measured beam (joxsz_funcs.py:46-76)
        wn, tf = sh.transfer_function(z['wn_as'], z['tf'], approx=False)              # measured transfer function
        pb.beam_2d = np.ascontiguousarray(beam_2d)
        pb.filtering = np.ascontiguousarray(sh.filter_image(wn, tf, S, 2.))
        pb = pb.validate()
    return pb


def rate(post, pb, W, reps=20):
    c = post.ctx
    big = np.ascontiguousarray(datasets.walker_ball(pb, W, spread=0.03, seed=1))
    tp, lp = c.dev_alloc(big.nbytes), c.dev_alloc(8 * W)
    c.h2d(tp, big)
    for _ in range(3):
        c.eval_device(tp, W, lp)
    c.sync()
    t1 = time.perf_counter()
    for _ in range(reps):
        c.eval_device(tp, W, lp)
    c.sync()
    return W * reps / (time.perf_counter() - t1)


cases = [('configs[1] 256^2/300 SZ-only, 256 walkers', 256, 300, 256, False, dict(sz_only=True), 'f64'),
         ('256^2/300 SZ-only, 1024 walkers', 256, 300, 1024, False, dict(sz_only=True), 'f64'),
         ('configs[2] 512^2/500 joint (headline)', 512, 500, 1024, False, {}, 'f64'),
         ('512^2/500, 4096 walkers', 512, 500, 4096, False, {}, 'f64'),
         ('bundled side 171^2/313', 171, 313, 1024, False, {}, 'f64'),
         ('257^2/300', 257, 300, 1024, False, {}, 'f64'),
         ('513^2/500', 513, 500, 1024, False, {}, 'f64'),
         ('configs[4] shape 1024^2/1000 f64', 1024, 1000, 1024, False, {}, 'f64'),
         ('configs[4] shape 1024^2/1000 f32', 1024, 1000, 1024, False, {}, 'f32'),
         ('1025^2/1000', 1025, 1000, 1024, False, {}, 'f64'),
         ('MEASURED beam + transfer function 171^2/313', 171, 313, 1024, True, {}, 'f64'),
         ('MEASURED 256^2/300', 256, 300, 1024, True, {}, 'f64'),
         ('MEASURED 257^2/300', 257, 300, 1024, True, {}, 'f64'),
         ('MEASURED 512^2/500', 512, 500, 1024, True, {}, 'f64'),
         ('MEASURED 513^2/500', 513, 500, 1024, True, {}, 'f64')]
if quick:
    cases = [c for c in cases if c[1] <= 513 and c[3] <= 1024]
for label, S, N, W, measured, kw, dtype in cases:
    pb = problem(S, N, measured, **kw)
    th = datasets.walker_ball(pb, 24, spread=0.05, seed=S)
    t = time.time()
    post = JoxszPosterior(pb, device=0, dtype=dtype)
    dt = time.time() - t
    tr, lay = post.ctx.truncation, post.ctx.conv_layout or {}
    a = post.log_prob(th)
    chi_a = post.stage(th, 'chisq') if dtype == 'f64' else None
    r = rate(post, pb, W)
    post.close()
    ref = JoxszPosterior(pb, device=0, conv='rocfft')
    b = ref.log_prob(th)
    chi_b = ref.stage(th, 'chisq')
    ref.close()
    want = orc.log_posterior_batch(pb, th[:2])
    fin = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), fin)
    print('%-46s %9.0f /s | form %s rank %s R %s, guard %s, context %.2f s | vs rocFFT: logp rel %.1e%s | vs oracle %.1e'
          % (label, r, lay.get('form'), lay.get('rank'), lay.get('R'),
             ('centre row %.1e, box SZ-like %.1e, cut %.0e%s' % (tr['est_rel_row_err'], tr['est_rel_sz_like_err_box'], tr['tol'], ', tightened x%d' % tr['retried'] if tr['retried'] else '')) if tr.get('points') else 'n/a (exact form)',
             dt, np.max(np.abs(a[fin] - b[fin]) / np.abs(b[fin])),
             (', |d chi2/2| %.1e' % (np.max(np.abs(chi_a[fin] - chi_b[fin])) / 2)) if chi_a is not None else '',
             max([abs(x - y) / abs(y) for x, y in zip(a[:2], want) if np.isfinite(y)] or [0.0])), flush=True)
