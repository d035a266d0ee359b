#!/usr/bin/env python3
"""The bundled (measured) beam profile and transfer function on LARGER maps than the bundled 171^2: what the default route's
singular-value cut does on real inputs (rank, finalize-time probe, re-builds), against the rocFFT sequence and the oracle.
python scripts/real_inputs_large.py      (GPU box; reads tests/golden/bundled_inputs.npz)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from joxsz_amd import setup_host as sh, datasets
from joxsz_amd.posterior import JoxszPosterior
from oracle import joxsz_oracle as orc

z = np.load(os.path.join(ROOT, 'tests', 'golden', 'bundled_inputs.npz'))
step, kpc_as = 2., datasets.KPC_AS_CLJ1226
prof = sh.clip_beam_profile(z['beam_r'], z['beam_prof'])
for S in (171, 257, 513, 512, 256):
    N = S
    pb = datasets.synthetic_problem(S=S, N=N, seed=S)                 # geometry, X-ray side and parameters: synthetic
    beam_2d, fwhm = sh.beam_image(step, 116.0, approx=False, profile=prof)          # measured beam (joxsz_funcs.py:46-76)
    wn, tf = sh.transfer_function(z['wn_as'], z['tf'], approx=False)                # measured transfer function
    pb.beam_2d = np.ascontiguousarray(beam_2d)
    pb.filtering = np.ascontiguousarray(sh.filter_image(wn, tf, S, step))
    pb = pb.validate() if hasattr(pb, 'validate') else pb
    th = datasets.walker_ball(pb, 24, spread=0.05, seed=S)
    t = time.time()
    post = JoxszPosterior(pb, device=0)
    dt = time.time() - t
    tr, lay, conv = post.ctx.truncation, post.ctx.conv_layout or {}, post.ctx.conv
    a = post.log_prob(th)
    chi_a = post.stage(th, 'chisq')
    big = np.ascontiguousarray(datasets.walker_ball(pb, 1024, spread=0.03, seed=1))
    c = post.ctx
    tp, lp = c.dev_alloc(big.nbytes), c.dev_alloc(8 * 1024)
    c.h2d(tp, big)
    for _ in range(3): c.eval_device(tp, 1024, lp)
    c.sync()
    t1 = time.perf_counter()
    for _ in range(10): c.eval_device(tp, 1024, lp)
    c.sync()
    rate = 1024 * 10 / (time.perf_counter() - t1)
    post.close()
    ref = JoxszPosterior(pb, device=0, conv='rocfft')
    b = ref.log_prob(th)
    chi_b = ref.stage(th, 'chisq')
    ref.close()
    want = orc.log_posterior_batch(pb, th[:2])
    fin = np.isfinite(b)
    print('S=%4d B=%d fwhm=%.2f: %.0f walker-likelihoods/s at 1024 walkers | route %s rank %s kact %s | probe %s (context %.1f s) | vs rocFFT: logp rel %.2e, |d chi2/2| %.2e | vs oracle %.2e'
          % (S, pb.B, fwhm, rate, conv, lay.get('rank'), lay.get('kact'), tr, dt, np.max(np.abs(a[fin] - b[fin]) / np.abs(b[fin])),
             np.max(np.abs(chi_a[fin] - chi_b[fin])) / 2, np.max(np.abs(a[:2] - want) / np.abs(want))), flush=True)
