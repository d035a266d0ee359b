#!/usr/bin/env python3
"""Exact form (default since round 5) against the rocFFT sequence of the same library and the CPU oracle -- extracted row,
chi^2, log-posterior -- at small, odd, headline and measured-input shapes; then its stage times at the headline shape.
    python scripts/exact_check.py [quick]      (GPU box)"""
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
    pb = datasets.synthetic_problem(S=S, N=N, seed=S, **kw)
    if measured:
        beam_2d, _ = sh.beam_image(2., 116.0, approx=False, profile=prof)
        wn, tf = sh.transfer_function(z['wn_as'], z['tf'], approx=False)
        pb.beam_2d = np.ascontiguousarray(beam_2d)
        pb.filtering = np.ascontiguousarray(sh.filter_image(wn, tf, S, 2.))
        pb = pb.validate()
    return pb


cases = [(31, 40, False, {}), (32, 40, False, {}), (64, 80, False, dict(sz_only=True)), (65, 80, False, dict(ne_mode='double')), (171, 313, False, {}),
         (171, 313, True, {}), (256, 300, False, dict(sz_only=True)), (512, 500, False, {}), (513, 500, False, {}), (512, 512, True, {})]
if not quick:
    cases += [(1024, 1000, False, {}), (1025, 1000, False, {})]
for S, N, measured, kw in cases:
    pb = problem(S, N, measured, **kw)
    th = datasets.walker_ball(pb, 37, spread=0.05, seed=S)
    t = time.time()
    post = JoxszPosterior(pb, device=0)
    dt = time.time() - t
    lay = post.ctx.conv_layout
    a = post.log_prob(th)
    row_a, chi_a = post.stage(th, 'map_row'), post.stage(th, 'chisq')
    a2 = post.log_prob(th)
    post.close()
    ref = JoxszPosterior(pb, device=0, conv='rocfft')
    b = ref.log_prob(th)
    row_b, chi_b = ref.stage(th, 'map_row'), ref.stage(th, 'chisq')
    ref.close()
    want = orc.log_posterior_batch(pb, th[:3])
    fin = np.isfinite(b)
    assert np.array_equal(np.isfinite(a), fin), (a, b)
    assert np.array_equal(a, a2)
    rel = np.max(np.abs(a[fin] - b[fin]) / np.abs(b[fin]))
    rrow = np.max(np.abs(row_a - row_b) / np.max(np.abs(row_b), axis=1, keepdims=True))
    dchi = np.max(np.abs(chi_a - chi_b)[fin]) / 2
    f3 = np.isfinite(want)
    ro = np.max(np.abs(a[:3][f3] - want[f3]) / np.abs(want[f3])) if f3.any() else float('nan')
    print('S %4d N %4d %s %-22s form %s Nk %d/%d nxt %d | context %.2f s | vs rocFFT: row %.1e  |dchi2/2| %.1e  logp rel %.1e | vs oracle %.1e'
          % (S, N, 'measured' if measured else 'synth   ', kw, lay['form'], lay['rank'], lay['beam_terms'], lay['nxt'], dt, rrow, dchi, rel, ro), flush=True)

# stage times at the headline shape
pb = problem(512, 500, False)
post = JoxszPosterior(pb, device=0)
c = post.ctx
W = 1024
big = np.ascontiguousarray(datasets.walker_ball(pb, W, spread=0.03, seed=1))
tp, lp = c.dev_alloc(big.nbytes), c.dev_alloc(8 * W)
c.h2d(tp, big)
for _ in range(5):
    c.eval_device(tp, W, lp)
c.sync()
for reps in (20, 200):
    t1 = time.perf_counter()
    for _ in range(reps):
        c.eval_device(tp, W, lp)
    c.sync()
    dt = time.perf_counter() - t1
    print('exact form, 1024 walkers, %d steps: %.4f ms per step = %.2f M walker-likelihoods/s' % (reps, 1e3 * dt / reps, W * reps / dt / 1e6))
c.timing_enable(1); c.timing_reset()
for _ in range(50):
    c.eval_device(tp, W, lp)
tm = c.timing()
print('stage ms per step (HIP events):', {k: round(v / 50, 5) for k, v in tm.items() if k.endswith('_ms')})
for mode, name in ((2, 'ordinate product'), (3, 'per-walker kernel'), (4, 'row product + tail')):
    c.timing_enable(mode); c.timing_reset()
    for _ in range(50):
        c.eval_device(tp, W, lp)
    tm = c.timing()
    print('  %-20s alone between events: %.2f us' % (name, 1e3 * max(tm['abel_map_ms'], tm['prep_ms'], tm['tail_ms']) / 50))
c.timing_enable(0)
post.close()
