#!/usr/bin/env python3
"""north_star's literal route (Abel + map kernel, then the rocFFT sequence) at several padded sides of the beam convolution: ms per 1024 walkers and
the stage split.  The default pad is the smallest even 2^a 3^b 5^c >= S + (B-1)/2 (540 at 512^2); VERDICT r04 item 6 asks whether a side whose strided
pass does not over-fetch is faster.      python scripts/literal_pad_scan.py [S N W]      (GPU box)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from joxsz_amd import datasets
from joxsz_amd.posterior import JoxszPosterior
S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else 500
W = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
pb = datasets.synthetic_problem(S=S, N=N, seed=0)
th = np.ascontiguousarray(datasets.walker_ball(pb, W, spread=0.02, seed=1))
ref = None
for pad in (0, 544, 560, 576, 600, 640, 648, 720, 768):
    post = JoxszPosterior(pb, device=0, conv='rocfft', max_batch=W, fft_pad=pad)
    c = post.ctx
    tp, lp = c.dev_alloc(th.nbytes), c.dev_alloc(8 * W)
    c.h2d(tp, th)
    for _ in range(2):
        c.eval_device(tp, W, lp)
    c.sync()
    t = time.perf_counter()
    for _ in range(5):
        c.eval_device(tp, W, lp)
    c.sync()
    ms = (time.perf_counter() - t) / 5 * 1e3
    c.timing_enable(1); c.timing_reset()
    for _ in range(3):
        c.eval_device(tp, W, lp)
    tm = c.timing()
    c.timing_enable(0)
    out = np.empty(W); c.d2h(out, lp)
    if ref is None:
        ref = out
    fin = np.isfinite(ref)
    print('pad %4d: %.3f ms per %d walkers = %.0f /s | stages (ms) %s | vs default pad %.1e'
          % (c.fft_pad, ms, W, W / ms * 1e3, {k[:-3]: round(v / 3, 3) for k, v in tm.items() if k.endswith('_ms') and v}, np.max(np.abs(out[fin] - ref[fin]) / np.abs(ref[fin]))), flush=True)
    post.close()
