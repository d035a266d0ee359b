#!/usr/bin/env python3
"""The literal route (conv = rocfft) with its transforms hand-written (jx_fft.hpp: columns and rows, default; columns only with
JOXSZ_FFT_ROWS=rocfft) against the same route with rocFFT's own 2-D plans (JOXSZ_FFT_COLUMNS=rocfft): convolved map, extracted row and log-posterior differences, ms per launch and the stage split.
    python scripts/literal_cols.py [W]      (GPU box)"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from joxsz_amd import datasets
from joxsz_amd.posterior import JoxszPosterior
W = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
SHAPES = [(64, 80, 16), (96, 120, 16), (128, 150, 64), (256, 300, 256), (512, 500, W), (1024, 1000, min(W, 256))]
for S, N, nw in SHAPES:
    pb = datasets.synthetic_problem(S=S, N=N, seed=0)
    th = np.ascontiguousarray(datasets.walker_ball(pb, nw, spread=0.02, seed=1))
    res = {}
    for cols, opts in (('rocfft', {'FFT_COLUMNS': 'rocfft'}), ('custom', {'FFT_ROWS': 'rocfft'}), ('all', {})):
        post = JoxszPosterior(pb, device=0, conv='rocfft', max_batch=nw, options=opts)
        c = post.ctx
        small = th[:min(nw, 4)]
        conv = c.eval_stage(small, 'conv_2d'); row = c.eval_stage(small, 'map_row')
        tp, lp = c.dev_alloc(th.nbytes), c.dev_alloc(8 * nw)
        c.h2d(tp, th)
        for _ in range(2):
            c.eval_device(tp, nw, lp)
        c.sync()
        t = time.perf_counter()
        for _ in range(5):
            c.eval_device(tp, nw, lp)
        c.sync()
        ms = (time.perf_counter() - t) / 5 * 1e3
        c.timing_enable(1); c.timing_reset()
        for _ in range(3):
            c.eval_device(tp, nw, lp)
        tm = c.timing(); c.timing_enable(0)
        out = np.empty(nw); c.d2h(out, lp)
        res[cols] = (conv, row, out, ms, {k[:-3]: round(v / 3, 3) for k, v in tm.items() if k.endswith('_ms') and v}, c.fft_pad, c.fft_info())
        post.close()
    a = res['rocfft']
    fin = np.isfinite(a[2])
    print('S=%d N=%d W=%d pad %d (radices %s | %s): rocFFT 2-D plans %.3f ms %s' % (S, N, nw, a[5], res['all'][6].get('radices_padded'), res['all'][6].get('radices_window'), a[3], a[4]), flush=True)
    for name in ('custom', 'all'):
        b = res[name]
        print('    %-28s conv %.1e row %.1e logp %.1e (nonfinite agree: %s) %.3f ms %s'
              % ('hand-written columns:' if name == 'custom' else 'hand-written rows + columns:', np.max(np.abs(a[0] - b[0])) / np.max(np.abs(a[0])),
                 np.max(np.abs(a[1] - b[1])) / np.max(np.abs(a[1])), np.max(np.abs(a[2][fin] - b[2][fin]) / np.abs(a[2][fin])),
                 bool(np.all(fin == np.isfinite(b[2]))), b[3], b[4]), flush=True)
