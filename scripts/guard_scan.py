#!/usr/bin/env python3
"""What jx_finalize's truncation guard measures over a family of beams and transfer functions at the headline shape: which
inputs sit inside the bounds at the default cut, which make it take the cap on the rank away, which make it tighten the cut.
    python scripts/guard_scan.py      (GPU box)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from joxsz_amd import datasets
from joxsz_amd.posterior import JoxszPosterior
for fwhm, sc, cc in [(18.5, 0.02, 0.95), (12.0, 0.02, 0.95), (25.0, 0.02, 0.95), (30.0, 0.02, 0.95), (18.5, 0.01, 0.95), (18.5, 0.03, 0.95), (18.5, 0.05, 0.95),
                     (18.5, 0.02, 0.8), (18.5, 0.008, 0.95), (9.0, 0.02, 0.95), (18.5, 0.015, 0.99)]:
    pb = datasets.synthetic_problem(S=512, N=500, seed=0, fwhm=fwhm, tf_scale=sc, tf_c=cc)
    t = time.time()
    post = JoxszPosterior(pb, device=0)
    dt = time.time() - t
    tr, lay = post.ctx.truncation, post.ctx.conv_layout
    print('fwhm %5.1f B %3d tf scale %.3f c %.2f: %s rank %2d (above cut %2d) tol %.0e retried %d cap removed %d | centre %.2e box row %.2e box ll %.2e | %.2f s | %s'
          % (fwhm, pb.B, sc, cc, lay['form'], tr['rank'], tr['rank_above_cut'], tr['tol'], tr['retried'], tr['cap_removed'], tr['est_rel_row_err'], tr['est_rel_row_err_box'],
             tr['est_rel_sz_like_err_box'], dt, tr.get('warning')), flush=True)
    post.close()
