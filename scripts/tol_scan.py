"""Singular-value cut of the default route (JOXSZ_LOWRANK_TOL: a cut set by hand is measured by the guard but never tightened) against
rank, what jx_finalize's guard measures, the log-posterior and the step time, at the headline shape."""
import os, sys, time, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from joxsz_amd import datasets
from joxsz_amd.posterior import JoxszPosterior
S, N = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (512, 500)
fwhm = float(sys.argv[3]) if len(sys.argv) > 3 else 18.5
spread = float(sys.argv[4]) if len(sys.argv) > 4 else 0.03
pb = datasets.synthetic_problem(S=S, N=N, seed=0, fwhm=fwhm)
print('S=%d N=%d fwhm=%.1f B=%d, walker ball %.2f' % (S, N, fwhm, pb.B, spread), flush=True)
th = datasets.walker_ball(pb, 1024, spread=spread, seed=1)
ref = None
for tol in ('1e-13', '1e-10', '1e-9', '1e-8', '3e-8', '1e-7', '9e-7'):
    os.environ['JOXSZ_LOWRANK_TOL'] = tol
    post = JoxszPosterior(pb, device=0)
    lp = post.log_prob(th)
    row = post.stage(th[:64], 'map_row')
    chi = post.stage(th[:64], 'chisq')
    for _ in range(3): post.log_prob(th)
    t = time.perf_counter()
    for _ in range(20): post.log_prob(th)
    dt = (time.perf_counter() - t) / 20
    lay = post.ctx.conv_layout
    tr = post.ctx.truncation
    post.close()
    if ref is None: ref = (lp, row, chi)
    fin = np.isfinite(ref[0])
    print('tol %s: form %s rank %d, guard: centre row %.2e, box SZ log-likelihood %.2e | vs tol 1e-13: row %.2e  logp rel %.2e  abs dchi2/2 max %.2e | %.3f ms per 1024 walkers (host loop)'
          % (tol, lay['form'], lay['rank'], tr['est_rel_row_err'], tr['est_rel_sz_like_err_box'], np.abs(row - ref[1]).max() / np.abs(ref[1]).max(),
             np.max(np.abs(lp[fin] - ref[0][fin]) / np.abs(ref[0][fin])), np.max(np.abs(chi - ref[2])) / 2, dt * 1e3), flush=True)
