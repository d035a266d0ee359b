"""GPU check of jx_abel_gemm_kernel: the spline arrays (y_k, M_k) and the log-posterior with phases 2-3 of the Abel kernel
as one matrix product (default) against the Abel kernel itself (JOXSZ_ABEL_GEMM=0)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from joxsz_amd import datasets
from joxsz_amd.posterior import JoxszPosterior

shapes = [(64, 80, 5), (128, 150, 37), (256, 300, 70), (512, 500, 1024), (171, 313, 33), (1024, 1000, 40)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
for S, N, W in shapes:
    pb = datasets.synthetic_problem(S=S, N=N, seed=S)
    th = datasets.walker_ball(pb, W, spread=0.03, seed=S)
    th[1, 1] = 9.0
    res = {}
    for mode in ('0', '1'):
        os.environ['JOXSZ_ABEL_GEMM'] = mode
        post = JoxszPosterior(pb, device=0, conv='custom')
        lp = post.log_prob(th)
        cf = post.ctx.workspace('splines')                  # [N, tW, 2] walker-minor (y_k, M_k)
        post.close()
        res[mode] = (lp, cf[:N, :W, :].copy())
    a, b = res['0'], res['1']
    fin = np.isfinite(a[0])
    ya, yb = a[1][:, fin, 0].T, b[1][:, fin, 0].T
    ma, mb = a[1][:, fin, 1].T, b[1][:, fin, 1].T
    print('S=%d N=%d W=%d: y rel %.3e  M rel (of row max) %.3e  logp rel %.3e  same-inf %s'
          % (S, N, W, np.max(np.abs(ya - yb) / np.abs(ya).max(axis=1, keepdims=True)),
             np.max(np.abs(ma - mb) / np.abs(ma).max(axis=1, keepdims=True)),
             np.max(np.abs(a[0][fin] - b[0][fin]) / np.abs(a[0][fin])), np.array_equal(fin, np.isfinite(b[0]))), flush=True)
