"""GPU check of jx_rowdct_kernel: walker-minor row spectra, column 0 and log-posterior with pass 1 fed from the spline
coefficients (default) against pass 1 fed from the stored map quadrant (JOXSZ_DCT=0)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from joxsz_amd import datasets
from joxsz_amd.posterior import JoxszPosterior

shapes = [(64, 80, 5), (128, 150, 37), (256, 300, 16), (512, 500, 50), (1024, 1000, 8)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
for S, N, W in shapes:
    pb = datasets.synthetic_problem(S=S, N=N, seed=S)
    th = datasets.walker_ball(pb, W, spread=0.03, seed=S)
    res = {}
    for mode in ('0', '1'):
        os.environ['JOXSZ_DCT'] = mode
        post = JoxszPosterior(pb, device=0, conv='custom')
        lp = post.log_prob(th)
        Rt, _ = post.ctx.workspace('rows_t')
        x0, _ = post.ctx.workspace('x0_t')
        row = post.stage(th, 'map_row')
        lay = post.ctx.conv_layout
        post.close()
        res[mode] = (lp, Rt[:lay['kact'], :lay['NU'], :W].copy(), x0[0, :lay['NU'], :W].copy(), row)
    a, b = res['0'], res['1']
    sc = np.abs(a[1]).max()
    print('S=%d N=%d W=%d kact=%d: Rt max diff %.3e (scale %.3e)  x0 diff %.3e  row rel %.3e  logp rel %.3e  same-inf %s'
          % (S, N, W, lay['kact'], np.abs(a[1] - b[1]).max() / sc, sc, np.abs(a[2] - b[2]).max() / np.abs(a[2]).max(),
             np.abs(a[3] - b[3]).max() / np.abs(a[3]).max(),
             np.nanmax(np.abs(a[0] - b[0]) / np.abs(a[0])), np.array_equal(np.isfinite(a[0]), np.isfinite(b[0]))), flush=True)
    k = np.unravel_index(np.argmax(np.abs(a[1] - b[1])), a[1].shape)
    print('   worst at (k,u,w) =', k, a[1][k], b[1][k])
