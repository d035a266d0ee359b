"""GPU check of the odd-side hand-written route against the rocFFT sequence and the oracle."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from joxsz_amd import datasets
from joxsz_amd.posterior import JoxszPosterior
from oracle import joxsz_oracle as orc

shapes = [(171, 313, 12), (257, 300, 20), (513, 500, 40), (65, 80, 7)]
if len(sys.argv) > 1:
    shapes = [tuple(int(v) for v in a.split(',')) for a in sys.argv[1:]]
for S, N, W in shapes:
    pb = datasets.synthetic_problem(S=S, N=N, seed=S)
    p0 = orc.pars_dict(pb, datasets.fiducial_theta(pb))
    datasets.fill_data(pb, orc.sz_stages(pb, p0)['bright'], orc.calc_profiles(pb, p0), seed=S)
    th = datasets.walker_ball(pb, W, spread=0.03, seed=S)
    res = {}
    for conv in ('rocfft', 'custom'):
        try:
            t = time.time()
            post = JoxszPosterior(pb, device=0, conv=conv)
            tc = time.time() - t
        except Exception as exc:
            print('S=%d conv=%s: %s' % (S, conv, exc)); res = None; break
        res[conv] = (post.log_prob(th), post.stage(th, 'map_row'), post.stage(th, 'chisq'), post.stage(th[:2], 'y_2d'))
        print('  %s: context %.2f s, layout %s' % (conv, tc, post.ctx.conv_layout))
        post.close()
    if not res:
        continue
    a, b = res['rocfft'], res['custom']
    st = orc.sz_stages(pb, orc.pars_dict(pb, th[0]))
    print('S=%d N=%d W=%d: row custom-vs-rocfft %.3e  custom-vs-oracle %.3e | chisq/2 abs %.3e | logp rel %.3e | y2d %.1e | same-inf %s'
          % (S, N, W, np.abs(a[1] - b[1]).max() / np.abs(a[1]).max(), np.abs(b[1][0] - st['map_row']).max() / np.abs(st['map_row']).max(),
             np.abs(a[2] - b[2]).max() / 2, np.nanmax(np.abs(a[0] - b[0]) / np.abs(a[0])), np.abs(a[3] - b[3]).max(),
             np.array_equal(np.isfinite(a[0]), np.isfinite(b[0]))), flush=True)
