import os, sys, numpy as np
sys.path.insert(0, '.')
from joxsz_amd import datasets
from joxsz_amd.posterior import JoxszPosterior
for S, N, nw in ((64, 80, 22), (256, 300, 6), (512, 500, 6)):
    pb = datasets.synthetic_problem(S=S, N=N, seed=11)
    th = datasets.walker_ball(pb, nw, spread=0.05, seed=11)
    res = {}
    for mode in ('1', '0'):
        os.environ['JOXSZ_LOWRANK'] = mode
        post = JoxszPosterior(pb, device=0, conv='custom')
        lay = post.ctx.conv_layout
        res[mode] = (post.stage(th, 'map_row'), post.log_prob(th), post.log_prob(th))
        post.close()
    a, b = res['1'], res['0']
    print('S', S, 'B', pb.B, lay, flush=True)
    print('  map_row rel diff per walker', np.abs(a[0] - b[0]).max(axis=1) / np.abs(b[0]).max())
    print('  logp lr ', a[1][:6], '\n  logp lr2', a[2][:6], '\n  logp full', b[1][:6])
