import os, sys, numpy as np
sys.path.insert(0, '.')
from joxsz_amd import datasets
from joxsz_amd.posterior import JoxszPosterior
for S, N, fwhm, nw in ((128, 150, 9.0, 6), (128, 150, 18.5, 6), (256, 300, 9.0, 6)):
    pb = datasets.synthetic_problem(S=S, N=N, seed=11, fwhm=fwhm)
    th = datasets.walker_ball(pb, nw, spread=0.04, seed=11)
    res = {}
    for mode, env in (('fused', {}), ('lowrank', {'JOXSZ_FUSED': '0'}), ('full', {'JOXSZ_LOWRANK': '0'})):
        os.environ.update(env)
        post = JoxszPosterior(pb, device=0, conv='custom')
        lay = post.ctx.conv_layout
        res[mode] = (post.stage(th, 'map_row'), post.log_prob(th), lay)
        post.close()
        for k in env: del os.environ[k]
    print('S', S, 'B', pb.B, res['fused'][2], flush=True)
    for m in ('fused', 'lowrank'):
        d = np.abs(res[m][0] - res['full'][0]).max(axis=1) / np.abs(res['full'][0]).max()
        print('  ', m, 'map_row rel diff per walker', d)
        bad = np.nonzero(d > 1e-9)[0]
        for w in bad[:2]:
            dd = np.abs(res[m][0][w] - res['full'][0][w]) / np.abs(res['full'][0]).max()
            print('     walker', w, 'bad k:', np.nonzero(dd > 1e-9)[0][:20], dd.max())
