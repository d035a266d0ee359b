import os, sys, numpy as np
sys.path.insert(0, '.')
from joxsz_amd import datasets
from joxsz_amd.posterior import JoxszPosterior
def run(S, N, fwhm, mb, nw=2):
    pb = datasets.synthetic_problem(S=S, N=N, seed=4, fwhm=fwhm)
    th = datasets.walker_ball(pb, nw, spread=0.02, seed=4)
    res = {}
    for mode, env in (('full', {}), ('rows', {'JOXSZ_CONV_XSYM': '0'})):
        os.environ.update(env)
        post = JoxszPosterior(pb, device=0, conv='custom', max_batch=mb)
        res[mode] = (post.log_prob(th), post.log_prob(th), post.stage(th, 'conv_2d'))
        post.close()
        for k in env: del os.environ[k]
    a, b = res['full'][2], res['rows'][2]
    d = np.abs(a - b).max(axis=(1, 2)) / np.abs(b).max()
    print(f'S={S} B={pb.B} mb={mb} nw={nw}: conv rel diff per walker {d}')
    print('   logp full', res['full'][0], res['full'][1], '\n   logp rows', res['rows'][0])
run(1024, 1000, 18.5, 2)
run(1024, 1000, 9.0, 2)
run(1024, 1000, 18.5, 0, nw=3)
run(512, 500, 18.5, 2)
run(512, 500, 18.5, 0, nw=5)
