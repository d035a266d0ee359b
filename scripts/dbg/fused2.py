import os, sys, numpy as np
sys.path.insert(0, '.')
from joxsz_amd import datasets
from joxsz_amd.posterior import JoxszPosterior
S, N, fwhm, nw = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3]), 6
pb = datasets.synthetic_problem(S=S, N=N, seed=11, fwhm=fwhm)
th = datasets.walker_ball(pb, nw, spread=0.04, seed=11)
os.environ['JOXSZ_FUSED'] = '0'
post = JoxszPosterior(pb, device=0, conv='custom'); lp0 = post.log_prob(th)
Clr, _ = post.ctx.workspace('combined', nw); c0lr, _ = post.ctx.workspace('combined_col0', nw)
post.close(); del os.environ['JOXSZ_FUSED']
post = JoxszPosterior(pb, device=0, conv='custom'); lp = post.log_prob(th)
lay = post.ctx.conv_layout; print(lay)
Ct, _ = post.ctx.workspace('combined_t', nw); Ct0, _ = post.ctx.workspace('combined_col0_t', nw)
Ph = lay['P'] // 2 + 1; r = lay['rank']; nt = (pb.B - 1) // 2 + 1
got = Ct[:, :, :r].transpose(0, 2, 1)          # [w][rho][k]
want = Clr[:, :, :Ph]
d = np.abs(got - want) / np.abs(want).max()
print('Ct vs Clr max rel', d.max(), 'per rho', d.max(axis=(0, 2)))
got0 = Ct0[:, :nt, :r]                           # [w][x][rho]
d0 = np.abs(got0 - c0lr) / np.abs(c0lr).max()
print('Ct0 vs col0lr max rel', d0.max())
print('logp fused', lp, '\nlogp lowrank', lp0)
post.close()
