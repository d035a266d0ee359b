import os, sys, numpy as np
sys.path.insert(0, '.')
from joxsz_amd import datasets
from joxsz_amd.posterior import JoxszPosterior
S = int(sys.argv[1]); N = int(sys.argv[2])
pb = datasets.synthetic_problem(S=S, N=N, seed=4)
th = datasets.walker_ball(pb, 2, spread=0.02, seed=4)
post = JoxszPosterior(pb, device=0, conv='custom', max_batch=2)
lp = post.log_prob(th)
ctx = post.ctx
y, _ = ctx.workspace('y_2d'); Y, xs = ctx.workspace('row_spectra'); C, _ = ctx.workspace('fir_rows'); x0, _ = ctx.workspace('col0')
jrow, _ = ctx.workspace('job_rows'); umap, _ = ctx.workspace('row_index')
jrow = jrow[0, :, 0]; umap = umap[0, :, 0]
print('xsym', xs, 'shapes', y.shape, Y.shape, C.shape, x0.shape, 'logp', lp)
P = 2 * (Y.shape[2] - 1) if xs else (Y.shape[2] - 2)
Ph = P // 2 + 1
c = S // 2
k = np.arange(Ph)
phase = np.exp(2j * np.pi * k * c / P)
for w in range(2):
    rows = np.array([np.nonzero(umap == u)[0][0] for u in range(Y.shape[1])])
    spec = np.fft.rfft(y[w][rows], n=P, axis=1)
    if xs:
        want = ((spec - y[w][rows][:, :1]) * phase)
        print(' w', w, 'imag residual', np.abs(want.imag).max() / np.abs(want).max())
        err = np.abs(Y[w] - want.real).max(axis=1) / np.abs(want).max()
        print(' w', w, 'pass1 R err max', err.max(), 'rows>1e-10:', np.nonzero(err > 1e-10)[0][:20])
        ek = np.abs(Y[w] - want.real).max(axis=0) / np.abs(want).max()
        print('     k>1e-10:', np.nonzero(ek > 1e-10)[0][:20])
post.close()
