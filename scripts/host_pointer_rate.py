import os, sys, time
import numpy as np
sys.path.insert(0, os.getcwd())
from joxsz_amd import datasets
from joxsz_amd.posterior import JoxszPosterior
pb = datasets.synthetic_problem(S=512, N=500, seed=0)
post = JoxszPosterior(pb, device=0)
th = np.ascontiguousarray(datasets.walker_ball(pb, 1024, spread=0.02, seed=1))
for _ in range(5): post.ctx.eval(th)
t = time.perf_counter()
for _ in range(100): post.ctx.eval(th)
dt = (time.perf_counter() - t) / 100
print('jx_eval host pointers: %.4f ms per call = %.0f /s' % (dt * 1e3, 1024 / dt))
post.close()
