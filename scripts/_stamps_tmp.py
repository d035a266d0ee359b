import os, sys, ctypes
import numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
from joxsz_amd import datasets
from joxsz_amd.posterior import JoxszPosterior
pb = datasets.synthetic_problem(S=512, N=500, seed=0)
post = JoxszPosterior(pb, device=0)
c = post.ctx
W = 1024
big = np.ascontiguousarray(datasets.walker_ball(pb, W, spread=0.03, seed=1))
tp, lp = c.dev_alloc(big.nbytes), c.dev_alloc(8 * W)
c.h2d(tp, big)
for _ in range(5):
    c.eval_device(tp, W, lp)
c.sync()
out = (ctypes.c_ulonglong * 64)()
c.lib.jx_dbg_read(out)
for base in (0, 32):
    st = [out[base + i] for i in range(4)]
    print('block', 'first' if base == 0 else 'last', [round((b - st[0]) * 0.01, 2) for b in st], 'us since kernel entry of that block (100 MHz clock)')
print('first block start -> last block end: %.2f us' % ((out[32 + 3] - out[0]) * 0.01))
