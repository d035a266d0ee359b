"""What the default route's truncations cost, as jx_finalize measures it (jx_get_truncation), for a few shapes and beams."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from joxsz_amd import datasets
from joxsz_amd.posterior import JoxszPosterior
for S, N, fwhm in ((512, 500, 18.5), (512, 500, 9.0), (512, 500, 6.5), (1024, 1000, 18.5), (256, 300, 18.5), (513, 500, 18.5)):
    pb = datasets.synthetic_problem(S=S, N=N, seed=0, fwhm=fwhm)
    t = time.time()
    post = JoxszPosterior(pb, device=0)
    print('S=%d N=%d fwhm=%.1f B=%d: %s  layout rank=%s kact=%s  (context %.2f s)'
          % (S, N, fwhm, pb.B, post.ctx.truncation, (post.ctx.conv_layout or {}).get('rank'), (post.ctx.conv_layout or {}).get('kact'), time.time() - t), flush=True)
    post.close()
