"""Walkers drawn uniformly over the whole prior box through the default route (the exact form) against the CPU oracle, beside the other
routes of the same library on the same walkers.   python scripts/box_parity.py [S N nwalkers]   (GPU box)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from joxsz_amd import datasets
from joxsz_amd.posterior import JoxszPosterior
from oracle import joxsz_oracle as orc
import warnings
warnings.simplefilter('ignore')
S = int(sys.argv[1]) if len(sys.argv) > 1 else 256
N = int(sys.argv[2]) if len(sys.argv) > 2 else 300
nw = int(sys.argv[3]) if len(sys.argv) > 3 else 400
pb = datasets.synthetic_problem(S=S, N=N, seed=S + 11)
p0 = orc.pars_dict(pb, datasets.fiducial_theta(pb))
datasets.fill_data(pb, orc.sz_stages(pb, p0)['bright'], orc.calc_profiles(pb, p0), seed=S + 11)
rng = np.random.default_rng(S)
th = np.repeat(datasets.fiducial_theta(pb)[None, :], nw, axis=0)
for k, ip in enumerate(pb.thawed_idx):
    lo, hi = pb.par_min[ip], pb.par_max[ip]
    if np.isfinite(lo) and np.isfinite(hi) and hi > lo:
        th[:, k] = rng.uniform(lo, hi, nw)
    elif pb.par_kind[ip] == 1 and pb.par_sigma[ip] > 0:
        th[:, k] = pb.par_mu[ip] + pb.par_sigma[ip] * rng.uniform(-3.0, 3.0, nw)
want = orc.log_posterior_batch(pb, th)
fin = np.isfinite(want)
print('S=%d N=%d: %d of %d box-uniform walkers finite' % (S, N, fin.sum(), nw))
for name, env in [('default (exact form)', {}), ("device library's exp / log", {'JOXSZ_PREP_FASTMATH': '0'}), ('reference kernels of the exact form', {'JOXSZ_X_PAIRWISE': '0'}),
                  ('contracted forms of round 4', {'JOXSZ_MIX_FORM': 'legacy', 'JOXSZ_QUIET': '1'}),
                  ('round 4 full form, every sample', {'JOXSZ_MIX_FORM': 'full', 'JOXSZ_MIX_SUBSAMPLE': '0', 'JOXSZ_AG_SUBSAMPLE': '0', 'JOXSZ_QUIET': '1'}),
                  ('rocFFT sequence', {'JOXSZ_CONV': 'rocfft'})]:
    for k, v in env.items(): os.environ[k] = v
    post = JoxszPosterior(pb, device=0)
    got = post.log_prob(th)
    chi = post.stage(th, 'chisq')
    lay = post.ctx.conv_layout or {}
    post.close()
    for k in env: del os.environ[k]
    rel = np.abs(got[fin] - want[fin]) / np.abs(want[fin])
    i = int(np.argmax(rel))
    print('%-36s form %-8s same rejections %s | log-posterior rel err max %.2e (walker %d: logp %.6g, chi^2 %.4g) median %.2e'
          % (name, lay.get('form', '-'), np.array_equal(np.isfinite(got), fin), rel.max(), np.flatnonzero(fin)[i], want[fin][i], chi[fin][i], np.median(rel)), flush=True)
