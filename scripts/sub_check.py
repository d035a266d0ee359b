#!/usr/bin/env python3
"""What the sub-grid of map samples costs at the guard's own probe points: the extracted row of the full form (nothing
truncated) with the sub-grid against the same form on every distinct sample, at the fiducial vector and the eight corners of the
prior box in (a, b, r_p), 2 % inside.      python scripts/sub_check.py [S N tf_scale]      (GPU box)"""
import itertools, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from joxsz_amd import datasets
from joxsz_amd.posterior import JoxszPosterior

S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else 500
sc = float(sys.argv[3]) if len(sys.argv) > 3 else 0.02
pb = datasets.synthetic_problem(S=S, N=N, seed=0, tf_scale=sc)
names, thawed = list(pb.par_names), list(pb.thawed_idx)
th0 = datasets.fiducial_theta(pb)
pts = [th0]
for corner in itertools.product((0, 1), repeat=3):
    tv = th0.copy()
    for k, bit in zip(('a', 'b', 'r_p'), corner):
        l, h = pb.par_min[names.index(k)], pb.par_max[names.index(k)]
        tv[thawed.index(names.index(k))] = (l + 0.02 * (h - l)) if bit == 0 else (h - 0.02 * (h - l))
    pts.append(tv)
pts = np.array(pts)
os.environ['JOXSZ_MIX_FORM'] = 'full'
os.environ['JOXSZ_TRUNC_PROBE'] = '0'
rows = {}
for sub in ('0', '40,160,14', '40,160,12', '48,160,14', '64,192,14', '40,160,8', '40,160,16', '24,96,12'):
    os.environ['JOXSZ_MIX_SUBSAMPLE'] = sub
    post = JoxszPosterior(pb, device=0)
    rows[sub] = (post.stage(pts, 'map_row'), post.stage(pts, 'pp'), post.ctx.sampling['rows_evaluated'])
    post.close()
ref = rows['0'][0]
for sub, (r, pp, ne) in rows.items():
    err = np.abs(r - ref).max(axis=1) / np.abs(ref).max(axis=1)
    err96 = np.abs(r - ref)[:, :96].max(axis=1) / np.abs(ref).max(axis=1)
    print('S=%d N=%d tf %.3f sub-grid %-10s rows %3d | row err per point (of its max): %s | first 96 outputs: worst %.1e | where the worst sits: output %d'
          % (S, N, sc, sub, ne, ' '.join('%.1e' % e for e in err), err96.max(), int(np.abs(r - ref)[int(np.argmax(err))].argmax())), flush=True)
print('core steepness of the probe profiles: pp[0]/pp[40] = %s' % ' '.join('%.1e' % v for v in rows['0'][1][:, 0] / rows['0'][1][:, 40]))
