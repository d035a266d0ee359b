"""Can stage 1 skip map samples?  Away from the cluster core the Compton-y map varies on the scale of the radius, far above the pixel:
the quadrant Q[u][x'] is then recoverable from a subset of its rows and columns by local polynomial interpolation, Q ~ L Q_sub L^T, and the
contraction only needs the transformed operators C_sub = L^T C, G_sub = G (L x I).  CPU experiment on the numpy statement of the
contraction: row error (first 96 outputs) of the subsampled tables against the full ones and against the oracle, at the fiducial
vector and at the corners of the prior box.     python scripts/proto/subsample.py [S N]"""
import sys, os, itertools
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, os.path.dirname(__file__))
from rowmix import factor_tables
from joxsz_amd import datasets
from oracle import joxsz_oracle as orc

S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else 500
pb = datasets.synthetic_problem(S=S, N=N, sz_only=True)
t = factor_tables(pb, tol=1e-8)
r, NU = t['r'], t['NU']
NOUT = 96
C = t['C']                                   # [r][NU]
G = t['G'].reshape(-1, r, NU)[:NOUT]         # [x][rho][x']
c = S // 2


def subset(u0, u1, s1, s2, u2=None, s3=8):
    """full resolution below u0, stride s1 up to u1, stride s2 beyond (and s3 beyond u2); the last index always kept"""
    u2 = NU if u2 is None else u2
    idx = list(range(0, u0)) + list(range(u0, u1, s1)) + list(range(u1, min(u2, NU), s2)) + list(range(u2, NU, s3))
    if idx[-1] != NU - 1:
        idx.append(NU - 1)
    return np.array(sorted(set(idx)))


def lagrange_matrix(sub, npts):
    """L [NU][len(sub)]: value at every index u from the npts nearest kept indices (the kept set mirrored about 0: Q is even in u)"""
    ns = len(sub)
    ext = np.concatenate((-sub[1:npts + 1][::-1], sub))          # mirror
    extmap = np.concatenate((np.arange(1, npts + 1)[::-1], np.arange(ns)))
    L = np.zeros((NU, ns))
    for u in range(NU):
        j = np.searchsorted(ext, u)
        lo = max(0, min(len(ext) - npts, j - npts // 2))
        nodes = ext[lo:lo + npts].astype(float)
        for a in range(npts):
            w = 1.0
            for b in range(npts):
                if b != a:
                    w *= (u - nodes[b]) / (nodes[a] - nodes[b])
            L[u, extmap[lo + a]] += w
    return L


names = list(pb.par_names)
th0 = datasets.fiducial_theta(pb)
thawed = list(pb.thawed_idx)
pts = [th0]
for corner in itertools.product((0, 1), repeat=3):
    tv = th0.copy()
    for k, bit in zip(('a', 'b', 'r_p'), corner):
        l, h = pb.par_min[names.index(k)], pb.par_max[names.index(k)]
        tv[thawed.index(names.index(k))] = (l + 0.02 * (h - l)) if bit == 0 else (h - 0.02 * (h - l))
    pts.append(tv)
iy = np.array([c + b if c + b < S else c - b for b in range(NU)])
Qs, refs = [], []
for tv in pts:
    ref = orc.row_chain(pb, orc.press_fun(orc.pars_dict(pb, tv), pb.r_pp))
    Qs.append(ref['y_2d'][np.ix_(iy, iy)]); refs.append(ref['map_row'][:NOUT])


def row(Cm, Gm, Q):
    D = Cm @ Q                                                    # [r][x']
    return np.einsum('xrp,rp->x', Gm, D)


full = [row(C, G, Q) for Q in Qs]
print('full tables vs oracle: %s' % ['%.1e' % (np.abs(f - o).max() / np.abs(o).max()) for f, o in zip(full, refs)])
CASES = [(32, 128, 2, 4, 8, None), (48, 128, 2, 4, 8, None), (64, 160, 2, 4, 8, None), (32, 128, 2, 4, 10, None), (40, 160, 2, 4, 12, None),
         (48, 160, 2, 4, 12, None), (40, 160, 2, 4, 12, 320), (40, 160, 2, 4, 12, 400), (64, 192, 2, 4, 12, 384)]
for (u0, u1, s1, s2, npts, u2) in CASES:
    if u2 is not None and u2 >= NU:
        continue
    sub = subset(u0, u1, s1, s2, u2)
    L = lagrange_matrix(sub, npts)
    Cs = C @ L                                                    # [r][ns]      (C_sub = L^T C in [u][j] layout)
    Gs = np.einsum('xrp,ps->xrs', G, L)                           # [x][rho][ns]
    errs, erro = [], []
    for Q, f, o in zip(Qs, full, refs):
        out = row(Cs, Gs, Q[np.ix_(sub, sub)])
        errs.append(np.abs(out - f).max() / np.abs(f).max()); erro.append(np.abs(out - o).max() / np.abs(o).max())
    print('full < %3d, stride %d to %3d, stride %d beyond%s, %2d-point: %3d of %d rows kept -> %4.1f %% of the samples | vs full tables: fiducial %.1e, worst box %.1e | vs oracle worst %.1e'
          % (u0, s1, u1, s2, '' if u2 is None else ' (8 beyond %d)' % u2, npts, len(sub), NU, 100.0 * len(sub) ** 2 / NU ** 2, errs[0], max(errs), max(erro)), flush=True)
