"""Does a row-weighted truncated SVD of the transfer-function weights Hy[q][kx] reach the guard's bounds with fewer terms?
The map row q far from the centre carries a Compton-y signal orders of magnitude below the central rows, so its weights
need correspondingly less relative accuracy.  CPU experiment on the numpy statement of the contraction (scripts/proto/rowmix.py):
weights w_q, SVD of diag(w) Hy, factors unscaled; error of the extracted row (first 96 outputs) and of chi^2 / 2 at the
fiducial vector and at the corners of the prior box in (a, b, r_p), per rank.   python scripts/proto/weighted_rank.py [S N]"""
import sys, os, itertools
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(__file__), '..', '..'))
sys.path.insert(0, os.path.dirname(__file__))
from rowmix import hy_table
from joxsz_amd import datasets
from oracle import joxsz_oracle as orc
from scipy.interpolate import interp1d

S = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N = int(sys.argv[2]) if len(sys.argv) > 2 else 500
pb = datasets.synthetic_problem(S=S, N=N, sz_only=True)
c = S // 2; Sh = S // 2 + 1; nrow = S - c; B = pb.beam_2d.shape[0]; o = (B - 1) // 2
umap = np.abs(np.arange(S) - c); NU = umap.max() + 1
bm = pb.beam_2d * pb.step ** 2
ub, sb, vbt = np.linalg.svd(bm)
by = ub[:, 0] * sb[0]; bx = vbt[0]
Hy = hy_table(pb.filtering).real
# beam along y folded onto distinct rows: T[q][u]
T = np.zeros((S, NU))
for q in range(S):
    for m in range(max(0, q - o), min(S - 1, q + o) + 1):
        T[q, umap[m]] += by[q - m + o]
Fx = np.zeros((S, NU))
for xx in range(S):
    for n in range(max(0, xx - o), min(S - 1, xx + o) + 1):
        Fx[xx, umap[n]] += bx[xx - n + o]
j = np.arange(S)
cosm = np.cos(2 * np.pi * np.outer(np.arange(Sh), j) / S)
NOUT = 96
idx = (c + np.arange(NOUT)[:, None] - j[None, :]) % S                      # [x][x'']

def row_from_factors(U, V, Q):
    """out[x] for x < NOUT from factors Hy ~ U^T V (U [r][S], V [r][Sh]) and the quadrant Q[u][x']."""
    D = (U @ T) @ Q                                                         # [r][x']
    krho = V @ cosm                                                         # [r][S]
    out = np.zeros(NOUT)
    for rho in range(U.shape[0]):
        kern = krho[rho][idx]                                               # [x][x'']
        out += kern @ (Fx @ D[rho])
    return out

names = list(pb.par_names)
th0 = datasets.fiducial_theta(pb)
thawed = list(pb.thawed_idx)
def vec(**kw):
    t = th0.copy()
    for k, v in kw.items():
        t[thawed.index(names.index(k))] = v
    return t
lo = {k: pb.par_min[names.index(k)] for k in ('a', 'b', 'r_p')}
hi = {k: pb.par_max[names.index(k)] for k in ('a', 'b', 'r_p')}
pts = [th0]
for corner in itertools.product((0, 1), repeat=3):
    kw = {}
    for k, bit in zip(('a', 'b', 'r_p'), corner):
        l, h = lo[k], hi[k]
        kw[k] = (l + 0.02 * (h - l)) if bit == 0 else (h - 0.02 * (h - l))
    pts.append(vec(**kw))
iy = np.array([c + b if c + b < S else c - b for b in range(NU)])
Qs, rows, envs = [], [], []
for t in pts:
    p = orc.pars_dict(pb, t)
    ref = orc.row_chain(pb, orc.press_fun(p, pb.r_pp))
    Q = ref['y_2d'][np.ix_(iy, iy)]
    Qs.append(Q); rows.append(ref['map_row'][:NOUT])
    e = np.abs(ref['y_2d']).max(axis=1); envs.append(e / e.max())
env = np.max(envs, axis=0)                                                  # envelope of the map rows over the probe points, [S]
print('row envelope at |q-c| = 0, 32, 64, 128, 200, 255:', ['%.1e' % env[c + d] for d in (0, 32, 64, 128, 200, 255)])
xk = pb.radius[c:]
E = np.zeros((pb.flux_data.shape[1], nrow))
for k in range(nrow):
    ek = np.zeros(nrow); ek[k] = 1
    E[:, k] = interp1d(xk, ek, 'cubic', fill_value='extrapolate')(pb.flux_data[0])
E = E[:, :NOUT]
conv = 1e3 * (-11.0)                                                        # a typical Compton -> mJy/beam factor
def dchi(out, ref):
    g, g0 = E @ (out * conv), E @ (ref * conv)
    return np.abs(np.sum(((g0 - g) / pb.flux_data[2]) * ((g0 + g) / pb.flux_data[2] + 0 * g))) / 2, np.abs((g - g0) / pb.flux_data[2]).max()
schemes = {'plain': np.ones(S), 'envelope': np.maximum(env, 1e-4), 'sqrt envelope': np.sqrt(np.maximum(env, 1e-6)),
           'envelope floor 1e-2': np.maximum(env, 1e-2)}
for name, w in schemes.items():
    uu, sv, vt = np.linalg.svd(w[:, None] * Hy, full_matrices=False)
    print('== %s: sigma_k/sigma_1 at k = 8, 10, 12, 14, 16, 20: %s' % (name, ['%.1e' % (sv[k] / sv[0]) for k in (8, 10, 12, 14, 16, 20)]))
    for r in (8, 10, 12, 14, 16, 17):
        U = ((uu[:, :r] * sv[:r]).T) / w[None, :]
        V = vt[:r]
        worst_row, worst_sig, centre = 0.0, 0.0, 0.0
        for i, (Q, ref) in enumerate(zip(Qs, rows)):
            out = row_from_factors(U, V, Q)
            er = np.abs(out - ref).max() / np.abs(ref).max()
            _, dsig = dchi(out, ref)
            worst_row = max(worst_row, er); worst_sig = max(worst_sig, dsig)
            if i == 0: centre = er
        print('   rank %2d: row error at the fiducial vector %.1e, worst over the box %.1e of max | worst change of a data residual %.1e sigma' % (r, centre, worst_row, worst_sig))
