"""Can the spline-array product skip radii?  The pressure profile is smooth away from the core: pp on a sub-grid of the radial grid (every
radius below u0, every second to u1, every fourth to 2 u1, every eighth beyond -- jxt::mix_row_subset) carries the others by 12-point Lagrange
interpolation, pp ~ L pp_sub, and the product needs only L^T Tm.  CPU experiment: the spline ordinates y_k and moments M_k (Tm pp, the operator the
library builds: jxt_abel_spline_operator) from the interpolated profile against those from the full one, at the fiducial vector and the corners
of the prior box in (a, b, c, r_p).     python scripts/proto/radial_subsample.py [N]"""
import ctypes, itertools, os, sys
import numpy as np
ROOT = os.path.join(os.path.dirname(__file__), '..', '..')
sys.path.insert(0, ROOT)
from joxsz_amd import datasets
from oracle import joxsz_oracle as orc

N = int(sys.argv[1]) if len(sys.argv) > 1 else 500
lib = ctypes.CDLL(os.path.join(ROOT, 'joxsz_amd', 'csrc', 'libjx_tables_host.so'))
DP, IP = ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_int)
_p = lambda a: a.ctypes.data_as(DP)
pb = datasets.synthetic_problem(S=512 if N <= 600 else 1024, N=N, sz_only=True)
r = np.ascontiguousarray(pb.r_pp, dtype=np.float64)
G = np.zeros((N, N))
lib.jxt_mirrored_spline_op.argtypes = [DP, ctypes.c_int, DP]
lib.jxt_mirrored_spline_op(_p(r), N, _p(G))
lib.jxt_band_halfwidth.argtypes = [DP, ctypes.c_int, ctypes.c_double]
K = lib.jxt_band_halfwidth(_p(G), N, 1e-17)
ld = 2 * N
Tm = np.zeros((N, ld))
lib.jxt_abel_spline_operator.argtypes = [DP, ctypes.c_int, DP, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, DP]
lib.jxt_abel_spline_operator(_p(r), N, _p(G), K, 1.0, N, ld, _p(Tm))
lib.jxt_mix_row_subset.argtypes = [ctypes.c_int] * 3 + [IP]
lib.jxt_mix_interp_matrix.argtypes = [ctypes.c_int, IP, ctypes.c_int, ctypes.c_int, DP]
names, thawed = list(pb.par_names), list(pb.thawed_idx)
th0 = datasets.fiducial_theta(pb)
pts = [th0]
keys = [k for k in ('a', 'b', 'c', 'r_p') if names.index(k) in thawed]
for corner in itertools.product((0, 1), repeat=len(keys)):
    tv = th0.copy()
    for k, bit in zip(keys, corner):
        l, h = pb.par_min[names.index(k)], pb.par_max[names.index(k)]
        tv[thawed.index(names.index(k))] = (l + 0.02 * (h - l)) if bit == 0 else (h - 0.02 * (h - l))
    pts.append(tv)
pps = [orc.press_fun(orc.pars_dict(pb, tv), pb.r_pp) for tv in pts]
kmax = 366 if N <= 600 else 727           # knots the quadrant of the map reaches
print('N = %d, band half-width %d, thawed shape parameters %s, %d probe profiles, core steepness pp[0]/pp[40]: %s'
      % (N, K, keys, len(pps), ' '.join('%.1e' % (q[0] / q[40]) for q in pps[:9])))
for (u0, u1, npts) in ((40, 160, 12), (32, 128, 12), (64, 160, 12), (40, 160, 8), (40, 160, 16), (24, 96, 12), (48, 192, 14), (64, 256, 14), (64, 256, 16), (48, 256, 14), (64, 320, 14), (96, 320, 14), (64, 256, 18)):
    ns = lib.jxt_mix_row_subset(N, u0, u1, None)
    sub = np.zeros(ns, np.int32)
    lib.jxt_mix_row_subset(N, u0, u1, sub.ctypes.data_as(IP))
    L = np.zeros((N, ns))
    lib.jxt_mix_interp_matrix(N, sub.ctypes.data_as(IP), ns, npts, _p(L))
    worst_y = worst_m = worst_pp = 0.0
    for q in pps:
        full = q @ Tm
        got = (L @ q[sub]) @ Tm
        y, m = full[0:2 * kmax:2], full[1:2 * kmax:2]
        worst_y = max(worst_y, np.abs(got[0:2 * kmax:2] - y).max() / np.abs(y).max())
        worst_m = max(worst_m, np.abs(got[1:2 * kmax:2] - m).max() / np.abs(m).max())
        worst_pp = max(worst_pp, (np.abs(L @ q[sub] - q) / np.abs(q)).max())
    print('every radius < %3d, every second to %3d, ... %2d-point: %3d of %d radii kept | worst over the probes: pp rel %.1e, y_k %.1e of max|y|, M_k %.1e of max|M|'
          % (u0, u1, npts, ns, N, worst_pp, worst_y, worst_m), flush=True)
if os.environ.get('WHERE'):
    u0, u1, npts = 40, 160, 12
    ns = lib.jxt_mix_row_subset(N, u0, u1, None); sub = np.zeros(ns, np.int32); lib.jxt_mix_row_subset(N, u0, u1, sub.ctypes.data_as(IP))
    L = np.zeros((N, ns)); lib.jxt_mix_interp_matrix(N, sub.ctypes.data_as(IP), ns, npts, _p(L))
    for i, q in enumerate(pps):
        e = np.abs(L @ q[sub] - q) / np.abs(q)
        j = int(np.argmax(e))
        print('probe %d: worst relative pp error %.1e at radius index %d (kept neighbours %s); beyond index 60: %.1e; the log of the profile interpolated instead: %.1e'
              % (i, e[j], j, sub[np.searchsorted(sub, j) - 2:np.searchsorted(sub, j) + 2], e[60:].max(), (np.abs(np.exp(L @ np.log(q[sub])) - q) / q).max()))
