"""Host-side table builders of the C library (joxsz_amd/csrc/jx_tables.hpp) against
scipy and the oracle.  CPU only: uses the host-only build libjx_tables_host.so."""
import ctypes
import os

import numpy as np
import pytest
from scipy.interpolate import interp1d, CubicSpline
from scipy.signal import fftconvolve
from scipy.fftpack import fft2, ifft2

from oracle import pyabel_direct

LIB = os.path.join(os.path.dirname(__file__), '..', 'joxsz_amd', 'csrc', 'libjx_tables_host.so')
DP = ctypes.POINTER(ctypes.c_double)


@pytest.fixture(scope='module')
def lib():
    if not os.path.exists(LIB):
        import __graft_entry__
        __graft_entry__.build_host_tables()
    return ctypes.CDLL(LIB)


def _p(a):
    return a.ctypes.data_as(DP)


GRIDS = {
    'uniform': 16.0024 * np.arange(1, 61),
    'arange': np.arange(2 * 8.0012, 5000. + 2 * 8.0012, 2 * 8.0012),      # joxsz_main.py:104
    'ragged': np.cumsum(np.random.default_rng(3).uniform(5., 25., 47)),
}


@pytest.mark.parametrize('name', list(GRIDS))
def test_abel_matrix(lib, name):
    r = np.ascontiguousarray(GRIDS[name])
    n = r.size
    A = np.zeros((n, n))
    lib.jxt_abel_matrix(_p(r), n, _p(A))
    want = pyabel_direct.abel_weight_matrix(r)
    assert np.all(np.tril(A, -1) == 0)
    np.testing.assert_allclose(A, want, rtol=1e-12, atol=1e-12 * np.abs(want).max())
    f = np.exp(-(r / 300.) ** 2)
    np.testing.assert_allclose(A @ f, pyabel_direct.direct_transform_forward(f, r), rtol=1e-12)


@pytest.mark.parametrize('name', list(GRIDS))
def test_mirrored_spline(lib, name):
    r = np.ascontiguousarray(GRIDS[name])
    n = r.size
    G = np.zeros((n, n))
    K = lib.jxt_mirrored_spline_op(_p(r), n, _p(G))
    assert 0 < K <= n
    y = 1. / (1. + (r / 150.) ** 2) ** 1.5
    f = interp1d(np.append(-r, r), np.append(y, y), 'cubic', bounds_error=False, fill_value=(0., 0.))
    cs = CubicSpline(np.concatenate((-r[::-1], r)), np.concatenate((y[::-1], y)), bc_type='not-a-knot')
    M = G @ y
    np.testing.assert_allclose(M, cs(r, 2), rtol=1e-9, atol=1e-12 * np.abs(M).max())
    # evaluate through the per-interval cubic exactly as the kernel does
    q = np.sort(np.random.default_rng(0).uniform(0., r[-1], 400))
    q = np.concatenate(([0., r[0] * 0.5, r[0], r[-1]], q))
    got = np.empty_like(q)
    for m, x in enumerate(q):
        if x < r[0]:
            got[m] = y[0] + 0.5 * M[0] * (x * x - r[0] ** 2)
            continue
        k = min(max(np.searchsorted(r, x, side='right') - 1, 0), n - 2)
        h = r[k + 1] - r[k]
        t = x - r[k]
        b = (y[k + 1] - y[k]) / h - h * (2 * M[k] + M[k + 1]) / 6
        got[m] = y[k] + t * (b + t * (M[k] / 2 + t * (M[k + 1] - M[k]) / (6 * h)))
    np.testing.assert_allclose(got, f(q), rtol=1e-11, atol=1e-14)
    if name != 'ragged':
        assert K < 50          # Green's function of a near-uniform grid decays like 0.268^|i-j|


def test_nak_eval_matrix(lib):
    x = 2.0 * np.arange(86)
    q = np.array([3.136, 9.409, 59.59, 116.1, 170.0, 180.5, -4.0, 0.0, 2.0])
    E = np.zeros((q.size, x.size))
    assert lib.jxt_nak_eval_matrix(_p(x), x.size, _p(q), q.size, _p(E)) == 0
    y = -2.5 * np.exp(-x / 40.) + 0.01 * np.sin(x)
    g = interp1d(x, y, 'cubic', fill_value='extrapolate')
    np.testing.assert_allclose(E @ y, g(q), rtol=1e-11, atol=1e-13)
    qn = np.array([1.0, np.nan])
    En = np.zeros((2, x.size))
    lib.jxt_nak_eval_matrix(_p(x), x.size, _p(qn), 2, _p(En))
    assert np.isnan(En[1]).all() and np.isfinite(En[0]).all()


@pytest.mark.parametrize('S,B,P', [(31, 9, 36), (32, 9, 36), (40, 11, 45), (64, 55, 96)])
def test_beam_spectrum(lib, S, B, P):
    rng = np.random.default_rng(S)
    img = rng.standard_normal((S, S))
    beam = rng.random((B, B))
    Ph = P // 2 + 1
    spec = np.zeros((P, Ph, 2))
    lib.jxt_beam_spectrum(_p(beam), B, P, ctypes.c_double(0.25 / P ** 2), _p(spec))
    bh = spec[..., 0] + 1j * spec[..., 1]
    pad = np.zeros((P, P))
    pad[:S, :S] = img
    conv = np.fft.irfft2(np.fft.rfft2(pad) * bh, s=(P, P)) * P ** 2       # unnormalised pair
    want = fftconvolve(img, beam, 'same') * 0.25
    np.testing.assert_allclose(conv[:S, :S], want, rtol=1e-10, atol=1e-12 * np.abs(want).max())


@pytest.mark.parametrize('S', [31, 32, 171])
def test_tf_row_table(lib, S):
    rng = np.random.default_rng(S)
    x = rng.standard_normal((S, S))
    for sym in (True, False):
        if sym:
            ax = np.linspace(-S // 2 + 1, S // 2, S)
            filt = np.roll(np.exp(-np.sqrt(ax ** 2 + ax[:, None] ** 2) / 9.), S // 2 + 1, axis=(0, 1))
        else:
            filt = rng.random((S, S))          # arbitrary real filter: np.real() symmetrises it
        Sh = S // 2 + 1
        H = np.zeros((S, Sh, 2))
        lib.jxt_tf_row_table(_p(np.ascontiguousarray(filt)), S, _p(H))
        Hc = H[..., 0] + 1j * H[..., 1]
        X = np.fft.rfft2(x)
        Z = (X * Hc).sum(axis=0)
        cols = np.arange(S // 2, S)
        row = np.array([np.real(Z * np.exp(2j * np.pi * np.arange(Sh) * c / S)).sum() for c in cols])
        want = np.real(ifft2(fft2(x) * filt))[S // 2, S // 2:]
        np.testing.assert_allclose(row, want, rtol=1e-10, atol=1e-12 * np.abs(want).max())


def test_next_smooth_even(lib):
    assert lib.jxt_next_smooth_even(539) == 540
    assert lib.jxt_next_smooth_even(198) == 200
    assert lib.jxt_next_smooth_even(1051) == 1080


def test_host_fft(lib):
    rng = np.random.default_rng(1)
    for n in (1, 2, 19, 171, 512, 540):
        x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        for sign, want in ((-1, np.fft.fft(x)), (+1, np.fft.ifft(x) * n)):
            re, im = np.ascontiguousarray(x.real), np.ascontiguousarray(x.imag)
            lib.jxt_host_fft(_p(re), _p(im), n, sign)
            np.testing.assert_allclose(re + 1j * im, want, rtol=0, atol=1e-11 * max(1, np.abs(want).max()))


@pytest.mark.parametrize('name', list(GRIDS))
def test_abel_onfly_tables(lib, name):
    """The kernel regenerates A[i][j] = cj[j]/sqrt(r_j^2-r_i^2) (j >= i+2) on the fly."""
    r = np.ascontiguousarray(GRIDS[name])
    n = r.size
    cj, dg, sp = np.zeros(n), np.zeros(n), np.zeros(n)
    lib.jxt_abel_onfly(_p(r), n, _p(cj), _p(dg), _p(sp))
    A = np.zeros((n, n))
    for i in range(n):
        A[i, i] = dg[i]
        if i + 1 < n:
            A[i, i + 1] = sp[i]
        j = np.arange(i + 2, n)
        A[i, j] = cj[j] / np.sqrt(r[j] ** 2 - r[i] ** 2)
    want = pyabel_direct.abel_weight_matrix(r)
    np.testing.assert_allclose(A, want, rtol=1e-12, atol=1e-13 * np.abs(want).max())


def test_lowrank_factor(lib):
    """Jacobi SVD of the host tables against numpy: singular values, reconstruction at the chosen rank."""
    rng = np.random.default_rng(5)
    m, n, true_rank = 60, 45, 9
    A = rng.standard_normal((m, true_rank)) @ np.diag(10.0 ** -np.arange(true_rank)) @ rng.standard_normal((true_rank, n))
    A += 1e-13 * rng.standard_normal((m, n))
    L = np.zeros((n, m)); Rt = np.zeros((n, n)); sg = np.zeros(n)
    lib.jxt_lowrank_factor.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double, ctypes.c_int] + [ctypes.c_void_p] * 3
    want = np.linalg.svd(A, compute_uv=False)
    for qr in (0, 1):                 # one-sided Jacobi on the whole matrix / rank-revealing QR first, Jacobi on the small factor
        r = lib.jxt_lowrank_factor(_p(np.ascontiguousarray(A)), m, n, 1e-10, qr, _p(L), _p(Rt), _p(sg))
        assert r == int((want > 1e-10 * want[0]).sum()) == true_rank
        np.testing.assert_allclose(sg[:true_rank], want[:true_rank], rtol=1e-11, atol=1e-14 * want[0])
        approx = L[:r].T @ Rt[:r]
        assert np.abs(approx - A).max() < 1e-9 * np.abs(A).max()
        np.testing.assert_allclose(np.linalg.norm(Rt[:r], axis=1), 1.0, rtol=1e-12)
    # full rank request reproduces the matrix to rounding (the QR form stops at the numerical rank: every column below 1e-17 of the largest)
    r2 = lib.jxt_lowrank_factor(_p(np.ascontiguousarray(A)), m, n, 0.0, 0, _p(L), _p(Rt), _p(sg))
    assert r2 == n
    assert np.abs(L[:r2].T @ Rt[:r2] - A).max() < 1e-13 * np.abs(A).max()
    r3 = lib.jxt_lowrank_factor(_p(np.ascontiguousarray(A)), m, n, 1e-15, 1, _p(L), _p(Rt), _p(sg))
    assert true_rank <= r3 <= n
    assert np.abs(L[:r3].T @ Rt[:r3] - A).max() < 1e-12 * np.abs(A).max()


def _dct_tables(lib, Qrad, r, S, LP):
    nb, na = Qrad.shape
    meta = np.zeros(4, np.int32)
    none = ctypes.POINTER(ctypes.c_double)()
    inone = ctypes.POINTER(ctypes.c_int)()
    ok = lib.jxt_dct_tables(_p(Qrad), na, nb, _p(r), len(r), S, LP, meta.ctypes.data_as(IP), inone, none, inone, none, none)
    if not ok:
        return None
    gl, na4, has_x0, amax = (int(v) for v in meta)
    dk, dw = np.zeros((nb, na4), np.int32), np.zeros((nb, na4, 4))
    x0k, x0w, pk = np.zeros(nb, np.int32), np.zeros((nb, 4)), np.zeros((LP // 4 + 1, 4))
    assert lib.jxt_dct_tables(_p(Qrad), na, nb, _p(r), len(r), S, LP, meta.ctypes.data_as(IP), dk.ctypes.data_as(IP), _p(dw),
                              x0k.ctypes.data_as(IP), _p(x0w), _p(pk))
    return dict(gl=gl, na4=na4, has_x0=has_x0, amax=amax, dk=dk, dw=dw, x0k=x0k, x0w=x0w, pk=pk)


def _mirrored_moments(lib, r, y):
    n = len(r)
    G = np.zeros((n, n))
    assert lib.jxt_mirrored_spline_op(_p(r), n, _p(G)) > 0
    return G, G @ y


@pytest.mark.parametrize('name', ['uniform', 'arange', 'ragged'])
def test_abel_spline_operator(lib, name):
    """Tm of jx_abel_gemm_kernel: pp @ Tm gives (y_k, M_k) = (y_scale * PyAbel forward transform, moments of the mirrored
    cubic spline through it), i.e. joxsz_funcs.py:457-460 as one matrix."""
    r = np.ascontiguousarray(GRIDS[name], dtype=np.float64)
    n = len(r)
    G = np.zeros((n, n))
    assert lib.jxt_mirrored_spline_op(_p(r), n, _p(G)) > 0
    lib.jxt_band_halfwidth.argtypes = [DP, ctypes.c_int, ctypes.c_double]
    K = lib.jxt_band_halfwidth(_p(G), n, 1e-20)
    assert 0 < K < n
    y_scale = 3.0856776e21 * 6.6524587158e-25 / 510.9989
    rows, ld = n + 40, 2 * n + 24
    Tm = np.zeros((rows, ld))
    lib.jxt_abel_spline_operator.argtypes = [DP, ctypes.c_int, DP, ctypes.c_int, ctypes.c_double, ctypes.c_int, ctypes.c_int, DP]
    lib.jxt_abel_spline_operator(_p(r), n, _p(G), K, y_scale, rows, ld, _p(Tm))
    assert np.all(Tm[n:] == 0) and np.all(Tm[:, 2 * n:] == 0)
    pp = 0.3 / ((r / 800.) ** 0.014 * (1 + (r / 800.) ** 1.3) ** 3.3)
    cf = pp @ Tm[:n]
    ab = pyabel_direct.direct_transform_forward(pp, r)
    y = y_scale * ab
    np.testing.assert_allclose(cf[0:2 * n:2], y, rtol=1e-12, atol=1e-14 * np.abs(y).max())
    M = G @ y
    assert np.abs(cf[1:2 * n:2] - M).max() < 1e-11 * np.abs(M).max()
    # the triangle the kernel skips: column tile t (knots 8t..8t+7) has no entry above row 8t - K
    for t in range((2 * n + 15) // 16):
        assert np.all(Tm[:max(0, 8 * t - K), 16 * t:16 * t + 16] == 0)




# ------------------------------------------------------------------------------------------------------------------
# contracted route (jx_mix.hpp): the tables the kernels consume, against the oracle's own chain of scipy calls
# ------------------------------------------------------------------------------------------------------------------
def _problem(S, N, beam=None, filt=None):
    from joxsz_amd import datasets
    pb = datasets.synthetic_problem(S=S, N=N, sz_only=True)
    if beam is not None:
        pb.beam_2d = np.ascontiguousarray(beam)
    if filt is not None:
        pb.filtering = np.ascontiguousarray(filt)
    return pb


def _quadrant(img):
    S = img.shape[0]
    c = S // 2
    NU = max(c, S - 1 - c) + 1
    iy = np.array([c + b if c + b < S else c - b for b in range(NU)])
    return img[np.ix_(iy, iy)], NU


def _reference_row(pb, seed=0):
    from joxsz_amd import datasets
    from oracle import joxsz_oracle as orc
    th = datasets.fiducial_theta(pb)
    par = orc.pars_dict(pb, th * (1 + 0.05 * np.random.default_rng(seed).standard_normal(th.size)))
    return orc.row_chain(pb, orc.press_fun(par, pb.r_pp))


@pytest.mark.parametrize('S,N', [(31, 40), (32, 40), (64, 80), (65, 80)])
def test_mix_column_tables_reproduce_the_mirrored_spline(lib, S, N):
    """Stage 1 walks a column of the quadrant as a list of spline intervals: the segment counts and the per-sample weights
    (A, B, C, D) with the spline arrays (y_k, M_k) give interp1d(..., 'cubic')(d_mat) (joxsz_funcs.py:460-462)."""
    pb = _problem(S, N)
    ref = _reference_row(pb)
    Q, NU = _quadrant(ref['y_2d'])
    r = np.ascontiguousarray(pb.r_pp)
    G = np.zeros((N, N))
    assert lib.jxt_mirrored_spline_op(_p(r), N, _p(G)) > 0
    y = ref['y']
    M = G @ y
    Qrad, _ = _quadrant(pb.d_mat)
    Qrad = np.ascontiguousarray(Qrad)
    meta = (ctypes.c_int * 3)()
    IP = ctypes.POINTER(ctypes.c_int)
    lib.jxt_mix_columns.argtypes = [DP, ctypes.c_int, ctypes.c_int, DP, ctypes.c_int, ctypes.c_int, IP, IP, IP, IP, IP, DP]
    ip = lambda a: a.ctypes.data_as(IP)
    ypad, Mpad = np.append(y, np.zeros(4)), np.append(M, np.zeros(4))
    for usplit in (1, 2, 3):                                                       # a column walked whole, or in pieces (one wave each)
        assert lib.jxt_mix_columns(_p(Qrad), NU, NU, _p(r), N, usplit, meta, None, None, None, None, None) == 1
        segld, wld, maxk = meta[0], meta[1], meta[2]
        nv = NU * usplit
        seg0, nseg, urange = np.zeros(nv, np.int32), np.zeros(nv, np.int32), np.zeros(nv, np.int32)
        seg, w4 = np.zeros((nv, segld), np.int32), np.zeros((NU, wld, 4))
        assert lib.jxt_mix_columns(_p(Qrad), NU, NU, _p(r), N, usplit, meta, ip(seg0), ip(nseg), ip(seg), ip(urange), _p(w4)) == 1
        assert segld % 8 == 0 and maxk <= N - 2
        for a in range(NU):
            ks, nxt = [], 0
            for h in range(usplit):
                v = a * usplit + h
                ub, un = urange[v] & 0xffff, urange[v] >> 16
                assert ub == nxt and un >= 1                                       # the pieces tile the column in order
                nxt = ub + un
                assert seg[v, :nseg[v]].sum() == un and np.all(seg[v, nseg[v]:] == 0) and seg[v, 0] >= 1
                ks.append(np.repeat(seg0[v] + np.arange(nseg[v]), seg[v, :nseg[v]]))    # interval of every sample of the piece
            assert nxt == NU
            k = np.concatenate(ks)
            assert np.all(np.diff(k) >= 0)
            w = w4[a, :NU]
            got = w[:, 0] * ypad[k] + w[:, 1] * ypad[k + 1] + w[:, 2] * Mpad[k] + w[:, 3] * Mpad[k + 1]
            np.testing.assert_allclose(got, Q[:, a], rtol=0, atol=2e-13 * np.abs(Q).max())
    # a radius that shrinks along a column is refused (the route then is not taken)
    bad = Qrad.copy()
    bad[NU // 2, 1] = bad[0, 1] * 0.5
    if np.searchsorted(r, bad[NU // 2, 1]) < np.searchsorted(r, bad[NU // 2 - 1, 1]):
        assert lib.jxt_mix_columns(_p(bad), NU, NU, _p(r), N, 1, meta, None, None, None, None, None) == 0


@pytest.mark.parametrize('S,N', [(31, 40), (32, 40), (64, 80), (65, 80), (128, 100)])
def test_mix_lowrank_operators_against_the_oracle(lib, S, N):
    """Stage 1 (C) and stage 2 (G) of the low-rank form, applied in numpy to the oracle's own Compton-y map, give the row
    map_out[S//2, S//2:] of joxsz_funcs.py:464-472 to rounding at the tight cut, and within the cut at a loose one."""
    pb = _problem(S, N)
    ref = _reference_row(pb, seed=S)
    Q, NU = _quadrant(ref['y_2d'])
    nrow = S - S // 2
    beam, filt = np.ascontiguousarray(pb.beam_2d), np.ascontiguousarray(pb.filtering)
    counts = (ctypes.c_int * 2)()
    args = [DP, ctypes.c_int, ctypes.c_double, DP, ctypes.c_int, ctypes.c_double, ctypes.c_double, ctypes.POINTER(ctypes.c_int), DP, DP]
    lib.jxt_mix_lowrank_operators.argtypes = args
    ranks = []
    for tol, bar in ((1e-13, 1e-13), (1e-6, 1e-5)):
        R = lib.jxt_mix_lowrank_operators(_p(beam), pb.B, pb.step ** 2, _p(filt), S, tol, 1e-14, counts, None, None)
        assert R == counts[0] * counts[1] and counts[1] == 1            # the Gaussian beam image is one separable term
        Cm, G = np.zeros((NU, R)), np.zeros((nrow, NU * R))
        lib.jxt_mix_lowrank_operators(_p(beam), pb.B, pb.step ** 2, _p(filt), S, tol, 1e-14, counts, _p(Cm), _p(G))
        D = Cm.T @ Q                                                     # [R][x']: what jx_rowmix_kernel leaves per column
        out = G @ D.T.reshape(-1)                                        # kappa = x' * R + j
        assert np.abs(out - ref['map_row']).max() <= bar * np.abs(ref['map_row']).max()
        ranks.append(R)
    assert ranks[1] < ranks[0]


@pytest.mark.parametrize('S,N', [(31, 40), (32, 40), (64, 80)])
def test_mix_full_operator_with_a_measured_like_beam_and_a_rough_transfer_function(lib, S, N):
    """The full form needs no structure in the beam image (beyond nothing at all: here not even flip symmetry) and none in
    the transfer function beyond real weights: Omega applied to the oracle's map quadrant gives the oracle's row."""
    from oracle.joxsz_oracle import centdistmat, dist
    rng = np.random.default_rng(S)
    B = 9
    ax = np.arange(B) - (B - 1) // 2
    rad = centdistmat(ax)
    beam = np.exp(-rad ** 2 / 7.) * (1 + 0.3 * np.cos(1.7 * rad)) + 0.05 * rng.random((B, B)) * np.exp(-rad / 3.)    # not separable, not symmetric
    beam /= beam.sum() * 4.0
    k = dist(S) / S
    knots = np.sort(rng.random(40)) * k.max() * 1.01
    vals = rng.random(40)
    filt = np.interp(k, knots, vals)                                     # radial, rough from knot to knot
    pb = _problem(S, N, beam=beam, filt=filt)
    ref = _reference_row(pb, seed=S)
    Q, NU = _quadrant(ref['y_2d'])
    nrow = S - S // 2
    Om = np.zeros((nrow, NU, NU))
    lib.jxt_mix_full_operator.argtypes = [DP, ctypes.c_int, ctypes.c_double, DP, ctypes.c_int, DP]
    assert lib.jxt_mix_full_operator(_p(np.ascontiguousarray(pb.beam_2d)), B, pb.step ** 2, _p(np.ascontiguousarray(pb.filtering)), S, _p(Om)) == 0
    out = np.einsum('xuv,uv->x', Om, Q)
    assert np.abs(out - ref['map_row']).max() <= 1e-13 * np.abs(ref['map_row']).max()
    # weights that are not real (a filter without the symmetry in each wavenumber) are refused
    assert lib.jxt_mix_full_operator(_p(np.ascontiguousarray(pb.beam_2d)), B, pb.step ** 2, _p(np.ascontiguousarray(rng.random((S, S)))), S, _p(Om)) == -1


@pytest.mark.parametrize('NU,u0,u1,npts', [(257, 40, 160, 12), (513, 40, 160, 12), (129, 40, 160, 12), (257, 32, 128, 8), (86, 40, 160, 12), (171, 40, 160, 12), (190, 40, 160, 12)])
def test_sub_grid_rows_and_interpolation_matrix(lib, NU, u0, u1, npts):
    """DESIGN 4.2 on the CPU: the rows the sub-grid keeps (every one below u0, every second to u1, every fourth to 2 u1, every eighth
    beyond, the last) and the matrix L that carries values on them to every row -- unit rows at kept indices, rows that sum to
    one, exact on even polynomials up to degree 2 (npts - 1) thanks to the mirrored stencil near the axis (the quadrant is even in
    each index) and on any polynomial of degree < npts away from it."""
    IP = ctypes.POINTER(ctypes.c_int)
    lib.jxt_mix_row_subset.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, IP]
    lib.jxt_mix_interp_matrix.argtypes = [ctypes.c_int, IP, ctypes.c_int, ctypes.c_int, DP]
    ns = lib.jxt_mix_row_subset(NU, u0, u1, None)
    sub = np.zeros(ns, np.int32)
    assert lib.jxt_mix_row_subset(NU, u0, u1, sub.ctypes.data_as(IP)) == ns
    assert sub[0] == 0 and sub[-1] == NU - 1 and np.all(np.diff(sub) >= 1) and np.array_equal(sub[:min(u0, NU)], np.arange(min(u0, NU)))
    # the rule, restated: a coarser stride is entered at its start row only where sixteen of its steps still fit before the last row
    want, u, stride = [], 0, 1
    while u < NU:
        want.append(u)
        for t, st in enumerate((u0, u1, 2 * u1)):
            if u >= st and stride < (2 << t) and NU - 1 - u >= 16 * (2 << t):
                stride = 2 << t
        u += stride
    if want[-1] != NU - 1:
        want.append(NU - 1)
    assert np.array_equal(sub, want)
    L = np.zeros((NU, ns))
    assert lib.jxt_mix_interp_matrix(NU, sub.ctypes.data_as(IP), ns, npts, _p(L)) == 0
    assert np.array_equal(L[sub], np.eye(ns))                            # kept rows: themselves
    np.testing.assert_allclose(L.sum(axis=1), 1.0, atol=1e-12)
    assert (L != 0).sum(axis=1).max() <= npts
    u = np.arange(NU, dtype=np.float64) / NU
    for deg in (2, 4, 2 * (min(npts, ns) - 1) if NU > 200 else 6):       # even powers: reproduced everywhere, the axis included
        f = u ** deg
        np.testing.assert_allclose(L @ f[sub], f, atol=2e-11 * max(1.0, np.abs(L).sum(axis=1).max()))
    far = np.arange(NU) > sub[min(npts, ns - 1)]                         # away from the axis no mirror enters: any polynomial below degree npts
    f = (u - 0.3) ** (min(npts, ns) - 1)
    np.testing.assert_allclose((L @ f[sub])[far], f[far], atol=1e-10)
    # a smooth even function of the radius, the shape of the map along a row: what the interpolation actually meets
    rho = np.hypot(np.arange(NU)[:, None], np.arange(NU)[None, :])
    Q = (1.0 + (rho / 30.0) ** 2) ** -1.2
    Qs = Q[np.ix_(sub, sub)]
    err = np.abs(L @ Qs @ L.T - Q).max() / np.abs(Q).max()
    assert err < (1e-8 if npts < 12 else 1e-10), err


@pytest.mark.parametrize('S,N', [(128, 140), (200, 160)])
def test_full_operator_through_the_sub_grid_against_the_oracle(lib, S, N):
    """The contraction of 4.2 stated in numpy on the ORACLE's own Compton-y map: Omega (every distinct sample) folded through L on
    both indices, applied to the map's samples on the sub-grid alone, gives the oracle's row -- to 1e-11 of its maximum with a
    sub-grid that is coarse for a map this small (every row within 24 pixels, every second to 48, then every fourth where sixteen
    such steps fit before the edge)."""
    IP = ctypes.POINTER(ctypes.c_int)
    lib.jxt_mix_row_subset.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_int, IP]
    lib.jxt_mix_interp_matrix.argtypes = [ctypes.c_int, IP, ctypes.c_int, ctypes.c_int, DP]
    pb = _problem(S, N)
    ref = _reference_row(pb, seed=S)
    Q, NU = _quadrant(ref['y_2d'])
    nrow = S - S // 2
    Om = np.zeros((nrow, NU, NU))
    lib.jxt_mix_full_operator.argtypes = [DP, ctypes.c_int, ctypes.c_double, DP, ctypes.c_int, DP]
    assert lib.jxt_mix_full_operator(_p(np.ascontiguousarray(pb.beam_2d)), pb.B, pb.step ** 2, _p(np.ascontiguousarray(pb.filtering)), S, _p(Om)) == 0
    ns = lib.jxt_mix_row_subset(NU, 24, 48, None)
    sub = np.zeros(ns, np.int32)
    lib.jxt_mix_row_subset(NU, 24, 48, sub.ctypes.data_as(IP))
    assert ns < 0.75 * NU
    L = np.zeros((NU, ns))
    lib.jxt_mix_interp_matrix(NU, sub.ctypes.data_as(IP), ns, 12, _p(L))
    Oms = np.einsum('xuv,ua,vb->xab', Om, L, L)
    full = np.einsum('xuv,uv->x', Om, Q)
    out = np.einsum('xab,ab->x', Oms, Q[np.ix_(sub, sub)])
    scale = np.abs(ref['map_row']).max()
    assert np.abs(full - ref['map_row']).max() <= 1e-13 * scale
    assert np.abs(out - ref['map_row']).max() <= 1e-11 * scale


def test_fastmath_tables_and_the_scheme_they_serve(lib):
    """csrc/jx_fastmath.hpp on the CPU: the tables the per-walker kernel stages in LDS (jxt::fastmath_tables) are what the scheme
    assumes -- 2^(j/64) and, for each of the 128 mantissa intervals, 1/c and log c with c = 1 exactly where 1.0 sits -- and the scheme
    itself, restated in numpy with long double standing in for the fused multiply-adds, stays within 2 ulp of exp and log in long
    double (the device functions are held to the same bar on the GPU: tests/test_gpu_fastmath.py)."""
    tab = np.zeros(320)
    lib.jxt_fastmath_tables.argtypes = [DP]
    assert lib.jxt_fastmath_tables(_p(tab)) == 320
    LD = np.longdouble
    et, lt = tab[:64], tab[64:].reshape(128, 2)
    np.testing.assert_allclose(et.astype(LD), np.exp2(np.arange(64, dtype=LD) / 64), rtol=1.2e-16)
    hi0 = 0x3FE5F000
    z0 = (np.array([(hi0 + (i << 13)) << 32 for i in range(129)], dtype=np.uint64)).view(np.float64)
    assert z0[80] == 1.0 - 2.0 ** -9 and z0[81] == 1.0 + 2.0 ** -8 and lt[80, 0] == 1.0 and lt[80, 1] == 0.0
    for i in range(128):                                                  # |z / c - 1| < 2^-7.9 over the interval; log c to half an ulp
        for z in (z0[i], np.nextafter(z0[i + 1], 0.0)):
            assert abs(LD(z) * LD(lt[i, 0]) - 1) < 2.0 ** -7.9, i
        assert abs(LD(lt[i, 1]) + np.log(LD(lt[i, 0]))) <= 0.51 * np.spacing(abs(lt[i, 1])) + 1e-300
    rng = np.random.default_rng(0)

    def ulps(got, want):
        return np.abs((got.astype(LD) - want) / np.spacing(np.abs(want.astype(np.float64))).astype(LD)).astype(np.float64)

    # exp: x = (64 m + j) ln 2 / 64 + r
    x = np.concatenate((rng.uniform(-700, 700, 200000), rng.uniform(-1e-3, 1e-3, 50000)))
    kd = np.rint(x * 92.332482616893657)
    r = ((x.astype(LD) - kd.astype(LD) * LD(1.08304246932675596e-02)) - kd.astype(LD) * LD(2.98158582698529328e-12)).astype(np.float64)
    k = kd.astype(np.int64)
    p = r * (1 + r * (0.5 + r * (1 / 6 + r * (1 / 24 + r * (1 / 120 + r / 720)))))
    sv = et[k & 63]
    got = np.ldexp((sv.astype(LD) * p.astype(LD) + sv.astype(LD)).astype(np.float64), (k >> 6).astype(np.int32))
    assert ulps(got, np.exp(x.astype(LD))).max() < 2.0
    # log: x = 2^k z, z in [OFF, 2 OFF)
    x = np.concatenate((np.exp(rng.uniform(np.log(1e-300), np.log(1e300), 200000)), rng.uniform(0.5, 2.0, 100000), 1.0 + rng.uniform(-1e-6, 1e-6, 20000)))
    bits = x.view(np.uint64)
    hi = (bits >> np.uint64(32)).astype(np.int64)
    tmp = hi - hi0
    i = (tmp >> 13) & 127
    kk = tmp >> 20
    z = (((hi - (kk << 20)).astype(np.uint64) << np.uint64(32)) | (bits & np.uint64(0xFFFFFFFF))).view(np.float64)
    invc, logc = lt[i, 0], lt[i, 1]
    rr = (z.astype(LD) * invc.astype(LD) - 1).astype(np.float64)
    kd = kk.astype(np.float64)
    w = (kd.astype(LD) * LD(0.69314718055989033) + logc.astype(LD)).astype(np.float64)
    q = ((((rr / 7 - 1 / 6) * rr + 1 / 5) * rr - 1 / 4) * rr + 1 / 3) * rr - 0.5
    hi2 = w + rr
    lo = (w - hi2) + rr + kd * 5.49792301870837116e-14
    got = (q.astype(LD) * (rr * rr).astype(LD) + lo.astype(LD)).astype(np.float64) + hi2
    assert ulps(got, np.log(x.astype(LD))).max() < 2.0


@pytest.mark.parametrize('S', [171, 256, 512, 513, 1024])
def test_the_data_radii_spline_reads_a_bounded_part_of_the_row(lib, S):
    """DESIGN 4.1 on the CPU: the evaluation matrix E of the not-a-knot spline through the extracted row (joxsz_funcs.py:476), as the
    library builds it (jxt::nak_eval_matrix), against scipy's interp1d on unit vectors; its columns decay by about 2 - sqrt(3)
    per knot beyond the last data radius, so that with CL J1226.9+3332's 19 radii (<= 116 arcsec = pixel 58 at 2 arcsec) no
    column beyond 96 carries a weight above 1e-22 of the largest: what the matrix-core product need not compute.  A row element
    dropped at that level changes g(r) by less than 1e-20 of the row's largest entry -- four orders under fp64's rounding."""
    from scipy.interpolate import interp1d
    c = S // 2
    xk = 2.0 * (np.arange(S) - c)[c:].astype(np.float64)
    q = 3.136 + 6.273 * np.arange(19)
    nrow = xk.size
    E = np.zeros((q.size, nrow))
    assert lib.jxt_nak_eval_matrix(_p(xk), nrow, _p(q), q.size, _p(E)) == 0
    ref = np.zeros_like(E)
    for k in range(0, nrow, max(1, nrow // 40)):
        e = np.zeros(nrow); e[k] = 1.0
        ref[:, k] = interp1d(xk, e, 'cubic', fill_value='extrapolate')(q)
        assert np.abs(E[:, k] - ref[:, k]).max() < 1e-12
    m = np.abs(E).max(axis=0)
    kuse = int(np.max(np.nonzero(m > 1e-22 * m.max())[0])) + 1
    assert kuse <= min(nrow, 96)
    if nrow > 120:
        k0 = 70
        ratio = m[k0 + 20] / m[k0]
        assert abs(ratio ** (1 / 20.0) - (2 - np.sqrt(3))) < 0.01          # the cardinal functions' decay per knot
        assert np.abs(E[:, 96:]).sum(axis=1).max() < 1e-20 * m.max()


# ---------------------------------------------------------------------------------------------------------------------
# exact form (round 5, csrc/jx_exact.hpp): the extracted row as one constant operator on the spline ordinates
# ---------------------------------------------------------------------------------------------------------------------
def _exact_operator(lib, pb):
    S, N = pb.S, pb.N
    nrow = S - S // 2
    Wy = np.zeros((nrow, N))
    nk = lib.jxt_exact_row_operator(_p(np.ascontiguousarray(pb.d_mat)), _p(np.ascontiguousarray(pb.beam_2d)), int(pb.beam_2d.shape[0]),
                                    ctypes.c_double(pb.step ** 2), _p(np.ascontiguousarray(pb.filtering)), S, _p(np.ascontiguousarray(pb.r_pp)), N, _p(Wy))
    return Wy, nk


@pytest.mark.parametrize('S,N,kw', [(31, 40, {}), (32, 40, {}), (64, 80, dict(fwhm=8.5, step=6.)), (65, 80, {}), (171, 313, {})])
def test_exact_row_operator_against_the_oracle(lib, S, N, kw):
    """map_out[S//2, S//2:] == Wy @ y for the oracle's own chain of joxsz_funcs.py:460-467 (interp1d 'cubic' -> f(d_mat) -> fftconvolve
    -> fft2 * filtering -> ifft2 -> central row): odd and even sides, several beams, random walkers -- to rounding (1e-13 of the row's
    maximum), with no term dropped: the operator is exact for every pixel and every radius.  The columns beyond the map's corner plus
    the spline's band are exactly zero (that many ordinates are never computed)."""
    from joxsz_amd import datasets
    from oracle import joxsz_oracle as orc
    pb = datasets.synthetic_problem(S=S, N=N, seed=S, **kw)
    Wy, nk = _exact_operator(lib, pb)
    assert 0 < nk <= N and np.all(Wy[:, nk:] == 0.0)
    corner = np.sqrt(2.0) * np.abs(pb.d_mat).max() / np.sqrt(2.0)         # (the largest pixel radius IS the corner's)
    if pb.r_pp[-1] > corner:
        assert nk < N or N - np.searchsorted(pb.r_pp, corner) < 60
    th = datasets.walker_ball(pb, 5, spread=0.06, seed=S)
    for t in th:
        pp = orc.press_fun(orc.pars_dict(pb, t), pb.r_pp)
        out = orc.row_chain(pb, pp)
        got = Wy @ out['y']
        assert np.max(np.abs(got - out['map_row'])) <= 1e-13 * np.max(np.abs(out['map_row']))
    # linear in the ordinates with constant coefficients: any y, not only physical profiles
    y = np.random.default_rng(S).normal(size=N)
    f = interp1d(np.append(-pb.r_pp, pb.r_pp), np.append(y, y), 'cubic', bounds_error=False, fill_value=(0., 0.))
    conv = fftconvolve(f(pb.d_mat), pb.beam_2d, 'same') * pb.step ** 2
    row = np.real(ifft2(fft2(conv) * pb.filtering))[S // 2, S // 2:]
    assert np.max(np.abs(Wy @ y - row)) <= 1e-12 * np.max(np.abs(row))


def test_exact_row_operator_with_the_measured_beam_and_transfer_function(lib):
    """The reference's default inputs (measured beam profile: an image that is not separable; measured transfer function: rough from
    one wavenumber to the next; joxsz_main.py:59-60) go through the same builder: nothing in it depends on smoothness or rank."""
    from joxsz_amd import datasets, setup_host as sh
    from oracle import joxsz_oracle as orc
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'bundled_inputs.npz'))
    pb = datasets.synthetic_problem(S=129, N=160, seed=9)
    prof = sh.clip_beam_profile(z['beam_r'], z['beam_prof'])
    beam_2d, _ = sh.beam_image(2., 116.0, approx=False, profile=prof)
    wn, tf = sh.transfer_function(z['wn_as'], z['tf'], approx=False)
    pb.beam_2d = np.ascontiguousarray(beam_2d)
    pb.filtering = np.ascontiguousarray(sh.filter_image(wn, tf, 129, 2.))
    pb = pb.validate()
    Wy, nk = _exact_operator(lib, pb)
    assert nk > 0
    pp = orc.press_fun(orc.pars_dict(pb, datasets.fiducial_theta(pb)), pb.r_pp)
    out = orc.row_chain(pb, pp)
    assert np.max(np.abs(Wy @ out['y'] - out['map_row'])) <= 1e-13 * np.max(np.abs(out['map_row']))


def test_exact_form_operand_layouts(lib):
    """The two operand layouts of jx_ordrow_kernel, read back the way the matrix cores consume them (v_mfma_f64_16x16x4: K slot lk of
    sub-step e of macro step s = index 16 s + 4 lk + e on BOTH operands): the ordinate operator reproduces y_scale * A (upper
    triangular: tile t has no entry before step t) and the row operator reproduces Wy, zero padded."""
    r = np.ascontiguousarray(GRIDS['arange'][:100])
    N = r.size
    A = np.zeros((N, N))
    lib.jxt_abel_matrix(_p(r), N, _p(A))
    ysc = 3.7
    nS, nSj = 5, (N + 15) // 16                                      # 80 ordinates of 100 radii
    Typ = np.zeros((nSj, nS, 64, 4))
    lib.jxt_abel_ordinate_layout(_p(r), N, ctypes.c_double(ysc), nS, nSj, _p(Typ))
    lane = np.arange(64)
    li, lk = lane & 15, lane >> 4
    for s in range(nSj):
        for t in range(nS):
            k = 16 * t + li[:, None]
            j = 16 * s + 4 * lk[:, None] + np.arange(4)[None, :]
            want = np.where((j < N) & (k <= j), ysc * A[np.minimum(k, N - 1), np.minimum(j, N - 1)], 0.0)
            np.testing.assert_array_equal(Typ[s, t], want)
            if t > s:
                assert np.all(Typ[s, t] == 0.0)
    nrow, NXT, ng = 37, 2, 2
    Wy = np.random.default_rng(1).normal(size=(nrow, N))
    Opk = np.zeros((ng, nSj, 4, 64, NXT))
    lib.jxt_exact_row_layout(_p(Wy), nrow, N, nSj, NXT, ng, _p(Opk))
    for g in range(ng):
        for s in range(nSj):
            for e in range(4):
                for t in range(NXT):
                    x, i = 16 * (g * NXT + t) + li, 16 * s + 4 * lk + e
                    want = np.where((x < nrow) & (i < N), Wy[np.minimum(x, nrow - 1), np.minimum(i, N - 1)], 0.0)
                    np.testing.assert_array_equal(Opk[g, s, e, :, t], want)


def test_folded_last_ordinate_tile(lib):
    """An odd number of ordinate tiles: the last tile's ordinates enter the row only through Wy, so their share of the row is a constant operator
    on the profile, Wf = Wy[:, tile] (y_scale A)[tile, :] (jxt::exact_fold_layout, what jx_ordrow_kernel's row product adds in the timed path).
    Entry by entry in the row operator's layout against numpy in long double, and the identity itself on a random profile."""
    r = np.ascontiguousarray(GRIDS['arange'][:100])
    N = r.size
    A = np.zeros((N, N))
    lib.jxt_abel_matrix(_p(r), N, _p(A))
    ysc, nS, nSj, nrow, NXT, ng = 2.9, 5, (N + 15) // 16, 37, 2, 2     # 80 ordinates (5 tiles) of 100 radii (7 macro steps): tile 4 folded, its k-range = steps 4..6
    rng = np.random.default_rng(2)
    Wy = np.zeros((nrow, N))
    Wy[:, :16 * nS] = rng.normal(size=(nrow, 16 * nS))
    s0, nSf = nS - 1, nSj - (nS - 1)
    out = np.zeros((ng, nSf, 4, 64, NXT))
    lib.jxt_exact_fold_layout(_p(Wy), nrow, _p(r), N, ctypes.c_double(ysc), nS, nSj, NXT, ng, _p(out))
    T = np.triu(ysc * A.astype(np.longdouble))                          # y = T pp
    Wf = (Wy[:, 16 * s0:16 * s0 + 16].astype(np.longdouble) @ T[16 * s0:16 * s0 + 16, :]).astype(np.float64)   # [nrow][N], zero before 16 s0
    assert np.all(Wf[:, :16 * s0] == 0.0)
    lane = np.arange(64)
    li, lk = lane & 15, lane >> 4
    for g in range(ng):
        for sf in range(nSf):
            for e in range(4):
                for t in range(NXT):
                    x, k = 16 * (g * NXT + t) + li, 16 * (s0 + sf) + 4 * lk + e
                    want = np.where((x < nrow) & (k < N), Wf[np.minimum(x, nrow - 1), np.minimum(k, N - 1)], 0.0)
                    np.testing.assert_allclose(out[g, sf, e, :, t], want, rtol=1e-15, atol=1e-300)
    pp = rng.normal(size=N)
    y = (T @ pp.astype(np.longdouble)).astype(np.float64)
    np.testing.assert_allclose(Wf @ pp + Wy[:, :16 * s0] @ y[:16 * s0], Wy @ y, rtol=1e-12, atol=1e-12 * np.abs(Wy @ y).max())


def _stockham(x, radices, sign):
    """The passes of csrc/jx_fft.hpp restated in numpy: in place, every butterfly of a pass read before any is written; butterfly j of a
    pass of radix R behind sub-transforms of size ns reads x[j + t n/R], multiplies by root^(t (j mod ns) n/(ns R)), transforms the R values and
    writes them to y[(j - j mod ns) R + j mod ns + t ns]."""
    n = x.size
    root = np.exp(sign * 2j * np.pi * np.arange(n) / n)
    ns = 1
    for R in radices:
        nb, tstep = n // R, n // (ns * R)
        j = np.arange(nb)
        k = j % ns
        v = np.stack([x[j + t * nb] * root[(t * k * tstep) % n] for t in range(R)])            # [R][nb]
        v = np.exp(sign * 2j * np.pi * np.outer(np.arange(R), np.arange(R)) / R) @ v            # the R-point transform
        y = np.empty_like(x)
        for t in range(R):
            y[(j - k) * R + k + t * ns] = v[t]
        x, ns = y, ns * R
    return x


def test_fft_radices_and_the_stockham_passes_of_the_literal_route(lib):
    """jx_fft.hpp's host side: every 2^a 3^b 5^c length up to 1280 splits into at most four passes of the radices the kernels have (<= 10 up to
    640: the 128-register kernels; 12 and 16 beyond), other lengths are refused; the pass formulas with those radices are the DFT."""
    rng = np.random.default_rng(5)
    smooth = [n for n in range(2, 1281) if max((p for p in range(2, n + 1) if n % p == 0 and all(p % q for q in range(2, p))), default=1) <= 5]
    assert 540 in smooth and 512 in smooth and 1080 in smooth
    for n in range(2, 1281):
        rx = (ctypes.c_int * 12)()
        npass = lib.jxt_fft_radices(n, rx)
        if n not in smooth:
            assert npass == 0, n
            continue
        r = list(rx[:npass])
        assert 1 <= npass <= 4 and int(np.prod(r)) == n, (n, r)
        assert set(r) <= ({2, 3, 4, 5, 6, 8, 9, 10} if n <= 640 else {2, 3, 4, 5, 6, 8, 9, 10, 12, 16}), (n, r)
        assert r == sorted(r, reverse=True), (n, r)                      # largest first: the pass without twiddles is the widest
        nu = next(v for v in (4, 8, 9, 10, 16, 17, 20) if v >= -(-n // 64))        # the kernel instance of this length (rows per lane, rounded up)
        assert all(-(-(n // R) // 64) <= -(-nu // R) for R in r), (n, r)               # butterflies per lane of a pass fit its register array
    assert list((lambda a: (lib.jxt_fft_radices(540, a), a[:3])[1])((ctypes.c_int * 12)())) == [10, 9, 6]
    assert list((lambda a: (lib.jxt_fft_radices(512, a), a[:3])[1])((ctypes.c_int * 12)())) == [8, 8, 8]
    for n in (12, 96, 160, 288, 512, 540, 1024, 1080, 75, 135, 243):
        rx = (ctypes.c_int * 12)()
        r = list(rx[:lib.jxt_fft_radices(n, rx)])
        x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
        np.testing.assert_allclose(_stockham(x.copy(), r, -1), np.fft.fft(x), rtol=0, atol=1e-12 * n)
        np.testing.assert_allclose(_stockham(x.copy(), r, +1), np.fft.ifft(x) * n, rtol=0, atol=1e-12 * n)
