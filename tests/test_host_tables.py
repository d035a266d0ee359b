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


@pytest.mark.parametrize('S,B,P', [(32, 9, 36), (64, 55, 96), (48, 11, 96)])
def test_mixed_domain_tables(lib, S, B, P):
    """Pass structure of jx_conv.hpp in numpy, fed with the C-built tables, against scipy."""
    from oracle.joxsz_oracle import centdistmat
    rng = np.random.default_rng(S)
    o, Ph, Sh, c = (B - 1) // 2, P // 2 + 1, S // 2 + 1, S // 2
    ax = np.arange(B) - o
    beam = np.exp(-centdistmat(ax) ** 2 / (2 * (B / 5.) ** 2)) * (1 + 0.1 * np.cos(centdistmat(ax)))
    y2d = rng.standard_normal((S, S))
    axf = np.linspace(-S // 2 + 1, S // 2, S)
    filt = np.roll(1 - np.exp(-np.sqrt(axf ** 2 + axf[:, None] ** 2) / 4.), S // 2 + 1, axis=(0, 1))
    taps = np.zeros((o + 1, Ph))
    assert lib.jxt_beam_fir_taps(_p(np.ascontiguousarray(beam)), B, P, ctypes.c_double(4.0 / P), _p(taps)) == 0
    Hy = np.zeros((S, Sh, 2))
    lib.jxt_tf_hy_table(_p(np.ascontiguousarray(filt)), S, _p(Hy))
    Hy = Hy[..., 0] + 1j * Hy[..., 1]
    Y = np.fft.rfft(np.pad(y2d, ((0, 0), (0, P - S))), axis=1)
    C = np.zeros((S, Ph), complex)
    for r in range(S):
        for m in range(max(0, r - o), min(S, r + o + 1)):
            C[r] += taps[abs(r - m)] * Y[m]
    conv = (np.fft.irfft(C, P, axis=1) * P)[:, :S]
    want = fftconvolve(y2d, beam, 'same') * 4.0
    np.testing.assert_allclose(conv, want, rtol=0, atol=1e-12 * np.abs(want).max())
    Z = (np.fft.rfft(conv, axis=1) * Hy).sum(0)
    row = np.array([np.real(Z * np.exp(2j * np.pi * np.arange(Sh) * x / S)).sum() for x in range(c, S)])
    want_row = np.real(ifft2(fft2(want) * filt))[c, c:]
    np.testing.assert_allclose(row, want_row, rtol=0, atol=1e-12 * np.abs(want_row).max())
    # an asymmetric beam is refused (the rocFFT path handles it)
    bad = beam.copy(); bad[0, 1] *= 1.5
    assert lib.jxt_beam_fir_taps(_p(bad), B, P, ctypes.c_double(1.0), _p(taps)) == -1
    assert lib.jxt_custom_conv_lp(512, 27) == 288 and lib.jxt_custom_conv_lp(171, 27) == 0


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


def test_constexpr_trig(lib):
    lib.jxt_cx_cos2pi.restype = ctypes.c_double
    lib.jxt_cx_sin2pi.restype = ctypes.c_double
    lib.jxt_cx_cos2pi.argtypes = lib.jxt_cx_sin2pi.argtypes = [ctypes.c_longlong, ctypes.c_longlong]
    pi = np.longdouble('3.14159265358979323846264338327950288')
    for n in (3, 4, 7, 16, 18, 288, 576, 1152):
        for k in list(range(-3, n + 3)) if n < 300 else (0, 1, n // 8, n // 4, n // 3, n // 2, n - 1, 5 * n + 7):
            t = 2 * pi * np.longdouble(k % n) / np.longdouble(n)          # 80-bit reference
            assert abs(lib.jxt_cx_cos2pi(k, n) - float(np.cos(t))) < 2.3e-16, (k, n)
            assert abs(lib.jxt_cx_sin2pi(k, n) - float(np.sin(t))) < 2.3e-16, (k, n)


@pytest.mark.parametrize('n', [2, 3, 4, 6, 8, 9, 12, 16, 18, 24, 27, 32])
def test_register_fft(lib, n):
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n) + 1j * rng.standard_normal(n)
    for inv, want in ((0, np.fft.fft(x)), (1, np.fft.ifft(x) * n)):
        re, im = np.ascontiguousarray(x.real), np.ascontiguousarray(x.imag)
        assert lib.jxt_regfft(n, inv, _p(re), _p(im)) == 0
        np.testing.assert_allclose(re + 1j * im, want, rtol=0, atol=2e-15 * n * np.abs(x).max())


@pytest.mark.parametrize('S,o,mirror', [(31, 4, 1), (32, 4, 1), (64, 27, 1), (32, 4, 0), (171, 27, 1)])
def test_conv_row_tables(lib, S, o, mirror):
    """Jobs computed once must reproduce every conv row of a FIR over mirrored map rows."""
    IP = ctypes.POINTER(ctypes.c_int)
    umap, urow, jrow, rowjob = (np.zeros(S, np.int32) for _ in range(4))
    seg, cnt = np.zeros(3 * S, np.int32), np.zeros(3, np.int32)
    lib.jxt_conv_row_tables(S, o, mirror, *[a.ctypes.data_as(IP) for a in (umap, urow, jrow, rowjob, seg, cnt)])
    NU, NJ, nseg = (int(v) for v in cnt)
    rng = np.random.default_rng(S)
    c = S // 2
    if mirror:
        base = rng.standard_normal(S)                 # one value per |m - c|
        Y = np.array([base[abs(m - c)] for m in range(S)])
        assert NU == max(c, S - 1 - c) + 1
    else:
        Y = rng.standard_normal(S)
        assert NU == S and NJ == S and nseg == 1
    np.testing.assert_array_equal(Y[urow[:NU]][umap], Y)             # distinct rows carry every row
    tap = rng.standard_normal(o + 1)
    conv = np.array([sum(tap[abs(r - m)] * Y[m] for m in range(max(0, r - o), min(S, r + o + 1))) for r in range(S)])
    np.testing.assert_allclose(conv[jrow[:NJ]][rowjob], conv, rtol=1e-13, atol=1e-13)
    segs = seg[:3 * nseg].reshape(-1, 3)
    rows = np.concatenate([np.arange(a, a + n) for a, n, _ in segs])
    np.testing.assert_array_equal(rows, jrow[:NJ])
    assert all(q == sum(n for _, n, _ in segs[:i]) for i, (_, _, q) in enumerate(segs))
    if mirror and S % 2 == 1:
        assert NJ == c + 1 and nseg == 1                              # odd side: exact mirror, half the rows


def test_lowrank_factor(lib):
    """Jacobi SVD of the host tables against numpy: singular values, reconstruction at the chosen rank."""
    rng = np.random.default_rng(5)
    m, n, true_rank = 60, 45, 9
    A = rng.standard_normal((m, true_rank)) @ np.diag(10.0 ** -np.arange(true_rank)) @ rng.standard_normal((true_rank, n))
    A += 1e-13 * rng.standard_normal((m, n))
    L = np.zeros((n, m)); Rt = np.zeros((n, n)); sg = np.zeros(n)
    lib.jxt_lowrank_factor.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double] + [ctypes.c_void_p] * 3
    r = lib.jxt_lowrank_factor(_p(np.ascontiguousarray(A)), m, n, 1e-10, _p(L), _p(Rt), _p(sg))
    want = np.linalg.svd(A, compute_uv=False)
    assert r == int((want > 1e-10 * want[0]).sum()) == true_rank
    np.testing.assert_allclose(sg[:true_rank], want[:true_rank], rtol=1e-12, atol=1e-14 * want[0])
    approx = L[:r].T @ Rt[:r]
    assert np.abs(approx - A).max() < 1e-9 * np.abs(A).max()
    np.testing.assert_allclose(np.linalg.norm(Rt[:r], axis=1), 1.0, rtol=1e-12)
    # full rank request reproduces the matrix to rounding
    r2 = lib.jxt_lowrank_factor(_p(np.ascontiguousarray(A)), m, n, 0.0, _p(L), _p(Rt), _p(sg))
    assert r2 == n
    assert np.abs(L[:r2].T @ Rt[:r2] - A).max() < 1e-13 * np.abs(A).max()


def test_fused_row_operator(lib):
    """FIR along rows + job combination as one matrix per column, against the two steps done one after the other."""
    S, o, r, nb = 32, 4, 5, 7
    IP = ctypes.POINTER(ctypes.c_int)
    umap = np.zeros(S, np.int32); urow = np.zeros(S, np.int32); jrow = np.zeros(S, np.int32); rowjob = np.zeros(S, np.int32)
    seg = np.zeros(3 * S, np.int32); cnt = np.zeros(3, np.int32)
    lib.jxt_conv_row_tables(S, o, 1, *[a.ctypes.data_as(IP) for a in (umap, urow, jrow, rowjob, seg, cnt)])
    NU, NJ = int(cnt[0]), int(cnt[1])
    rng = np.random.default_rng(2)
    U = rng.standard_normal((r, NJ)); coef = rng.standard_normal((o + 1, nb)); R = rng.standard_normal((NU, nb))
    RP, KU = 16, NU + 3
    out = np.zeros((nb, RP, KU))
    assert lib.jxt_fused_row_operator(_p(U), r, S, o, 1, _p(coef), nb, nb, RP, KU, _p(out)) == NJ
    # step by step: FIR per job, then the combination
    C = np.zeros((NJ, nb))
    for q in range(NJ):
        rq = jrow[q]
        for m in range(max(0, rq - o), min(S - 1, rq + o) + 1):
            C[q] += coef[abs(rq - m)] * R[umap[m]]
    want = U @ C                                             # [r][nb]
    got = np.einsum('bru,ub->rb', out[:, :r, :NU], R)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-13)
    assert not out[:, r:].any() and not out[:, :, NU:].any()


# ---------------------------------------------------------------------------------------
# tables of this round's kernels: rows evaluated from the spline + quarter-length transform (jx_dct.hpp), the
# spline-array operator (jx_abel_gemm_kernel), the real-space kernels of the odd-side route
# ---------------------------------------------------------------------------------------
IP = ctypes.POINTER(ctypes.c_int)


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


@pytest.mark.parametrize('S,LP', [(64, 48), (65, 48), (171, 144), (128, 96)])
def test_row_sample_tables_reproduce_the_mirrored_spline(lib, S, LP):
    """f(|x|) = A y_k + B y_{k+1} + C M_k + D M_{k+1} with the table's (interval, weights) per (row, sample) equals
    interp1d((-r, r), (y, y), 'cubic', fill_value=(0, 0)) of joxsz_funcs.py:460-462 at the pixel radii of the quadrant."""
    step, kpc_as = 2.0, 8.0012
    h = step * kpc_as
    N = int(0.8 * S)                                             # the grid ends inside the map: fill value 0 beyond it
    r = h * np.arange(1, N + 1)
    c = S // 2
    a = np.arange(c + 1) * h
    Qrad = np.ascontiguousarray(np.hypot(a[:, None], a[None, :]))      # centdistmat quadrant (joxsz_funcs.py:78-88)
    t = _dct_tables(lib, Qrad, r, S, LP)
    assert t is not None and t['has_x0'] == (S % 2 == 0) and t['amax'] == (c - 1 if S % 2 == 0 else c)
    rng = np.random.default_rng(S)
    y = np.exp(-(r / (12 * h)) ** 2) * (1 + 0.05 * rng.standard_normal(N))
    G, M = _mirrored_moments(lib, r, y)
    ym = np.zeros(2 * (N + 2))
    ym[0:2 * N:2], ym[1:2 * N:2] = y, M
    f = interp1d(np.concatenate((-r[::-1], r)), np.concatenate((y[::-1], y)), 'cubic', bounds_error=False, fill_value=(0., 0.))
    for u in (0, 1, c // 3, c - 1, c):
        k = t['dk'][u, :t['amax'] + 1] // 16
        w = t['dw'][u, :t['amax'] + 1]
        got = w[:, 0] * ym[2 * k] + w[:, 1] * ym[2 * k + 2] + w[:, 2] * ym[2 * k + 1] + w[:, 3] * ym[2 * k + 3]
        want = f(Qrad[u, :t['amax'] + 1])
        assert np.abs(got - want).max() < 1e-12 * np.abs(y).max(), u
        assert np.all(t['dw'][u, t['amax'] + 1:] == 0)                                  # padding: zero weights
        if t['has_x0']:
            k0, w0 = t['x0k'][u] // 16, t['x0w'][u]
            got0 = w0[0] * ym[2 * k0] + w0[1] * ym[2 * k0 + 2] + w0[2] * ym[2 * k0 + 1] + w0[3] * ym[2 * k0 + 3]
            assert abs(got0 - f(Qrad[u, c])) < 1e-12 * np.abs(y).max()


def _quarter_transform_with_tables(q, LP, pk):
    """The steps of jx_rowdct_kernel in numpy: z from groups of four samples, one complex FFT of length Q = LP/2, the
    real-even split with the kernel's own constants pk = (cos/2, -sin/2, 1/(2 sin(2 pi k/P)), 1/(2 sin(2 pi (Q-k)/P)))."""
    Q, P = LP // 2, 2 * LP
    amax = len(q) - 1
    xq = lambda n: q[abs(n)] if abs(n) <= amax else 0.0
    z = np.zeros(Q, complex)
    for g in range(Q // 2 + 1):
        if g <= (Q - 1) // 2:
            z[g] = (xq(4 * g) + xq(4 * g + 1) - xq(4 * g - 1)) + 1j * (xq(4 * g + 2) + xq(4 * g + 3) - xq(4 * g + 1))
        if g >= 1 and Q - g > (Q - 1) // 2:
            z[Q - g] = (xq(4 * g) - xq(4 * g + 1) + xq(4 * g - 1)) + 1j * (xq(4 * g - 2) - xq(4 * g - 1) + xq(4 * g - 3))
    Z = np.fft.fft(z)
    R = np.zeros(LP + 1)
    b0 = 2.0 * q[1::2].sum()
    for k in range(Q // 2 + 1):
        zk, zq = Z[k], Z[0 if k == 0 else Q - k]
        sx, sy, dx, dy = zk.real + zq.real, zk.imag - zq.imag, zk.real - zq.real, zk.imag + zq.imag
        tx, ty = pk[k, 0] * dx - pk[k, 1] * dy, pk[k, 0] * dy + pk[k, 1] * dx
        Ak, Aq, Ik, Iq = 0.5 * sx + ty, 0.5 * sx - ty, 0.5 * sy - tx, -0.5 * sy - tx
        bk, bq = (b0 if k == 0 else pk[k, 2] * Ik), pk[k, 3] * Iq
        R[k], R[LP - k] = Ak + bk, Ak - bk
        if 2 * k != Q:
            R[Q - k] = Aq + bq
            if k > 0:
                R[Q + k] = Aq - bq
    return R


@pytest.mark.parametrize('S,LP', [(64, 48), (128, 96), (512, 288), (171, 144)])
def test_quarter_length_real_even_transform_with_the_kernel_constants(lib, S, LP):
    """R(k) = sum_n x[n] cos(2 pi k n / P) of an even row through one complex FFT of length P/4 (Cooley, Lewis & Welch),
    with the split constants the library hands the kernel, against the direct cosine sum."""
    c = S // 2
    h = 16.0
    a = np.arange(c + 1) * h
    Qrad = np.ascontiguousarray(np.hypot(a[:, None], a[None, :]))
    r = h * np.arange(1, int(0.9 * S) + 1)
    t = _dct_tables(lib, Qrad, r, S, LP)
    assert t is not None
    rng = np.random.default_rng(LP)
    amax = t['amax']
    q = np.exp(-(np.arange(amax + 1) / (0.3 * amax)) ** 2) + 0.01 * rng.standard_normal(amax + 1)
    n = np.arange(-amax, amax + 1)
    want = (q[np.abs(n)][None, :] * np.cos(2 * np.pi * np.arange(LP + 1)[:, None] * n[None, :] / (2 * LP))).sum(1)
    got = _quarter_transform_with_tables(q, LP, t['pk'])
    assert np.abs(got - want).max() < 1e-13 * np.abs(want).max()


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


@pytest.mark.parametrize('S', [31, 65, 171])
def test_odd_rowspace_operator(lib, S):
    """K[rho][b][a] of the odd-side route: sum_a K[b][a] cc[c + a] equals the circular convolution of the symmetric row cc
    with k_rho[j] = sum_kc v[kc] cos(2 pi kc j / S), read at columns c + b (joxsz_funcs.py:466-467, 472 in real space)."""
    c, Sh, nout = S // 2, S // 2 + 1, S // 2 + 1
    rng = np.random.default_rng(S)
    r = 3
    V = rng.standard_normal((r, Sh))
    KQ = (nout + 3) // 4 * 4
    nmg = lib.jxt_odd_rowspace_operator(_p(V), r, S, KQ, ctypes.POINTER(ctypes.c_double)())
    out = np.zeros((nmg, r, 64, KQ))
    assert lib.jxt_odd_rowspace_operator(_p(V), r, S, KQ, _p(out)) == nmg == (nout + 63) // 64
    half = rng.standard_normal(nout)
    cc = np.concatenate((half[:0:-1], half))                            # symmetric about the centre column c, length S
    j = np.arange(S)
    for rho in range(r):
        k = (V[rho][None, :] * np.cos(2 * np.pi * np.arange(Sh)[None, :] * j[:, None] / S)).sum(1)
        want = np.array([sum(k[(c + b - x) % S] * cc[x] for x in range(S)) for b in range(nout)])
        Kb = np.concatenate([out[mg, rho] for mg in range(nmg)])[:nout, :nout]
        got = Kb @ half
        assert np.abs(got - want).max() < 1e-12 * np.abs(want).max()


def test_odd_padded_lengths(lib):
    for S, o, want in ((171, 27, 144), (31, 4, 48), (513, 27, 288), (1025, 27, 576), (65, 27, 48), (512, 27, 0), (2001, 27, 0)):
        assert lib.jxt_custom_conv_lp_odd(S, o) == want
