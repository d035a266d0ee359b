"""Cross-checks of the parts of the oracle that the reference cannot pin (PyAbel and
mbproj2 are absent): analytic known answers and structural identities."""
import math

import numpy as np
from scipy.special import gamma

from oracle import pyabel_direct, mbproj2_parts as mbp, joxsz_oracle as orc


def test_abel_gaussian_known_answer():
    # A[exp(-r^2/s^2)](y) = s sqrt(pi) exp(-y^2/s^2); the discretisation (trapezoid + linear
    # end cell, truncated at r_N) agrees to its own error, not to rounding
    r = 2.0 * np.arange(1, 1200)
    s = 180.0
    got = pyabel_direct.direct_transform_forward(np.exp(-(r / s) ** 2), r)
    want = s * math.sqrt(math.pi) * np.exp(-(r / s) ** 2)
    # the trapezoid next to the inverse-square-root singularity converges like sqrt(h): ~2e-3 at h=2
    assert np.max(np.abs(got - want)) / want.max() < 3e-3
    r2 = 0.5 * np.arange(1, 4800)
    got2 = pyabel_direct.direct_transform_forward(np.exp(-(r2 / s) ** 2), r2)
    want2 = s * math.sqrt(math.pi) * np.exp(-(r2 / s) ** 2)
    assert np.max(np.abs(got2 - want2)) / want2.max() < 0.55 * np.max(np.abs(got - want)) / want.max()


def test_abel_beta_model_known_answer():
    # gNFW with a=2, c=0: P0 (1+x^2)^(-b/2) -> sqrt(pi) Gamma((b-1)/2)/Gamma(b/2) r_p P0 (1+y^2/r_p^2)^((1-b)/2)
    r = 4.0 * np.arange(1, 3000)
    rp, b, P0 = 300., 5.0, 0.3
    p = {'P_0': P0, 'r_p': rp, 'a': 2., 'b': b, 'c': 0.}
    got = pyabel_direct.direct_transform_forward(orc.press_fun(p, r), r)
    want = math.sqrt(math.pi) * gamma((b - 1) / 2) / gamma(b / 2) * rp * P0 * (1 + (r / rp) ** 2) ** ((1 - b) / 2)
    sel = r < 2000.
    assert np.max(np.abs(got[sel] - want[sel])) / want.max() < 5e-3


def test_abel_is_linear_and_last_point_zero():
    r = 16.0024 * np.arange(1, 80)
    rng = np.random.default_rng(0)
    f, g = rng.random(r.size), rng.random(r.size)
    T = pyabel_direct.direct_transform_forward
    np.testing.assert_allclose(T(2 * f + 3 * g, r), 2 * T(f, r) + 3 * T(g, r), rtol=1e-13)
    assert T(f, r)[-1] == 0.0
    A = pyabel_direct.abel_weight_matrix(r)
    np.testing.assert_allclose(A @ f, T(f, r), rtol=1e-13)


def test_uniform_vs_nonuniform_trapezoid_branches_agree():
    # the arange grid of joxsz_main.py:104 fails PyAbel's 1e-13 uniformity test (ulp noise at
    # r ~ 5000); both branches must give the same numbers to rounding
    r_exact = 16.0 * np.arange(1, 314)                   # exactly uniform
    r_noisy = np.arange(2 * 8.0012, 5000. + 2 * 8.0012, 2 * 8.0012)
    assert pyabel_direct.is_uniform_sampling(r_exact)
    f = 1. / (1. + (r_noisy / 300.) ** 2) ** 2
    a = pyabel_direct.direct_transform_forward(f, r_noisy)
    rr = r_noisy.copy()
    b = pyabel_direct._trapz(np.ones((1, rr.size)), rr, False)
    assert abs(b[0] - (rr[-1] - rr[0])) < 1e-9
    assert np.all(np.isfinite(a)) and a[0] > a[10] > a[100] > 0


def test_sz_path_is_linear_in_pressure(golden_tiny):
    """Everything after pp is linear: bright(P_0 = 2x) = 2 bright when the T profile is held."""
    pb, ref = golden_tiny
    p = orc.pars_dict(pb, ref['thetas'][0])
    st1 = orc.sz_stages(pb, p)
    p2 = dict(p); p2['P_0'] *= 2
    st2 = orc.sz_stages(pb, p2)
    np.testing.assert_allclose(st2['map_row'], 2 * st1['map_row'], rtol=1e-11, atol=1e-20)
    np.testing.assert_allclose(st2['y_2d'], 2 * st1['y_2d'], rtol=1e-13)


def test_priors_and_cash():
    assert mbp.param_prior(0.5, 0., 1.) == 0.0
    assert mbp.param_prior(1.0, 0., 1.) == 0.0            # bounds are inclusive
    assert mbp.param_prior(1.0000001, 0., 1.) == -np.inf
    g = mbp.param_gaussian_prior(1.07, 1.0, 0.07)
    assert abs(g - (-0.5 * math.log(2 * math.pi) - math.log(0.07) - 0.5)) < 1e-14
    assert mbp.param_gaussian_prior(1.0, 1.0, 0.0) == -np.inf
    d, m = np.array([3., 0., 7.]), np.array([2.5, 0.4, 6.0])
    want = 3 * math.log(2.5) - 2.5 + 0 - 0.4 + 7 * math.log(6.0) - 6.0
    assert abs(mbp.cash_log_likelihood(d, m) - want) < 1e-13
    assert mbp.cash_log_likelihood(np.array([1.0]), np.array([0.0])) == -np.inf


def test_count_rate_interpolation_clamps():
    lnT = np.linspace(math.log(0.06), math.log(60.), 100)
    z0 = -150. + 0.5 * lnT
    z1 = z0 + 0.3
    lo = mbp.count_rate(lnT, z0, z1, np.array([0.01]), 0.5, np.array([1e-2]))
    edge = mbp.count_rate(lnT, z0, z1, np.array([0.06]), 0.5, np.array([1e-2]))
    np.testing.assert_allclose(lo, edge, rtol=1e-14)
    mid = mbp.count_rate(lnT, z0, z1, np.array([3.0]), 0.0, np.array([2e-3]))
    np.testing.assert_allclose(mid, np.exp(-150. + 0.5 * math.log(3.0)) * 4e-6, rtol=1e-12)


def test_sz_row_is_linear_in_the_pressure_profile(golden_tiny):
    """funcs:457-472 between ``press_fun`` and the extracted row: linear with constant coefficients.  The operator
    tabulated from unit profiles reproduces the row of every golden parameter vector, and superposition holds."""
    pb, ref = golden_tiny
    G = orc.sz_operator(pb)
    assert G.shape == (pb.N, pb.nrow)
    rng = np.random.default_rng(0)
    for th in ref['thetas'][np.isfinite(ref['ref_logp'])][:4]:
        st = orc.sz_stages(pb, orc.pars_dict(pb, th))
        np.testing.assert_allclose(st['pp'] @ G, st['map_row'], rtol=0, atol=1e-12 * np.abs(st['map_row']).max())
    a, b = rng.random(pb.N), rng.random(pb.N)
    ra, rb, rab = (orc.row_chain(pb, v)['map_row'] for v in (a, b, 2.5 * a - 0.5 * b))
    np.testing.assert_allclose(rab, 2.5 * ra - 0.5 * rb, rtol=0, atol=1e-12 * np.abs(rab).max())


# ---- behaviours of PyAbel's direct integral that the restatement relies on (source absent: these pin the restatement to
# ---- the published algorithm case by case, they cannot pin PyAbel itself -- DESIGN.md section 3)

def test_abel_last_interior_row_keeps_a_quarter_of_the_end_sample():
    """Row i = N-2: the 'extra triangle' correction takes HALF of the two-sample trapezoid of columns {i, i+1}, and
    column i+1 = N-1 is the last sample, whose trapezoid weight is already dx/2 -> 0.25 dx stays (interior rows: 0.5 dx)."""
    r = 10.0 * np.arange(1, 12)
    n, dx = r.size, 10.0
    A = pyabel_direct.abel_weight_matrix(r)

    def cell(i):                                                   # analytic end cell, coefficient of F_{i+1}
        s = math.sqrt(r[i + 1] ** 2 - r[i] ** 2)
        return s / dx - math.acosh(r[i + 1] / r[i]) * r[i] / dx

    i = n - 2
    want_last = (0.25 * dx / math.sqrt(r[i + 1] ** 2 - r[i] ** 2) + cell(i)) * 2 * r[i + 1]
    assert abs(A[i, i + 1] - want_last) < 1e-13 * abs(want_last)
    i = 3
    want_mid = (0.5 * dx / math.sqrt(r[i + 1] ** 2 - r[i] ** 2) + cell(i)) * 2 * r[i + 1]
    assert abs(A[i, i + 1] - want_mid) < 1e-13 * abs(want_mid)
    # beyond the first off-diagonal: plain trapezoid weights dx (dx/2 for the last column)
    assert abs(A[3, 7] - dx * 2 * r[7] / math.sqrt(r[7] ** 2 - r[3] ** 2)) < 1e-12
    assert abs(A[3, n - 1] - 0.5 * dx * 2 * r[n - 1] / math.sqrt(r[n - 1] ** 2 - r[3] ** 2)) < 1e-12
    assert np.all(A[n - 1] == 0.0) and np.all(np.tril(A, -1) == 0.0)


def test_abel_grid_starting_at_zero_takes_the_cosh1_branch():
    """r[0] = 0: acosh(r_1 / r_0) is replaced by acosh(cosh(1)) = 1 (the F_0 term vanishes anyway since F = 2 r f)."""
    r = 5.0 * np.arange(0, 40)
    f = np.exp(-(r / 60.) ** 2)
    out = pyabel_direct.direct_transform_forward(f, r)
    assert np.all(np.isfinite(out)) and out[-1] == 0.0
    F = 2 * r * f
    h = 5.0
    trap = h * (F[2:-1] / r[2:-1]).sum() + 0.5 * h * F[-1] / r[-1] + 0.5 * h * F[1] / r[1]      # row 0, j >= 1, minus half the first cell
    fr = (F[1] - F[0]) / h
    want0 = trap + r[1] * fr + 1.0 * (F[0] - fr * r[0])
    assert abs(out[0] - want0) < 1e-12 * abs(want0)
    # and the known answer at the centre to the method's accuracy
    assert abs(out[0] - 60. * math.sqrt(math.pi)) / (60. * math.sqrt(math.pi)) < 2e-2


def test_abel_uniform_and_nonuniform_code_paths_give_the_same_numbers(monkeypatch):
    """np.trapz(dx=...) and np.trapz(x=...) on one exactly uniform grid: identical to rounding, so it does not matter on
    which side of PyAbel's 1e-13 uniformity test the reference's arange grid (joxsz_main.py:104) falls."""
    r = 16.0 * np.arange(1, 200)
    f = 1. / (1. + (r / 250.) ** 2) ** 1.7
    assert pyabel_direct.is_uniform_sampling(r)
    a = pyabel_direct.direct_transform_forward(f, r)
    monkeypatch.setattr(pyabel_direct, 'is_uniform_sampling', lambda _r: False)
    b = pyabel_direct.direct_transform_forward(f, r)
    np.testing.assert_allclose(a, b, rtol=1e-13)


def test_projection_volumes_cash_and_priors_edge_cases():
    """mbproj2 pieces (restated, unpinned): shells wholly inside the outermost annulus project to their full volume; a
    non-finite Cash sum is -inf; a value on a box bound is inside, a hair beyond it is -inf."""
    e = np.array([0., 1., 2.5, 4., 7.])
    V = mbp.projection_volume_matrix(e)
    vol = 4. / 3. * math.pi * (e[1:] ** 3 - e[:-1] ** 3)
    np.testing.assert_allclose(V.sum(axis=0), vol, rtol=1e-13)
    assert np.all(np.triu(V) == V) or np.all(np.tril(V) == V)           # an annulus sees only shells at or beyond its radius
    assert mbp.cash_log_likelihood(np.array([2., 3.]), np.array([1., np.inf])) == -np.inf
    assert mbp.cash_log_likelihood(np.array([2., 3.]), np.array([1., np.nan])) == -np.inf
    assert mbp.param_prior(0., 0., 1.) == 0.0 and mbp.param_prior(np.nextafter(0., -1.), 0., 1.) == -np.inf
    # a NaN passes mbproj2's comparisons (prior 0); the reference then fails the mass-monotonicity test with it (-inf)
    assert mbp.param_prior(np.nan, 0., 1.) == 0.0
