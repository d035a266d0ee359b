"""The table-driven exp and log of the per-walker kernel (csrc/jx_fastmath.hpp) against long double on the host: 2 ulp over the
ranges the profile chains meet and far beyond, the special values of the device library, and the log-posterior with and without
them (JOXSZ_PREP_FASTMATH=0)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _ulps(got, want_ld):
    want = want_ld.astype(np.float64)
    ulp = np.spacing(np.abs(want))
    return np.abs((got.astype(np.longdouble) - want_ld) / ulp.astype(np.longdouble)).astype(np.float64)


@pytest.fixture(scope='module')
def ctx():
    from joxsz_amd import datasets
    from joxsz_amd.posterior import JoxszPosterior
    post = JoxszPosterior(datasets.synthetic_problem(S=64, N=80, seed=1), device=0)
    yield post.ctx
    post.close()


def test_exp_and_log_within_two_ulp_of_long_double(ctx):
    assert np.finfo(np.longdouble).nmant >= 63
    rng = np.random.default_rng(0)
    n = 400000
    for lo, hi in ((-700.0, 700.0), (-40.0, 40.0), (-1e-3, 1e-3), (-1e-12, 1e-12)):
        x = rng.uniform(lo, hi, n)
        e, _ = ctx.fastmath_eval(x)
        assert _ulps(e, np.exp(x.astype(np.longdouble))).max() < 2.0, (lo, hi)
    for lo, hi in ((1e-300, 1e300), (1e-8, 1e8), (0.5, 2.0)):
        x = np.exp(rng.uniform(np.log(lo), np.log(hi), n))
        _, l = ctx.fastmath_eval(x)
        assert _ulps(l, np.log(x.astype(np.longdouble))).max() < 2.0, (lo, hi)
    x = 1.0 + rng.uniform(-1e-6, 1e-6, n)                                # around 1: the interval whose centre is exactly 1
    _, l = ctx.fastmath_eval(x)
    assert _ulps(l, np.log(x.astype(np.longdouble))).max() < 1.0
    x = 1.0 + np.exp(rng.uniform(np.log(1e-12), np.log(1e6), n))        # log(1 + x^a): what the pressure profile asks for
    _, l = ctx.fastmath_eval(x)
    assert _ulps(l, np.log(x.astype(np.longdouble))).max() < 2.0


def test_special_values_as_the_device_library_has_them(ctx):
    x = np.array([0.0, -0.0, 1.0, -1.0, np.inf, -np.inf, np.nan, 5e-324, 2.2250738585072014e-308, 1.7976931348623157e308,
                  709.782712893384, 709.79, -745.13, -745.14, -800.0, 800.0, 1e-320, -1e-320])
    e, l = ctx.fastmath_eval(x)
    with np.errstate(all='ignore'):
        we, wl = np.exp(x), np.log(x)
    for got, want in ((e, we), (l, wl)):
        assert np.array_equal(np.isnan(got), np.isnan(want))
        assert np.array_equal(np.isinf(got), np.isinf(want)) and np.array_equal(np.sign(got[np.isinf(got)]), np.sign(want[np.isinf(want)]))
        fin = np.isfinite(want)
        np.testing.assert_allclose(got[fin], want[fin], rtol=5e-16, atol=5e-324)


def test_log_posterior_with_and_without_the_tables(monkeypatch):
    """Same rejections; log-posteriors, X-ray log-likelihoods and T_SZ profiles to 1e-11 -- the difference is the rounding of the
    exponents' arguments, present in either library."""
    from joxsz_amd import datasets
    from joxsz_amd.posterior import JoxszPosterior
    for kw in ({}, dict(ne_mode='double')):
        pb = datasets.synthetic_problem(S=128, N=150, seed=3, **kw)
        th = datasets.walker_ball(pb, 300, spread=0.08, seed=3)          # wide: box rejections, mass vetoes, r_c > r_s
        res = {}
        for fm in ('1', '0'):
            monkeypatch.setenv('JOXSZ_PREP_FASTMATH', fm)
            post = JoxszPosterior(pb, device=0)
            res[fm] = (post.log_prob(th), post.stage(th, 'parts'), post.stage(th[:32], 'tprof'), post.stage(th[:32], 'pp'))
            post.close()
        a, b = res['1'][0], res['0'][0]
        fin = np.isfinite(b)
        assert 50 < fin.sum() < 300 and np.array_equal(np.isfinite(a), fin)
        np.testing.assert_allclose(a[fin], b[fin], rtol=1e-11)
        np.testing.assert_array_equal(res['1'][1][:, 3], res['0'][1][:, 3])          # the same reasons for every rejection
        ok = np.isfinite(res['0'][1][:, 0])
        np.testing.assert_allclose(res['1'][1][ok, 0], res['0'][1][ok, 0], rtol=1e-11)
        np.testing.assert_allclose(res['1'][2], res['0'][2], rtol=1e-12)
