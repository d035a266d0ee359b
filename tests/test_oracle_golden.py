"""The CPU oracle against the reference's own outputs (tests/golden/, produced
by oracle/make_golden.py running /root/reference/joxsz_funcs.py)."""
import numpy as np

from oracle import joxsz_oracle as orc

RTOL = 1e-12      # same scipy primitives, same order of operations


def _finite_rows(ref):
    return np.flatnonzero(np.isfinite(ref['ref_logp']))


def test_profile_components(golden):
    pb, ref = golden
    for k, th in enumerate(ref['thetas']):
        p = orc.pars_dict(pb, th)
        np.testing.assert_allclose(orc.press_fun(p, pb.r_pp), ref['ref_pp'][k], rtol=RTOL)
        np.testing.assert_allclose(orc.press_derivative(p, pb.r_pp), ref['ref_dpp'][k], rtol=RTOL)
        np.testing.assert_allclose(orc.vikh_function(p, pb.r_pp, pb.ne_mode), ref['ref_ne'][k], rtol=RTOL)
        np.testing.assert_allclose(orc.temp_fun(p, pb.r_pp, pb.ne_mode, getT_SZ=True), ref['ref_tsz'][k], rtol=RTOL)
        np.testing.assert_allclose(orc.mass_fun(p, pb.r_pp, pb.ne_mode), ref['ref_mass'][k], rtol=RTOL)
        assert orc.dens_prior(p) == ref['ref_densprior'][k]


def test_sz_stages(golden):
    pb, ref = golden
    for k, th in enumerate(ref['thetas']):
        p = orc.pars_dict(pb, th)
        st = orc.sz_stages(pb, p)
        np.testing.assert_allclose(st['bright'], ref['ref_bright'][k], rtol=1e-10, atol=1e-14)
        np.testing.assert_allclose(st['chisq'], ref['ref_chisq'][k], rtol=1e-10)
        np.testing.assert_allclose(st['ll'], ref['ref_ll'][k], rtol=1e-10)


def test_log_posterior(golden):
    pb, ref = golden
    got = orc.log_posterior_batch(pb, ref['thetas'])
    want = ref['ref_logp']
    assert np.array_equal(np.isfinite(got), np.isfinite(want))
    assert np.all(got[~np.isfinite(want)] == -np.inf)
    fin = _finite_rows(ref)
    assert fin.size >= 8 and (~np.isfinite(want)).sum() >= 4
    np.testing.assert_allclose(got[fin], want[fin], rtol=1e-11)


def test_xray_like(golden):
    pb, ref = golden
    for k in _finite_rows(ref):
        p = orc.pars_dict(pb, ref['thetas'][k])
        profs = orc.calc_profiles(pb, p)
        np.testing.assert_allclose(orc.like_from_profs(pb, profs), ref['ref_xlike'][k], rtol=1e-12)


def test_calc_integ_branch(golden_integ):
    """joxsz_funcs.py:480-487 executed by the reference (integrated Compton parameter, its chi^2 term inside 'll' and the
    total), and the linear-functional form of it that the HIP library uploads (``Problem.integ_weights``)."""
    from scipy.interpolate import interp1d
    pb, ref = golden_integ
    assert pb.calc_integ and 'ref_integ' in ref
    got = orc.log_posterior_batch(pb, ref['thetas'])
    fin = np.isfinite(ref['ref_logp'])
    assert np.array_equal(np.isfinite(got), fin)
    np.testing.assert_allclose(got[fin], ref['ref_logp'][fin], rtol=1e-11)
    w = pb.integ_weights()
    for k, th in enumerate(ref['thetas']):
        st = orc.sz_stages(pb, orc.pars_dict(pb, th))
        np.testing.assert_allclose(st['integ'], ref['ref_integ'][k], rtol=1e-12)
        np.testing.assert_allclose(st['ll'], ref['ref_ll'][k], rtol=1e-10)
        assert abs(st['ll'] + st['chisq'] / 2) > 1e-3                            # the extra term is there
        y = st['y']
        f0 = interp1d(np.append(-pb.r_pp, pb.r_pp), np.append(y, y), 'cubic')(0.)
        np.testing.assert_allclose(w @ np.append(f0, y), st['integ'], rtol=1e-12)
