"""BASELINE.json configs as stated, through the C-ABI on the GPU, with stage-level parity at full size.

configs[1]: 256 walkers, 256^2 map, 300-pt grid, SZ-only likelihood.
configs[2]/[3]: 512^2 map, 500-pt grid, joint likelihood; a 512-walker shard (what one of 8 ranks holds of configs[3]).
configs[4]: 1024^2 map, 1000-pt grid; a 1024-walker shard (what one of 8 ranks holds), fp64 leg.

The log-posterior of these problems is ~1e4 and dominated by the Cash term while chi^2/2 of the SZ side is ~10, so a
relative bound on the log-posterior alone is a weak statement about the SZ path (joxsz_funcs.py:472-479).  Every case
therefore also holds the extracted row, the surface-brightness profile and chi^2 to the oracle, chi^2/2 ABSOLUTELY.
"""
import ctypes

import numpy as np
import pytest

from oracle import joxsz_oracle as orc

pytestmark = pytest.mark.gpu

RTOL = 1e-6              # north_star: relative, on the log-posterior
RTOL_STAGE = 1e-9        # rows and profiles, relative to their largest entry
ATOL_HALF_CHISQ = 1e-6   # |chi^2/2 (GPU) - chi^2/2 (oracle)|, absolute


def _post(pb, **kw):
    from joxsz_amd.posterior import JoxszPosterior
    return JoxszPosterior(pb, device=0, **kw)


def _problem(S, N, seed, **kw):
    """Synthetic problem whose observations are the model at the fiducial vector plus noise (chi^2 ~ nflux)."""
    from joxsz_amd import datasets
    pb = datasets.synthetic_problem(S=S, N=N, seed=seed, **kw)
    p0 = orc.pars_dict(pb, datasets.fiducial_theta(pb))
    datasets.fill_data(pb, orc.sz_stages(pb, p0)['bright'], None if pb.sz_only else orc.calc_profiles(pb, p0), seed=seed)
    return pb


def _check_stages(pb, post, th, nref):
    """map_row / bright / chisq / logp of the first nref walkers against one oracle call per walker."""
    rows = post.stage(th[:nref], 'map_row')
    bright = post.stage(th[:nref], 'bright')
    chisq = post.stage(th[:nref], 'chisq')
    logp = post.log_prob(th[:nref])
    want_lp = orc.log_posterior_batch(pb, th[:nref])
    worst = dict(row=0.0, bright=0.0, half_chisq=0.0, logp=0.0)
    nfin = 0
    for k in range(nref):
        st = orc.sz_stages(pb, orc.pars_dict(pb, th[k]))
        worst['row'] = max(worst['row'], np.abs(rows[k] - st['map_row']).max() / np.abs(st['map_row']).max())
        worst['bright'] = max(worst['bright'], np.abs(bright[k] - st['bright']).max() / np.abs(st['bright']).max())
        worst['half_chisq'] = max(worst['half_chisq'], abs(chisq[k] - st['chisq']) / 2)
        assert np.isfinite(want_lp[k]) == np.isfinite(logp[k])
        if np.isfinite(want_lp[k]):
            nfin += 1
            worst['logp'] = max(worst['logp'], abs(logp[k] - want_lp[k]) / abs(want_lp[k]))
            # the parts add up to the total the way joxsz_funcs.py:538 adds them
    assert nfin >= nref // 2, 'too few finite walkers in the reference sample'
    assert worst['row'] < RTOL_STAGE, worst
    assert worst['bright'] < RTOL_STAGE, worst
    assert worst['half_chisq'] < ATOL_HALF_CHISQ, worst
    assert worst['logp'] < RTOL, worst
    return worst


def test_config1_sz_only_256():
    """configs[1] exactly: sz_only, S=256, N=300, 256 walkers; 16 of them against the oracle stage by stage, all of them
    twice (bitwise), and split over ragged launches (bitwise)."""
    from joxsz_amd import datasets
    pb = _problem(256, 300, seed=1, sz_only=True)
    th = datasets.walker_ball(pb, 256, spread=0.03, seed=1)
    post = _post(pb)
    assert post.ctx.conv == 'custom' and post.ctx.conv_layout['form'] == 'exact'
    a = post.log_prob(th)
    b = post.log_prob(th)
    np.testing.assert_array_equal(a, b)
    assert np.isfinite(a).sum() >= 200
    _check_stages(pb, post, th, 16)
    # SZ-only: the total is prior + sz_like, no X-ray term
    parts = post.stage(th[:16], 'parts')
    fin = np.isfinite(a[:16])
    np.testing.assert_allclose(parts[fin, 0], 0.0)
    np.testing.assert_allclose(a[:16][fin], parts[fin, 1] + parts[fin, 2], rtol=1e-13)
    post.close()
    small = _post(pb, max_batch=100)
    np.testing.assert_array_equal(small.log_prob(th), a)
    small.close()


def test_config3_shard_512_walkers():
    """One rank's share of configs[3] (4096 walkers over 8 ranks = 512 walkers at 512^2 / 500-pt, joint likelihood):
    16 walkers against the oracle with the absolute chi^2/2 bound, all 512 for determinism and rejections."""
    from joxsz_amd import datasets
    from joxsz_amd.dist import shard_bounds
    pb = _problem(512, 500, seed=2)
    full = datasets.walker_ball(pb, 4096, spread=0.03, seed=2)
    lo, hi = shard_bounds(4096, 8, 3)
    assert hi - lo == 512
    th = np.ascontiguousarray(full[lo:hi])
    post = _post(pb)
    assert post.ctx.conv == 'custom' and post.ctx.conv_layout['form'] == 'exact'
    a = post.log_prob(th)
    np.testing.assert_array_equal(a, post.log_prob(th))
    assert np.isfinite(a).sum() >= 400
    worst = _check_stages(pb, post, th, 16)
    parts = post.stage(th[:64], 'parts')
    fin = np.isfinite(a[:64])
    np.testing.assert_allclose(a[:64][fin], parts[fin, 0] + parts[fin, 1] + parts[fin, 2], rtol=1e-14)
    assert np.all(parts[~fin, 3] != 0)
    post.close()
    print('config3 shard worst errors', worst)


def test_config4_shard_1024_walkers_fp64():
    """One rank's share of configs[4] (8192 walkers over 8 ranks = 1024 walkers at 1024^2 / 1000-pt), fp64: 6 walkers
    against the oracle stage by stage (absolute chi^2/2 bound), all 1024 evaluated twice (bitwise)."""
    from joxsz_amd import datasets
    pb = _problem(1024, 1000, seed=4)
    th = datasets.walker_ball(pb, 1024, spread=0.02, seed=4)
    post = _post(pb)
    assert post.ctx.conv == 'custom'
    a = post.log_prob(th)
    np.testing.assert_array_equal(a, post.log_prob(th))
    assert np.isfinite(a).sum() >= 900
    worst = _check_stages(pb, post, th, 6)
    post.close()
    print('config4 shard worst errors', worst)


def test_headline_stage_parity_32_walkers():
    """configs[2] (the headline shape): 32 walkers spread wider around the fiducial vector, stage by stage."""
    from joxsz_amd import datasets
    pb = _problem(512, 500, seed=0)
    th = datasets.walker_ball(pb, 32, spread=0.06, seed=5)
    post = _post(pb)
    worst = _check_stages(pb, post, th, 32)
    post.close()
    print('headline worst errors', worst)


def test_nan_and_inf_parameters_are_rejections():
    """NaN / +-inf in any thawed parameter gives -inf (never NaN, never an exception): joxsz_funcs.py:519-520 for the box
    priors, and emcee cannot take NaN from a log-probability function."""
    from joxsz_amd import datasets
    pb = _problem(64, 80, seed=7)
    t0 = datasets.fiducial_theta(pb)
    ok = _post(pb)
    base = ok.log_prob(t0[None, :])[0]
    assert np.isfinite(base)
    bad = []
    for k in range(pb.ndim):
        for v in (np.nan, np.inf, -np.inf):
            t = t0.copy(); t[k] = v
            bad.append(t)
    bad = np.array(bad)
    th = np.vstack((t0[None, :], bad, t0[None, :]))
    for route in ('map', 'operator'):
        ok.ctx.set_route(route)
        got = ok.log_prob(th)
        assert got[0] == base or route == 'operator'
        assert np.isfinite(got[0]) and np.isfinite(got[-1])
        assert np.all(got[1:-1] == -np.inf), (route, got)
        assert not np.isnan(got).any()
    ok.ctx.set_route('map')
    parts = ok.stage(bad, 'parts')
    assert np.all(parts[:, 3] != 0)
    assert ok.getLikelihood(bad[0]) == -np.inf
    ok.close()


def test_c_abi_error_codes():
    """Every misuse returns its negative jx_status (include/joxsz_hip.h) and leaves the context usable or cleanly dead."""
    from joxsz_amd import datasets, hip_backend as hb
    lib = hb.load_library()
    pb = datasets.synthetic_problem(S=64, N=80, seed=3)
    vp, dp = ctypes.c_void_p, ctypes.POINTER(ctypes.c_double)
    INVALID, STATE, MISSING, NODEVICE, UNSUPPORTED = -1, -2, -3, -7, -8

    def make(**over):
        cfg = hb.config_from_problem(pb, **{k: v for k, v in over.items() if k in ('device', 'conv', 'max_batch')})
        for k, v in over.items():
            if k not in ('device', 'conv', 'max_batch'):
                setattr(cfg, k, v)
        h = vp()
        return lib.jx_create(ctypes.byref(cfg), ctypes.byref(h)), h

    def upload_all(h, skip=()):
        for tid, name in enumerate(hb.TENSORS):
            if name in skip or name == 'integ_w':               # (calc_integ is off: that tensor is not wanted)
                continue
            a = pb.thawed_idx if name == 'thawed_idx' else getattr(pb, name)
            a = np.ascontiguousarray(a, dtype=np.int32 if name in ('par_kind', 'thawed_idx') else np.float64)
            assert lib.jx_upload(h, tid, a.ctypes.data_as(vp), a.nbytes) == 0, name

    # jx_create
    assert make(abi_version=99)[0] == INVALID
    assert make(B=4)[0] == INVALID                                   # even beam side
    assert make(npar=17)[0] == INVALID
    assert make(device=63)[0] == NODEVICE
    assert make(N=5000)[0] == UNSUPPORTED                            # beyond the LDS-resident spline
    assert lib.jx_create(None, None) == INVALID

    rc, h = make()
    assert rc == 0
    th = np.ascontiguousarray(datasets.walker_ball(pb, 4, spread=0.01, seed=1))
    out = np.empty(4)
    # before finalize
    assert lib.jx_eval(h, th.ctypes.data_as(dp), 4, out.ctypes.data_as(dp)) == STATE
    assert lib.jx_set_route(h, 1) == STATE
    assert lib.jx_get_conv_mode(h) == STATE
    assert lib.jx_finalize(h) == MISSING
    assert b'missing' in lib.jx_last_error(h)
    a = np.zeros(5)
    assert lib.jx_upload(h, 0, a.ctypes.data_as(vp), a.nbytes) == INVALID          # wrong size for r_pp
    assert lib.jx_upload(h, 999, a.ctypes.data_as(vp), a.nbytes) == INVALID
    assert lib.jx_upload(h, 0, None, 0) == INVALID
    upload_all(h, skip=('lnrate',))
    assert lib.jx_finalize(h) == MISSING
    upload_all(h)
    assert lib.jx_finalize(h) == 0
    assert lib.jx_finalize(h) == STATE
    r = np.ascontiguousarray(pb.r_pp)
    assert lib.jx_upload(h, 0, r.ctypes.data_as(vp), r.nbytes) == STATE             # upload after finalize
    # evaluation arguments
    assert lib.jx_eval(h, th.ctypes.data_as(dp), -1, out.ctypes.data_as(dp)) == INVALID
    assert lib.jx_eval(h, None, 4, out.ctypes.data_as(dp)) == INVALID
    assert lib.jx_eval(h, None, 0, None) == 0                                       # an empty batch is fine
    assert lib.jx_eval_stage(h, th.ctypes.data_as(dp), 4, 99, out.ctypes.data_as(dp), out.nbytes) == INVALID
    assert lib.jx_eval_stage(h, th.ctypes.data_as(dp), 4, 7, out.ctypes.data_as(dp), 8) == INVALID   # chisq: 4 doubles expected
    assert lib.jx_set_route(h, 5) == INVALID
    big = np.empty((pb.N, pb.nrow))
    assert lib.jx_get_operator(h, big.ctypes.data_as(dp), big.nbytes) == STATE      # operator route never selected
    assert lib.jx_sample(h, th.ctypes.data_as(dp), 3, 1, 2.0, 0, None, None, None) == INVALID       # odd ensemble
    assert lib.jx_sample(h, th.ctypes.data_as(dp), 4, 1, 1.0, 0, None, None, None) == INVALID       # a must exceed 1
    assert lib.jx_set_par_vals(h, th.ctypes.data_as(dp), 3) == INVALID
    # still usable after all of that
    assert lib.jx_eval(h, th.ctypes.data_as(dp), 4, out.ctypes.data_as(dp)) == 0
    want = orc.log_posterior_batch(pb, th)
    fin = np.isfinite(want)
    np.testing.assert_allclose(out[fin], want[fin], rtol=RTOL)
    lib.jx_destroy(h)
    lib.jx_destroy(None)                                              # no-op

    # a non-increasing radial grid is refused at finalize
    rc, h = make()
    upload_all(h)
    bad = np.ascontiguousarray(pb.r_pp[::-1])
    assert lib.jx_upload(h, 0, bad.ctypes.data_as(vp), bad.nbytes) == 0
    assert lib.jx_finalize(h) == INVALID
    lib.jx_destroy(h)


def test_calc_integ_branch_on_the_gpu(golden_integ):
    """The integrated-Compton term (joxsz_funcs.py:480-487, SZ_data.calc_integ) against the reference's own run of it:
    'integ' output, 'll' with the extra chi^2 term, the total; both routes."""
    pb, ref = golden_integ
    th = ref['thetas']
    fin = np.isfinite(ref['ref_logp'])
    post = _post(pb)
    got = post.log_prob(th)
    integ = post.stage(th, 'integ')
    parts = post.stage(th, 'parts')
    assert np.array_equal(np.isfinite(got), fin)
    np.testing.assert_allclose(got[fin], ref['ref_logp'][fin], rtol=1e-9)
    np.testing.assert_allclose(integ[fin], ref['ref_integ'][fin], rtol=1e-11)
    np.testing.assert_allclose(parts[fin, 1], ref['ref_ll'][fin], rtol=1e-8)
    post.updateThawed(th[1])
    np.testing.assert_allclose(post.get_sz_like('integ'), ref['ref_integ'][1], rtol=1e-11)
    np.testing.assert_allclose(post.get_sz_like('ll'), ref['ref_ll'][1], rtol=1e-8)
    post.ctx.set_route('operator')
    got_op = post.log_prob(th)
    post.close()
    assert np.array_equal(np.isfinite(got_op), fin)
    np.testing.assert_allclose(got_op[fin], ref['ref_logp'][fin], rtol=1e-9)
    # without the switch the same problem gives the plain chi^2 likelihood
    import copy
    pb0 = copy.deepcopy(pb)
    pb0.calc_integ = False
    p0 = _post(pb0)
    ll0 = p0.stage(th, 'parts')[:, 1]
    p0.close()
    assert np.all(np.abs(ll0[fin] - parts[fin, 1]) > 1e-6)


def test_loader_tensors_through_the_hip_path(golden_bundled):
    """The setup layer (joxsz_amd/setup_host.py + datasets.py) builds the problem tensors from the parsed
    bundled data files alone (tests/golden/bundled_inputs.npz); the HIP path on THOSE tensors reproduces the reference's
    log-posteriors, which it computed on tensors from its own setup functions (joxsz_funcs.py:16-134)."""
    import os
    from joxsz_amd import setup_host as sh, datasets
    from joxsz_amd.problem import Problem
    ref_pb, ref = golden_bundled
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'bundled_inputs.npz'))
    step, kpc_as, R_b = 2., datasets.KPC_AS_CLJ1226, 5000.
    flux_data = z['flux_data']
    prof = sh.clip_beam_profile(z['beam_r'], z['beam_prof'])
    beam_2d, fwhm = sh.beam_image(step, flux_data[0][-1], approx=False, profile=prof)
    radius, sep, r_pp = sh.sz_axes(step, kpc_as, flux_data[0][-1], fwhm, R_b)
    d_mat = sh.pixel_radius_matrix(radius * kpc_as)
    wn, tf = sh.transfer_function(z['wn_as'], z['tf'], approx=False)
    filtering = sh.filter_image(wn, tf, d_mat.shape[0], step)
    geo = sh.annuli_geometry(z['edges_arcmin'], kpc_as)
    bands = [sh.band_from_profiles(f, b) for f, b in zip(z['fg'], z['bg'])]
    kw = {k: getattr(ref_pb, k) for k in Problem._ARRAYS + Problem._SCALARS}
    kw.update(par_names=ref_pb.par_names, flux_data=flux_data, beam_2d=beam_2d, radius=radius, r_pp=r_pp, d_mat=d_mat,
              filtering=filtering, conv_T=z['conv_T'], conv_v=1e3 * z['conv_jy'],
              x_r_ne_kpc=geo['midpt_kpc'].copy(), x_r_T_kpc=geo['midpt_kpc'].copy(), projvols=geo['projvols'],
              geomarea=geo['geomarea'], cts=np.array([b['cts'] for b in bands]),
              areascales=np.array([b['areascales'] for b in bands]), exposures=np.array([b['exposures'] for b in bands]),
              backrates=np.array([b['backrates'] for b in bands]))
    pb = Problem(**kw).validate()
    assert pb.S == ref_pb.S == 171 and pb.N == ref_pb.N and pb.B == ref_pb.B
    post = _post(pb)
    assert post.ctx.conv == 'custom'
    got = post.log_prob(ref['thetas'])
    bright = post.stage(ref['thetas'][:3], 'bright')
    post.close()
    want = ref['ref_logp']
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), fin)
    np.testing.assert_allclose(got[fin], want[fin], rtol=1e-9)
    np.testing.assert_allclose(bright, ref['ref_bright'][:3], rtol=1e-8, atol=1e-12 * np.abs(ref['ref_bright'][:3]).max())


@pytest.mark.parametrize('S,N,W', [(512, 500, 512), (1024, 1000, 1024)])
def test_fp32_variant_tolerance_sweep(S, N, W):
    """BASELINE configs[4]: the two fp32 variants against the fp64 path on the same walkers around the posterior mode --
    'f32' (spline arrays rounded to fp32 once, where the matrix product stores them, and read as fp32 by the sample
    evaluation; every sum, the matrix-core product and the tail in fp64) and 'f32c' (fp32 arithmetic: stage 1 in packed fp32
    FMAs, stage 2 on the fp32 matrix cores, partial rows in fp32, K slices added in fp64): relative difference of the
    log-posterior, absolute difference of chi^2 / 2, the extracted row.  The bars are what each variant is sold as: 'f32'
    rounding of inputs only (1e-9 class), 'f32c' fp32 sums over 257-513 samples and 256-term slices (1e-7 class) -- both inside
    the north_star's 1e-6 on the log-posterior."""
    from joxsz_amd import datasets
    pb = _problem(S, N, seed=S)
    th = datasets.walker_ball(pb, W, spread=0.02, seed=S)
    p64 = _post(pb)
    lp64, ch64, row64 = p64.log_prob(th), p64.stage(th[:64], 'chisq'), p64.stage(th[:64], 'map_row')
    p64.close()
    fin = np.isfinite(lp64)
    assert fin.sum() > 0.8 * W
    bars = {'f32': (1e-6, 1e-7, 1e-4, 1e-4), 'f32c': (1e-6, 2e-7, 5e-5, 2e-2)}     # rel dlogp max, median; row; |d chi^2 / 2|
    seen = {}
    for dt in ('f32', 'f32c'):
        p32 = _post(pb, dtype=dt)
        assert p32.ctx.dtype == dt and p32.ctx.conv_layout['form'] == 'lowrank'
        lp32, ch32, row32 = p32.log_prob(th), p32.stage(th[:64], 'chisq'), p32.stage(th[:64], 'map_row')
        np.testing.assert_array_equal(lp32, p32.log_prob(th))               # deterministic too
        p32.close()
        assert np.array_equal(np.isfinite(lp32), fin)
        rel = np.abs(lp32[fin] - lp64[fin]) / np.abs(lp64[fin])
        dchi = np.abs(ch32 - ch64) / 2
        rrow = np.abs(row32 - row64).max(axis=1) / np.abs(row64).max(axis=1)
        print('%s vs f64 at %d^2/%d-pt, %d walkers: rel dlogp max %.3e median %.3e | abs dchi^2/2 max %.3e median %.3e | row max %.3e'
              % (dt, S, N, W, rel.max(), np.median(rel), dchi.max(), np.median(dchi), rrow.max()))
        b = bars[dt]
        assert rel.max() < b[0] and np.median(rel) < b[1] and rrow.max() < b[2] and dchi.max() < b[3]
        assert np.any(lp32[fin] != lp64[fin])                               # it really is another arithmetic
        seen[dt] = (rel.max(), rrow.max())
    assert seen['f32c'][1] > seen['f32'][1]                                 # fp32 sums cost more than fp32 inputs


def test_fp32_variant_exists_on_the_contracted_route_only(monkeypatch):
    """dtype f32 on the rocFFT sequence must be an error, never a silent fp64 evaluation; on the contracted route it exists at
    every side (odd and small ones included, both forms) and differs from fp64 by fp32 rounding only."""
    from joxsz_amd.hip_backend import JoxszHipError
    from joxsz_amd import datasets
    pb = datasets.synthetic_problem(S=512, N=500, seed=1)
    with pytest.raises(JoxszHipError, match='unsupported'):
        _post(pb, dtype='f32', conv='rocfft')
    monkeypatch.setenv('JOXSZ_MIX_FORM', 'full')
    with pytest.raises(JoxszHipError, match='low-rank form'):               # fp32 arithmetic: low-rank form only
        _post(datasets.synthetic_problem(S=171, N=313, seed=1), dtype='f32c')
    monkeypatch.delenv('JOXSZ_MIX_FORM')
    for S, N in ((513, 500), (96, 120), (171, 313)):
        pb = datasets.synthetic_problem(S=S, N=N, seed=1)
        th = datasets.walker_ball(pb, 12, spread=0.02, seed=1)
        p64 = _post(pb)
        a = p64.log_prob(th)
        p64.close()
        p32 = _post(pb, dtype='f32')
        b = p32.log_prob(th)
        with pytest.raises(JoxszHipError):                               # the profile taps exist in the f64 build only
            p32.stage(th[:1], 'y')
        p32.close()
        fin = np.isfinite(a)
        assert fin.sum() >= 8 and np.array_equal(np.isfinite(b), fin)
        np.testing.assert_allclose(b[fin], a[fin], rtol=2e-5)            # (small maps: a small difference of large terms)
        assert np.any(b[fin] != a[fin])


def test_truncation_guard_and_automatic_tightening(monkeypatch, legacy_forms):
    """jx_finalize measures what the singular-value cut of the low-rank form costs, against the rocFFT sequence, at the
    current parameter values and at the corners of the prior box in (a, b, r_p) (jx_get_truncation); above the bounds (row at
    the current values 1e-9, SZ log-likelihood over the box 1e-8; lowered here to force the cases) it rebuilds the tables in
    place with a cut ten times tighter, again and again until the bounds hold or every term above rounding is kept."""
    from joxsz_amd import datasets
    pb = _problem(512, 500, seed=3)
    th = datasets.walker_ball(pb, 24, spread=0.03, seed=3)
    post = _post(pb)
    tr = post.ctx.truncation
    assert tr['tol'] == 1e-8 and not tr['retried'] and tr['points'] == 9 and tr['rank'] == post.ctx.conv_layout['rank']
    assert 0 <= tr['est_rel_row_err'] < 1e-10 and tr['est_rel_row_err'] <= tr['est_rel_row_err_box'] < 1e-6
    assert 0 <= tr['est_rel_sz_like_err_box'] <= 1e-8
    a = post.log_prob(th)
    chi_a = post.stage(th, 'chisq')
    post.close()
    monkeypatch.setenv('JOXSZ_TRUNC_BOUND', '1e-11')                  # row bound 1e-11, log-likelihood bound 1e-10 over the box
    post = _post(pb)
    tr1 = post.ctx.truncation
    assert tr1['retried'] >= 1
    if post.ctx.conv_layout['form'] == 'full':                       # (more terms made the exact form the cheaper one: nothing left to truncate)
        # (what is left to measure is the sub-grid of map samples, kept when it is inside the lowered bounds too)
        assert tr1['rank'] == 0 and (tr1['est_rel_row_err'] == -1.0 or (post.ctx.sampling['active'] and 0 <= tr1['est_rel_row_err'] <= 1e-11))
    else:
        assert tr1['tol'] < 1e-8 and tr1['rank'] > tr['rank'] and tr1['rank'] == post.ctx.conv_layout['rank']
        assert 0 <= tr1['est_rel_row_err'] <= 1e-11 and 0 <= tr1['est_rel_sz_like_err_box'] <= 1e-10
    post.close()
    monkeypatch.setenv('JOXSZ_TRUNC_BOUND', '1e-18')                  # never met: ends with every term above rounding
    post = _post(pb)
    tr2 = post.ctx.truncation
    # (with every term kept the cost model may hand the problem to the full form, which has none to drop: rank 0)
    exact = post.ctx.conv_layout['form'] == 'full'
    if exact:
        assert tr2['retried'] >= 1 and tr2['rank'] == 0 and tr2['est_rel_row_err'] == -1.0          # nothing truncated, nothing to estimate
        assert not post.ctx.sampling['active'] and post.ctx.sampling['removed_by_the_guard'] == 1   # (no sub-grid meets 1e-18 either)
    else:
        assert tr2['retried'] >= 5 and tr2['tol'] <= 1.01e-13 and tr2['rank'] > tr1['rank'] and 0 <= tr2['est_rel_row_err'] < 1e-12
    b = post.log_prob(th)
    chi_b = post.stage(th, 'chisq')
    rows = post.stage(th[:4], 'map_row')
    post.close()
    fin = np.isfinite(a)
    np.testing.assert_allclose(b[fin], a[fin], rtol=1e-9)
    assert np.max(np.abs(chi_a[fin] - chi_b[fin])) / 2 < 1e-7         # the default cut against no cut: |d(chi^2/2)| far inside 1e-6
    st = orc.sz_stages(pb, orc.pars_dict(pb, th[0]))
    assert np.abs(rows[0] - st['map_row']).max() / np.abs(st['map_row']).max() < 1e-12
    # an explicit cut is measured, never overridden
    monkeypatch.setenv('JOXSZ_LOWRANK_TOL', '1e-7')
    post = _post(pb)
    tr3 = post.ctx.truncation
    post.close()
    assert tr3['tol'] == 1e-7 and tr3['retried'] == 0 and tr3['est_rel_sz_like_err_box'] > tr['est_rel_sz_like_err_box']


@pytest.mark.parametrize('S,N,form', [(512, 500, 'lowrank'), (512, 500, 'full'), (513, 500, 'lowrank'), (1024, 1000, 'lowrank'), (256, 300, 'full')])
def test_the_product_computes_the_outputs_the_data_radii_spline_reads(S, N, form, monkeypatch):
    """The tail interpolates the extracted row at the data radii with a cubic spline (joxsz_funcs.py:476); the weights of that
    spline decay by 2 - sqrt(3) per knot, so the row beyond the last data radius + ~35 pixels never reaches an fp64 sum.  The
    timed matrix-core product computes the outputs read (whole tiles of 16; jx_get_output_pruning) -- 96 of 256 at 512^2, 96 of
    512 at 1024^2 -- and the result is the one of the product over every output (JOXSZ_PRUNE_OUTPUTS=0) to the last bits; the
    row tap still delivers the whole row."""
    from joxsz_amd import datasets
    pb = _problem(S, N, seed=S + 1)
    th = datasets.walker_ball(pb, 150, spread=0.03, seed=S + 1)
    monkeypatch.setenv('JOXSZ_MIX_FORM', form)
    post = _post(pb)
    pr = post.ctx.output_pruning
    nrow = S - S // 2
    assert pr['nrow'] == nrow and pr['active'] and 80 <= pr['outputs_read_by_the_tail'] <= 96 and pr['outputs_computed'] == 96
    a = post.log_prob(th)
    row = post.stage(th[:3], 'map_row')
    chi_a = post.stage(th, 'chisq')                                  # (a tap: the whole row is computed)
    post.close()
    assert row.shape == (3, nrow)
    monkeypatch.setenv('JOXSZ_PRUNE_OUTPUTS', '0')
    ref = _post(pb)
    assert not ref.ctx.output_pruning['active']
    b = ref.log_prob(th)
    chi_b = ref.stage(th, 'chisq')
    ref.close()
    fin = np.isfinite(b)
    assert fin.sum() >= 100 and np.array_equal(np.isfinite(a), fin)
    np.testing.assert_allclose(a[fin], b[fin], rtol=1e-13)
    np.testing.assert_allclose(chi_a[fin], chi_b[fin], rtol=1e-11)
    st = orc.sz_stages(pb, orc.pars_dict(pb, th[0]))
    assert np.abs(row[0] - st['map_row']).max() / np.abs(st['map_row']).max() < 1e-9


@pytest.mark.parametrize('S,N,form', [(512, 500, 'lowrank'), (513, 500, 'lowrank'), (1024, 1000, 'lowrank'), (300, 260, 'lowrank'), (512, 500, 'full'), (257, 300, 'full')])
def test_stage_one_evaluates_a_sub_grid_of_the_quadrant(S, N, form, monkeypatch):
    """Away from the core the Compton-y map varies on the scale of the radius, not of the pixel: stage 1 of the low-rank form
    evaluates a tensor sub-grid of the quadrant's rows and columns (jx_get_sampling: every row below 40 pixels from the axis,
    every second up to 160, every fourth up to 320, every eighth beyond) and the interpolation to the others sits in both operators.  Against the
    same context with every distinct sample evaluated (JOXSZ_MIX_SUBSAMPLE=0): the extracted row to 1e-10 of its maximum,
    the log-posterior to 1e-10 -- two orders inside what the truncation of the transfer-function weights costs already -- and
    both against the oracle.  The full form (any beam, any real transfer function) lists the samples of the same sub-grid as
    rows of its operator; there the guard's measurement is that of the sub-grid alone."""
    from joxsz_amd import datasets
    pb = _problem(S, N, seed=S + 7)
    th = datasets.walker_ball(pb, 80, spread=0.04, seed=S + 7)
    monkeypatch.setenv('JOXSZ_MIX_FORM', form)
    post = _post(pb)
    smp = post.ctx.sampling
    NU = S // 2 + 1
    assert smp['active'] and smp['rows_of_the_quadrant'] == NU and smp['removed_by_the_guard'] == 0
    rows_kept = smp['rows']
    assert len(rows_kept) == smp['rows_evaluated'] < 0.75 * NU and rows_kept[0] == 0 and rows_kept[-1] == NU - 1
    assert np.array_equal(rows_kept[:smp['full_below']], np.arange(smp['full_below'])) and np.all(np.diff(rows_kept) >= 1) and np.diff(rows_kept).max() <= 8
    a, row_a, chi_a = post.log_prob(th), post.stage(th[:4], 'map_row'), post.stage(th, 'chisq')
    tr_a = post.ctx.truncation
    post.close()
    monkeypatch.setenv('JOXSZ_MIX_SUBSAMPLE', '0')
    ref = _post(pb)
    assert not ref.ctx.sampling['active'] and ref.ctx.sampling['rows_evaluated'] == NU
    b, row_b, chi_b = ref.log_prob(th), ref.stage(th[:4], 'map_row'), ref.stage(th, 'chisq')
    tr_b = ref.ctx.truncation
    ref.close()
    fin = np.isfinite(b)
    assert fin.sum() >= 50 and np.array_equal(np.isfinite(a), fin)
    np.testing.assert_allclose(a[fin], b[fin], rtol=1e-10)
    np.testing.assert_allclose(chi_a[fin], chi_b[fin], rtol=1e-8, atol=1e-8)
    assert np.abs(row_a - row_b).max() / np.abs(row_b).max() < 1e-10
    if form == 'full':
        # nothing truncated: what the guard measures is the sub-grids alone (without the one of the map samples: the radial one of
        # the spline-array product, at rounding level)
        assert 0 <= tr_b['est_rel_row_err'] < 1e-12 and 0 <= tr_a['est_rel_row_err'] < 1e-10 and 0 <= tr_a['est_rel_sz_like_err_box'] < 1e-9
    else:
        # the guard's own measurement (against the rocFFT facility) moves by less than a fifth of its bounds
        assert abs(tr_a['est_rel_row_err'] - tr_b['est_rel_row_err']) < 0.2 * tr_a['bound']
        assert abs(tr_a['est_rel_sz_like_err_box'] - tr_b['est_rel_sz_like_err_box']) < 0.2 * tr_a['bound_sz_like']
    want = orc.log_posterior_batch(pb, th[:6])
    np.testing.assert_allclose(a[:6][np.isfinite(want)], want[np.isfinite(want)], rtol=1e-6)
    st = orc.sz_stages(pb, orc.pars_dict(pb, th[0]))
    assert np.abs(row_a[0] - st['map_row']).max() / np.abs(st['map_row']).max() < 1e-7


def test_sub_grid_is_not_taken_beyond_the_radial_grid_and_the_guard_removes_a_coarse_one(monkeypatch, legacy_forms):
    """(a) A quadrant that reaches beyond the last radius of the grid holds fill values (joxsz_funcs.py:455): no smoothness to
    interpolate on, every distinct sample is evaluated.  (b) A sub-grid far too coarse for the profile (JOXSZ_MIX_SUBSAMPLE=
    8,8,4: every fourth row from 8 pixels on, 4-point) is measured by the guard of jx_finalize like the truncation, taken away,
    and HipContext says so once; results then equal those of the context that never had one."""
    import warnings
    from joxsz_amd import datasets
    from joxsz_amd.hip_backend import JoxszTruncationWarning
    monkeypatch.setenv('JOXSZ_MIX_FORM', 'lowrank')
    pb = _problem(512, 300, seed=9)                                   # corner of the quadrant at 362 pixels, grid ends at 300
    post = _post(pb)
    assert not post.ctx.sampling['active'] and post.ctx.sampling['rows_evaluated'] == 257
    th = datasets.walker_ball(pb, 6, spread=0.03, seed=9)
    got = post.log_prob(th)
    post.close()
    want = orc.log_posterior_batch(pb, th)
    fin = np.isfinite(want)
    assert fin.sum() >= 3
    np.testing.assert_allclose(got[fin], want[fin], rtol=1e-6)
    pb = _problem(512, 500, seed=9)
    th = datasets.walker_ball(pb, 40, spread=0.03, seed=9)
    monkeypatch.setenv('JOXSZ_MIX_SUBSAMPLE', '8,8,4')
    with pytest.warns(JoxszTruncationWarning, match='sub-grid') as rec:
        post = _post(pb)
    smp, tr = post.ctx.sampling, post.ctx.truncation
    assert len(rec) == 1 and not smp['active'] and smp['removed_by_the_guard'] == 1 and smp['rows_evaluated'] == 257
    assert tr['retried'] == 0 and tr['cap_removed'] == 0 and tr['rank'] == 16 and 0 <= tr['est_rel_row_err'] <= tr['bound']
    a = post.log_prob(th)
    post.close()
    monkeypatch.setenv('JOXSZ_MIX_SUBSAMPLE', '0')
    with warnings.catch_warnings():
        warnings.simplefilter('error', JoxszTruncationWarning)
        ref = _post(pb)
    b = ref.log_prob(th)
    ref.close()
    np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize('S,N,measured', [(512, 500, False), (512, 500, True), (256, 300, False)])
def test_default_route_over_the_whole_prior_box_against_the_oracle(S, N, measured):
    """Walkers drawn UNIFORMLY OVER THE WHOLE PRIOR BOX -- steep and flat profiles, knees, everything the box allows, not a ball around
    the fiducial vector -- through the default route (the exact form: nothing truncated, every pixel and radius) and through the CPU
    oracle: the same walkers rejected; the log-posterior of the others within 1e-9 relative (median under 1e-12: what the two
    libraries' exp / log / pow leave), and an ABSOLUTE bar where the chain lives: |delta chi^2 / 2| <= 1e-8 on every walker with
    chi^2 < 1e3.  Synthetic and measured (bundled) beam and transfer function."""
    from joxsz_amd import datasets
    if measured:
        from test_gpu_parity import _measured_problem
        pb = _measured_problem(S, N)
        p0 = orc.pars_dict(pb, datasets.fiducial_theta(pb))
        datasets.fill_data(pb, orc.sz_stages(pb, p0)['bright'], orc.calc_profiles(pb, p0), seed=S)
    else:
        pb = _problem(S, N, seed=S + 11)
    rng = np.random.default_rng(S + int(measured))
    names = list(pb.par_names)
    th0 = datasets.fiducial_theta(pb)
    nw = 400                                                           # (most of the box is rejected: r_c > r_s, the mass veto, non-positive count rates)
    th = np.repeat(th0[None, :], nw, axis=0)
    for k, ip in enumerate(pb.thawed_idx):
        lo, hi = pb.par_min[ip], pb.par_max[ip]
        if np.isfinite(lo) and np.isfinite(hi) and hi > lo:
            th[:, k] = rng.uniform(lo, hi, nw)
        elif pb.par_kind[ip] == 1 and pb.par_sigma[ip] > 0:
            th[:, k] = pb.par_mu[ip] + pb.par_sigma[ip] * rng.uniform(-3.0, 3.0, nw)
    # (a band of walkers near the fiducial vector as well: the box itself rarely lands where chi^2 is small)
    th[:40] = datasets.walker_ball(pb, 40, spread=0.05, seed=S)
    post = _post(pb)
    got = post.log_prob(th)
    smp, form = post.ctx.sampling, post.ctx.conv_layout['form']
    assert form == 'exact' and not smp['active'] and not post.ctx.radial_sampling['active']
    want = orc.log_posterior_batch(pb, th)
    fin = np.isfinite(want)
    assert fin.sum() >= 20, (fin.sum(), names)
    assert np.array_equal(np.isfinite(got), fin)
    chi_gpu = post.stage(th[fin], 'chisq')
    post.close()
    chi_cpu = np.array([orc.sz_stages(pb, orc.pars_dict(pb, t))['chisq'] for t in th[fin]])
    near = chi_cpu < 1e3
    rel = np.abs(got[fin] - want[fin]) / np.abs(want[fin])
    dhalf = np.abs(chi_gpu - chi_cpu)[near] / 2 if near.any() else np.zeros(1)
    print('whole prior box, S=%d N=%d %s: %d of %d walkers finite (%d with chi^2 < 1e3), form %s: log-posterior max rel err %.2e, median %.2e; |delta chi^2/2| max %.2e'
          % (S, N, 'measured inputs' if measured else 'synthetic inputs', fin.sum(), nw, near.sum(), form, rel.max(), np.median(rel), dhalf.max()))
    assert near.sum() >= 10
    assert rel.max() < 1e-9, (rel.max(), form)
    assert np.median(rel) < 1e-12
    assert dhalf.max() <= 1e-8


@pytest.mark.parametrize('S,N,kw', [(512, 500, {}), (1024, 1000, {}), (256, 300, dict(sz_only=True)), (512, 500, dict(ne_mode='double'))])
def test_spline_array_product_on_a_radial_sub_grid(S, N, kw, monkeypatch, legacy_forms):
    """The pressure profile is smooth away from the core: the spline-array product multiplies its values on a sub-grid of the radial grid
    (jx_get_radial_sampling: every radius below 64, every second to 256, every fourth beyond -- 222 of 500) by an operator that carries
    the 18-point interpolation to the others.  Against the same context with every radius (JOXSZ_AG_SUBSAMPLE=0): spline ordinates to
    1e-12 of their maximum, the extracted row to 1e-11, the log-posterior to 1e-10; and against the oracle."""
    from joxsz_amd import datasets
    pb = _problem(S, N, seed=S + 3, **kw)
    th = datasets.walker_ball(pb, 64, spread=0.05, seed=S + 3)
    post = _post(pb)
    rs = post.ctx.radial_sampling
    assert rs['active'] and rs['radii_of_the_grid'] == N and rs['radii_in_use'] == len(rs['rows']) < 0.75 * N and rs['removed_by_the_guard'] == 0
    assert np.array_equal(rs['rows'][:rs['full_below']], np.arange(rs['full_below'])) and rs['rows'][-1] == N - 1 and np.diff(rs['rows']).max() <= 8
    a, row_a = post.log_prob(th), post.stage(th[:4], 'map_row')
    post.log_prob(th[:8])
    cf_a = post.ctx.workspace('splines')[:, :8, :].copy()
    post.close()
    monkeypatch.setenv('JOXSZ_AG_SUBSAMPLE', '0')
    ref = _post(pb)
    assert not ref.ctx.radial_sampling['active'] and ref.ctx.radial_sampling['radii_in_use'] == N
    b, row_b = ref.log_prob(th), ref.stage(th[:4], 'map_row')
    ref.log_prob(th[:8])
    cf_b = ref.ctx.workspace('splines')[:, :8, :].copy()
    ref.close()
    fin = np.isfinite(b)
    assert fin.sum() >= 30 and np.array_equal(np.isfinite(a), fin)
    np.testing.assert_allclose(a[fin], b[fin], rtol=1e-10)
    assert np.abs(row_a - row_b).max() / np.abs(row_b).max() < 1e-11
    for comp, bar in ((0, 1e-12), (1, 1e-9)):                             # y_k; M_k (second differences amplify)
        assert np.abs(cf_a[:, :, comp] - cf_b[:, :, comp]).max() <= bar * np.abs(cf_b[:, :, comp]).max()
    want = orc.log_posterior_batch(pb, th[:4])
    ok = np.isfinite(want)
    np.testing.assert_allclose(a[:4][ok], want[ok], rtol=1e-6)


def test_guard_takes_a_coarse_radial_sub_grid_away(monkeypatch, legacy_forms):
    """A radial sub-grid far too coarse for the profile (JOXSZ_AG_SUBSAMPLE=8,8,4) is measured by the guard of jx_finalize like the
    other approximations and taken away (no tables to rebuild: the full operator is resident); HipContext says so once; results equal
    those of the context that never had one."""
    import warnings
    from joxsz_amd import datasets
    from joxsz_amd.hip_backend import JoxszTruncationWarning
    pb = _problem(512, 500, seed=19)
    th = datasets.walker_ball(pb, 40, spread=0.03, seed=19)
    monkeypatch.setenv('JOXSZ_AG_SUBSAMPLE', '8,8,4')
    with pytest.warns(JoxszTruncationWarning, match='radial sub-grid'):
        post = _post(pb)
    rs, tr = post.ctx.radial_sampling, post.ctx.truncation
    assert not rs['active'] and rs['removed_by_the_guard'] == 1 and rs['radii_in_use'] == 500
    assert post.ctx.sampling['active'] and tr['retried'] == 0 and 0 <= tr['est_rel_row_err'] <= tr['bound']
    a = post.log_prob(th)
    post.close()
    monkeypatch.setenv('JOXSZ_AG_SUBSAMPLE', '0')
    with warnings.catch_warnings():
        warnings.simplefilter('error', JoxszTruncationWarning)
        ref = _post(pb)
    b = ref.log_prob(th)
    ref.close()
    np.testing.assert_array_equal(a, b)


def test_truncation_guard_speaks_up_when_it_changes_the_tables(legacy_forms):
    """Transfer functions that sit just over the guard's bounds at the default tables (a sharper and a softer normal-cdf
    roll-off than CL J1226.9+3332's, found with scripts/guard_scan.py): jx_finalize first takes the 16-term cap away, then --
    for the softer one -- tightens the cut; HipContext says so once (JoxszTruncationWarning, `truncation['warning']`) with
    the rank it ended at and what that costs; the results hold the 1e-6 bar either way.  The default inputs stay silent."""
    import warnings
    from joxsz_amd import datasets
    from joxsz_amd.hip_backend import JoxszTruncationWarning
    from joxsz_amd.posterior import JoxszPosterior
    with warnings.catch_warnings():
        warnings.simplefilter('error', JoxszTruncationWarning)
        post = JoxszPosterior(datasets.synthetic_problem(S=512, N=500, seed=3), device=0)      # no warning: would raise here
        tr0 = post.ctx.truncation
        post.close()
    assert tr0['warning'] is None and tr0['rank'] == 16 and tr0['rank_above_cut'] in (17, 18) and tr0['cap_removed'] == 0 and tr0['retried'] == 0
    for scale, cap_removed, tightened in ((0.010, 1, False), (0.050, 0, True)):
        pb = datasets.synthetic_problem(S=512, N=500, seed=3, tf_scale=scale)
        with pytest.warns(JoxszTruncationWarning) as rec:
            post = JoxszPosterior(pb, device=0)
        tr = post.ctx.truncation
        assert len(rec) == 1 and tr['warning'] == str(rec[0].message)
        assert tr['cap_removed'] == cap_removed and (tr['retried'] > tr['cap_removed']) == tightened
        if post.ctx.conv_layout['form'] == 'full':                   # the tightening ended in the exact form (cheaper than 30 terms)
            assert tightened and tr['rank'] == 0 and 'full form' in tr['warning']
        else:
            assert tr['rank'] > 16 and ('%d terms' % tr['rank']) in tr['warning']
            assert 0 <= tr['est_rel_row_err'] <= tr['bound'] and 0 <= tr['est_rel_sz_like_err_box'] <= tr['bound_sz_like']
        th = datasets.walker_ball(pb, 12, spread=0.03, seed=3)
        a = post.log_prob(th)
        post.close()
        ref = JoxszPosterior(pb, device=0, conv='rocfft')
        b = ref.log_prob(th)
        ref.close()
        fin = np.isfinite(b)
        assert fin.sum() >= 6 and np.array_equal(np.isfinite(a), fin)
        np.testing.assert_allclose(a[fin], b[fin], rtol=1e-8)


def test_truncation_guard_odd_side_and_measured_transfer_function(monkeypatch, legacy_forms):
    """Odd sides: same guard, same kernels.  The bundled measured transfer function (rough from one wavenumber to the next:
    its weights have nearly full rank) leaves nothing to truncate -- the full form runs and the guard reports so (its
    measurement is then that of the sub-grid of map samples alone)."""
    pb = _problem(513, 500, seed=4)
    post = _post(pb, conv='custom')
    tr = post.ctx.truncation
    assert tr['tol'] == 1e-8 and not tr['retried'] and 0 <= tr['est_rel_row_err'] < 1e-10 and 0 <= tr['est_rel_sz_like_err_box'] <= 1e-8
    post.close()
    monkeypatch.setenv('JOXSZ_TRUNC_BOUND', '1e-18')
    post = _post(pb, conv='custom')
    tr2 = post.ctx.truncation
    if post.ctx.conv_layout['form'] == 'full':                       # (the tightening ended in the exact form)
        assert tr2['retried'] >= 1 and tr2['rank'] == 0
    else:
        assert tr2['retried'] >= 5 and tr2['tol'] <= 1.01e-13 and tr2['rank'] > tr['rank']
    post.close()
    monkeypatch.delenv('JOXSZ_TRUNC_BOUND')
    from joxsz_amd import datasets, setup_host as sh
    import os
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'bundled_inputs.npz'))
    pb = datasets.synthetic_problem(S=257, N=300, seed=4)
    wn, tf = sh.transfer_function(z['wn_as'], z['tf'], approx=False)
    pb.filtering = np.ascontiguousarray(sh.filter_image(wn, tf, 257, 2.))
    th = datasets.walker_ball(pb, 8, spread=0.03, seed=4)
    post = _post(pb)
    assert post.ctx.conv_layout['form'] == 'full'
    tr = post.ctx.truncation
    # (nothing truncated; what the guard measures here is the sub-grid of map samples alone, at rounding level)
    assert post.ctx.sampling['active'] and 0 <= tr['est_rel_row_err'] < 1e-12 and tr['rank'] == 0 and tr['retried'] == 0
    a = post.log_prob(th)
    post.close()
    ref = _post(pb, conv='rocfft')
    b = ref.log_prob(th)
    ref.close()
    fin = np.isfinite(b)
    assert fin.sum() >= 6 and np.array_equal(np.isfinite(a), fin)
    np.testing.assert_allclose(a[fin], b[fin], rtol=1e-10)


def test_radial_grid_beyond_the_abel_kernels_lds():
    """N = 1818 at 1025^2 (joxsz_main.py:24,104: R_b and the step set N; round 3 refused it): the Abel + map kernel cannot hold
    that spline in LDS, but the timed kernels of the contracted route never needed it.  The context comes up without the
    kernel: every singular term above rounding is kept (nothing could measure a truncation), the log-posterior and the profile
    taps -- read off the matrix product's own arrays -- agree with the oracle, and the taps that only the kernel could serve
    (Compton-y map, beam-convolved map) and the rocFFT route say so."""
    from joxsz_amd import datasets
    from joxsz_amd.hip_backend import JoxszHipError
    pb = datasets.synthetic_problem(S=1025, N=1818, seed=8)
    th = datasets.walker_ball(pb, 6, spread=0.03, seed=8)
    with pytest.raises(JoxszHipError, match='unsupported'):
        _post(pb, conv='rocfft')
    post = _post(pb)
    assert post.ctx.conv == 'custom'
    tr = post.ctx.truncation
    assert tr['retried'] == 0 and (tr['rank'] == 0 or tr['tol'] <= 1.01e-13)
    assert not post.ctx.sampling['active'] and post.ctx.sampling['removed_by_the_guard'] == 0     # (nor a sub-grid of map samples: nothing could measure it)
    got = post.log_prob(th)
    pp, y, row = post.stage(th[:2], 'pp'), post.stage(th[:2], 'y'), post.stage(th[:2], 'map_row')
    for tap in ('y_2d', 'conv_2d'):
        with pytest.raises(JoxszHipError, match='unsupported'):
            post.stage(th[:1], tap)
    post.close()
    want = orc.log_posterior_batch(pb, th[:3])
    fin = np.isfinite(want)
    assert fin.sum() >= 2 and np.array_equal(np.isfinite(got[:3]), fin)
    np.testing.assert_allclose(got[:3][fin], want[fin], rtol=1e-9)
    for w in range(2):
        st = orc.sz_stages(pb, orc.pars_dict(pb, th[w]))
        np.testing.assert_allclose(pp[w], st['pp'], rtol=1e-12)
        np.testing.assert_allclose(y[w], st['y'], rtol=1e-10, atol=1e-13 * np.abs(st['y']).max())
        assert np.abs(row[w] - st['map_row']).max() <= 1e-9 * np.abs(st['map_row']).max()
