"""Parity of the HIP path (through the C-ABI) against the CPU oracle and against
the reference's golden vectors.  Needs an MI355X: run with -m gpu."""
import os

import numpy as np
import pytest

from oracle import joxsz_oracle as orc

pytestmark = pytest.mark.gpu

RTOL = 1e-6          # BASELINE.json north_star: <= 1e-6 relative on the log-posterior
RTOL_STAGE = 1e-9    # fp64 end to end: intermediate arrays agree far tighter


def _post(pb, **kw):
    from joxsz_amd.posterior import JoxszPosterior
    return JoxszPosterior(pb, device=0, **kw)


def _relerr(got, want, scale=None):
    want = np.asarray(want)
    s = np.abs(want).max() if scale is None else scale
    return np.abs(np.asarray(got) - want).max() / s


def test_golden_logp(golden):
    """The reference's own log-posteriors (tests/golden) on identical parameter vectors."""
    pb, ref = golden
    post = _post(pb)
    assert post.ctx.conv == 'custom'                  # the reference's odd sides run on the contracted route
    got = post.log_prob(ref['thetas'])
    want = ref['ref_logp']
    post.close()
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), fin)
    assert np.all(got[~fin] == -np.inf)
    np.testing.assert_allclose(got[fin], want[fin], rtol=RTOL)
    assert np.max(np.abs(got[fin] - want[fin]) / np.abs(want[fin])) < 1e-9


def test_golden_stages(golden):
    pb, ref = golden
    post = _post(pb)
    th = ref['thetas']
    fin = np.isfinite(ref['ref_logp'])
    pp = post.stage(th, 'pp')
    bright = post.stage(th, 'bright')
    chisq = post.stage(th, 'chisq')
    parts = post.stage(th, 'parts')
    post.close()
    for k in np.flatnonzero(fin):
        assert _relerr(pp[k], ref['ref_pp'][k]) < 1e-13
        assert _relerr(bright[k], ref['ref_bright'][k]) < RTOL_STAGE
        assert abs(chisq[k] - ref['ref_chisq'][k]) <= RTOL_STAGE * abs(ref['ref_chisq'][k])
        assert abs(parts[k, 0] - ref['ref_xlike'][k]) <= 1e-11 * abs(ref['ref_xlike'][k])
        assert parts[k, 3] == 0
    # rejection reasons: box prior, r_c > r_s, mass veto, mass veto
    rej = parts[~fin, 3].astype(int)
    assert rej[0] & 1 and rej[1] & 4 and rej[2] & 2 and rej[3] & 2


def test_stage_by_stage_vs_oracle(golden_tiny):
    pb, ref = golden_tiny
    post = _post(pb)
    th = ref['thetas'][:5]
    got = {s: post.stage(th, s) for s in ('pp', 'ab', 'y', 'y_2d', 'conv_2d', 'map_row', 'bright', 'chisq', 'tprof', 'xprofs')}
    post.close()
    for k, t in enumerate(th):
        p = orc.pars_dict(pb, t)
        st = orc.sz_stages(pb, p)
        for name in ('pp', 'ab', 'y', 'y_2d', 'conv_2d', 'map_row', 'bright'):
            assert _relerr(got[name][k], st[name]) < RTOL_STAGE, name
        assert abs(got['chisq'][k] - st['chisq']) < RTOL_STAGE * st['chisq']
        assert _relerr(got['tprof'][k], np.append(st['t0'], st['t_prof'])) < 1e-11
        assert _relerr(got['xprofs'][k], orc.calc_profiles(pb, p)) < 1e-12


@pytest.mark.parametrize('S,N,conv', [(31, 40, 'rocfft'), (32, 40, 'rocfft'), (64, 80, 'rocfft'), (64, 80, 'custom'),
                                      (48, 60, 'custom'), (171, 313, 'auto'), (256, 300, 'rocfft'), (256, 300, 'custom'),
                                      (257, 300, 'auto'), (513, 500, 'auto')])
def test_random_walkers_vs_oracle(S, N, conv):
    """Odd (reference-shaped) and even (BASELINE-shaped) map sides, seeded walkers, both
    back ends (rocFFT sequence / contracted route)."""
    from joxsz_amd import datasets
    pb = datasets.synthetic_problem(S=S, N=N, seed=S)
    p0 = orc.pars_dict(pb, datasets.fiducial_theta(pb))
    datasets.fill_data(pb, orc.sz_stages(pb, p0)['bright'], orc.calc_profiles(pb, p0), seed=S)
    W = 24 if S <= 64 else 6
    th = datasets.walker_ball(pb, W, spread=0.05, seed=S)
    th[1, 1] = 9.0                                    # one rejected walker in the batch
    post = _post(pb, conv=conv)
    # every map side with the mirror structure of centdistmat takes the contracted route
    assert post.ctx.conv == ('custom' if conv == 'auto' else conv)
    got = post.log_prob(th)
    post.close()
    want = orc.log_posterior_batch(pb, th)
    fin = np.isfinite(want)
    assert fin.sum() >= W // 2
    assert np.array_equal(np.isfinite(got), fin)
    np.testing.assert_allclose(got[fin], want[fin], rtol=RTOL)


def test_sz_only_and_double_beta():
    from joxsz_amd import datasets
    for kw in (dict(sz_only=True), dict(ne_mode='double')):
        pb = datasets.synthetic_problem(S=48, N=60, seed=5, **kw)
        p0 = orc.pars_dict(pb, datasets.fiducial_theta(pb))
        datasets.fill_data(pb, orc.sz_stages(pb, p0)['bright'], orc.calc_profiles(pb, p0), seed=5)
        th = datasets.walker_ball(pb, 8, spread=0.03, seed=5)
        post = _post(pb)
        got = post.log_prob(th)
        post.close()
        want = orc.log_posterior_batch(pb, th)
        fin = np.isfinite(want)
        assert fin.any() and np.array_equal(np.isfinite(got), fin)
        np.testing.assert_allclose(got[fin], want[fin], rtol=RTOL)


def test_reference_signature(golden_tiny):
    """getLikelihood(vals) -> float with the side effect of joxsz_funcs.py:515-516."""
    pb, ref = golden_tiny
    post = _post(pb)
    v = post.getLikelihood(ref['thetas'][1])
    assert isinstance(v, float)
    np.testing.assert_allclose(v, ref['ref_logp'][1], rtol=RTOL)
    np.testing.assert_allclose(post.thawedParVals(), ref['thetas'][1])
    np.testing.assert_allclose(post.getLikelihood(), ref['ref_logp'][1], rtol=RTOL)      # vals=None: current pars
    np.testing.assert_allclose(post.get_sz_like('bright'), ref['ref_bright'][1], rtol=1e-8, atol=1e-12)
    np.testing.assert_allclose(post.get_sz_like('ll'), ref['ref_ll'][1], rtol=1e-8)
    assert post.getLikelihood(ref['thetas'][10]) == -np.inf
    assert list(post.pool().map(None, ref['thetas'][:3])) == list(post.log_prob(ref['thetas'][:3]))
    post.close()


def test_chunking_and_ragged_batches(golden_tiny):
    """Batches that do not divide the chunk, a batch of one, and an empty batch."""
    pb, ref = golden_tiny
    post = _post(pb, max_batch=4)
    th = np.repeat(ref['thetas'][:5], 3, axis=0)[:13]
    got = post.log_prob(th)
    np.testing.assert_allclose(got, np.repeat(ref['ref_logp'][:5], 3)[:13], rtol=RTOL)
    assert post.log_prob(th[:1]).shape == (1,)
    assert post.log_prob(np.zeros((0, pb.ndim))).shape == (0,)
    post.close()


def test_linearity_full_size():
    """Size-independent property at BASELINE's headline size (S=512, N=500): every
    stage after the pressure profile is linear in it, so the surface-brightness
    profile scales with P_0 and the maps of two walkers add."""
    from joxsz_amd import datasets
    pb = datasets.synthetic_problem(S=512, N=500, seed=0)
    t0 = datasets.fiducial_theta(pb)
    th = np.array([t0, t0, t0])
    th[1, 7] *= 2.0                                   # P_0 doubled: same T shape only if n_e fixed -> use map_row
    th[2, 7] *= 3.0
    post = _post(pb, conv='rocfft', max_batch=8)
    rows_fft = post.stage(th, 'map_row')
    post.close()
    post = _post(pb, conv='custom', max_batch=8)
    rows = post.stage(th, 'map_row')
    y2d = post.stage(th[:2], 'y_2d')
    post.close()
    # (the low-rank form's singular-value cut moves the row by ~1e-11 of its largest entry: absolute, so relatively more in its tail)
    np.testing.assert_allclose(rows, rows_fft, rtol=1e-9, atol=1e-10 * np.abs(rows_fft).max())
    np.testing.assert_allclose(rows[1], 2.0 * rows[0], rtol=1e-10, atol=1e-18)
    np.testing.assert_allclose(rows[2], 3.0 * rows[0], rtol=1e-10, atol=1e-18)
    np.testing.assert_allclose(y2d[1], 2.0 * y2d[0], rtol=1e-12)
    # 8-fold symmetry of the map about the centre pixel (d_mat is symmetric)
    m = y2d[0]
    c = 256
    np.testing.assert_allclose(m[c + 5, c + 9], m[c - 5, c - 9], rtol=1e-14)
    np.testing.assert_allclose(m[c + 5, c + 9], m[c + 9, c + 5], rtol=1e-14)
    assert m[c, c] == m.max()


@pytest.mark.parametrize('S,N,step,fwhm', [(32, 40, 6., 8.5), (64, 80, 2., 18.5)])
def test_contracted_route_stages(S, N, step, fwhm):
    """Contracted route: the Compton-y map and beam-convolved map taps (Abel + map kernel; reference facility) and the
    extracted row against the oracle, and against the rocFFT back end on the same context inputs."""
    from joxsz_amd import datasets
    pb = datasets.synthetic_problem(S=S, N=N, seed=3, step=step, fwhm=fwhm)
    th = datasets.walker_ball(pb, 5, spread=0.04, seed=3)
    out = {}
    for conv in ('custom', 'rocfft'):
        post = _post(pb, conv=conv)
        assert post.ctx.conv == conv
        out[conv] = {s: post.stage(th, s) for s in ('y_2d', 'conv_2d', 'map_row', 'bright', 'chisq')}
        out[conv]['logp'] = post.log_prob(th)
        post.close()
    for k, t in enumerate(th):
        st = orc.sz_stages(pb, orc.pars_dict(pb, t))
        for name in ('y_2d', 'conv_2d', 'map_row', 'bright'):
            assert _relerr(out['custom'][name][k], st[name]) < RTOL_STAGE, name
    for name in ('conv_2d', 'map_row', 'chisq', 'logp'):
        np.testing.assert_allclose(out['custom'][name], out['rocfft'][name], rtol=1e-9, atol=1e-30)


def test_contracted_route_refuses_what_it_cannot_do(golden_tiny):
    """The contracted route needs the mirror structure of a centred distance matrix (what centdistmat builds,
    joxsz_funcs.py:78-88): asked for explicitly without it -- or with a padded length, which only the rocFFT sequence has --
    it refuses; on 'auto' the rocFFT sequence runs, and that one still matches the oracle.  A beam image that is not
    flip-symmetric is no obstacle (the separable terms and the full form make no use of the symmetry)."""
    import copy
    from joxsz_amd.hip_backend import JoxszHipError
    pb, ref = golden_tiny
    bad = copy.deepcopy(pb)
    bad.d_mat = bad.d_mat.copy()
    bad.d_mat[3, 5] *= 1.0001                            # one pixel off the mirror structure
    with pytest.raises(JoxszHipError):
        _post(bad, conv='custom')
    with pytest.raises(JoxszHipError):
        _post(pb, conv='custom', fft_pad=64)
    post = _post(bad)
    assert post.ctx.conv == 'rocfft'
    got = post.log_prob(ref['thetas'][:4])
    post.close()
    np.testing.assert_allclose(got, orc.log_posterior_batch(bad, ref['thetas'][:4]), rtol=RTOL)
    skew = copy.deepcopy(pb)
    skew.beam_2d = skew.beam_2d.copy()
    o = skew.B // 2
    skew.beam_2d[o, o + 1] *= 1.5                        # no longer symmetric under the flips
    post = _post(skew, conv='custom')
    got = post.log_prob(ref['thetas'][:4])
    post.close()
    np.testing.assert_allclose(got, orc.log_posterior_batch(skew, ref['thetas'][:4]), rtol=RTOL)


@pytest.mark.parametrize('S,N', [(65, 80), (171, 313), (257, 300), (513, 500)])
def test_odd_side_custom_route(S, N):
    """Odd map sides -- the only kind the reference itself can run (joxsz_main.py:100-105, joxsz_funcs.py:472-473) -- on
    the contracted route (the same kernels as even sides: nothing in them depends on the parity).  Stage by stage against
    the oracle and against the rocFFT sequence; ragged launches bitwise."""
    from joxsz_amd import datasets
    pb = datasets.synthetic_problem(S=S, N=N, seed=S + 1)
    p0 = orc.pars_dict(pb, datasets.fiducial_theta(pb))
    datasets.fill_data(pb, orc.sz_stages(pb, p0)['bright'], orc.calc_profiles(pb, p0), seed=S)
    th = datasets.walker_ball(pb, 21, spread=0.04, seed=S)
    th[3, 1] = 9.0
    post = _post(pb, conv='custom')
    lay = post.ctx.conv_layout
    assert post.ctx.conv == 'custom' and lay['NU'] == S // 2 + 1 and lay['form'] == 'exact'
    got = post.log_prob(th)
    rows, bright, chisq, y2d, conv = (post.stage(th[:6], s) for s in ('map_row', 'bright', 'chisq', 'y_2d', 'conv_2d'))
    assert np.array_equal(post.log_prob(th), got)
    post.close()
    small = _post(pb, conv='custom', max_batch=8)
    np.testing.assert_array_equal(small.log_prob(th), got)
    small.close()
    ref = _post(pb, conv='rocfft')
    want_fft = ref.log_prob(th)
    ref.close()
    want = orc.log_posterior_batch(pb, th)
    fin = np.isfinite(want)
    assert fin.sum() >= 15 and np.array_equal(np.isfinite(got), fin)
    np.testing.assert_allclose(got[fin], want[fin], rtol=1e-9)
    np.testing.assert_allclose(got[fin], want_fft[fin], rtol=1e-9)
    for k in range(6):
        st = orc.sz_stages(pb, orc.pars_dict(pb, th[k]))
        assert _relerr(y2d[k], st['y_2d']) < RTOL_STAGE
        assert _relerr(conv[k], st['conv_2d']) < RTOL_STAGE
        assert _relerr(rows[k], st['map_row']) < RTOL_STAGE
        assert _relerr(bright[k], st['bright']) < RTOL_STAGE       # (the rejected walker too: nothing is truncated)
        assert abs(chisq[k] - st['chisq']) / 2 < 1e-6 * max(1.0, 1e-3 * st['chisq'])      # absolute near the mode, relative far from it


def test_odd_side_1025():
    """SURVEY 8(d)'s largest odd side (the odd neighbour of BASELINE configs[4]'s 1024^2 map, 1000-point grid) on the
    contracted route, against the oracle and the rocFFT sequence."""
    from joxsz_amd import datasets
    pb = datasets.synthetic_problem(S=1025, N=1000, seed=7)
    p0 = orc.pars_dict(pb, datasets.fiducial_theta(pb))
    datasets.fill_data(pb, orc.sz_stages(pb, p0)['bright'], orc.calc_profiles(pb, p0), seed=7)
    th = datasets.walker_ball(pb, 6, spread=0.03, seed=7)
    th[2, 1] = 9.0
    post = _post(pb, conv='custom')
    assert post.ctx.conv == 'custom' and post.ctx.conv_layout['form'] == 'exact'
    got = post.log_prob(th)
    rows = post.stage(th[:2], 'map_row')
    conv = post.stage(th[:1], 'conv_2d')[0]                    # (round 2 refused this tap at 1025^2; it comes from the rocFFT facility now)
    post.close()
    ref = _post(pb, conv='rocfft')
    want_fft = ref.log_prob(th)
    ref.close()
    want = orc.log_posterior_batch(pb, th)
    fin = np.isfinite(want)
    assert fin.sum() == 5 and np.array_equal(np.isfinite(got), fin)
    np.testing.assert_allclose(got[fin], want[fin], rtol=1e-9)
    np.testing.assert_allclose(got[fin], want_fft[fin], rtol=1e-9)
    for k in range(2):
        st = orc.sz_stages(pb, orc.pars_dict(pb, th[k]))
        assert _relerr(rows[k], st['map_row']) < RTOL_STAGE
        if k == 0:
            assert _relerr(conv, st['conv_2d']) < RTOL_STAGE


@pytest.mark.parametrize('S,N,W', [(64, 80, 3), (171, 313, 70), (512, 500, 130)])
def test_timed_kernels_against_numpy_on_the_oracle_map(S, N, W, monkeypatch):
    """The kernels the log-posterior is timed on, stage by stage (jx_debug_workspace), in the low-rank form: the rows stage 1
    leaves per map column against C^T Q in numpy -- Q the quadrant of the ORACLE's Compton-y map, C the operator the device
    holds -- and the partial rows of the matrix-core product, summed, against G D in numpy, the row tap and the oracle."""
    from joxsz_amd import datasets
    monkeypatch.setenv('JOXSZ_MIX_FORM', 'lowrank')
    pb = datasets.synthetic_problem(S=S, N=N, seed=5)
    th = datasets.walker_ball(pb, W, spread=0.03, seed=5)
    post = _post(pb, conv='custom')
    ctx = post.ctx
    lay = ctx.conv_layout
    prune = ctx.output_pruning
    ctx_sampling = ctx.sampling
    assert lay['form'] == 'lowrank'
    rows = post.stage(th, 'map_row')                           # (runs the profile taps' Abel kernel for the spline arrays)
    D_tap = ctx.workspace('stage1')[:, :, :W].copy()
    lp = post.log_prob(th)                                      # (spline arrays from the matrix product: the timed sequence)
    D = ctx.workspace('stage1')[:, :, :W]                       # [NU][R][walker]
    part = ctx.workspace('partials')[:, :W, :]                  # [ksplit][walker][ldx]
    cf = ctx.workspace('splines')[:, :W, :]                     # [N][walker][2]
    Cm = ctx.workspace('stage1_op')[0]
    Op = ctx.workspace('product_op')
    y_tap = post.stage(th[:2], 'y')
    post.close()
    # (stage 1 evaluates a sub-grid of the quadrant's rows and columns -- jx_get_sampling -- the interpolation to the others is
    # folded into both operators: the numpy statement below is the same with Q restricted to that sub-grid)
    smp = ctx_sampling
    sub = smp['rows']
    NU, R, nrow = len(sub), lay['R'], S - S // 2
    assert smp['rows_of_the_quadrant'] == lay['NU'] and D.shape[0] == NU and (NU < lay['NU']) == smp['active']
    Cm = Cm[:NU, :R]
    G = np.zeros((nrow, NU * R))
    for x in range(nrow):
        G[x] = Op[:NU * R, x & 15, x >> 4]
    c = S // 2
    iy = np.array([c + b if c + b < S else c - b for b in sub])
    for w in (0, W - 1):
        st = orc.sz_stages(pb, orc.pars_dict(pb, th[w]))
        Q = st['y_2d'][np.ix_(iy, iy)]
        want_D = Cm.T @ Q                                       # [R][x']
        scale = np.abs(want_D).max()
        assert np.abs(D[:, :, w].T - want_D).max() < 1e-12 * scale
        assert np.abs(D_tap[:, :, w].T - want_D).max() < 1e-12 * scale
        want_row = G @ D[:, :, w].reshape(-1)                   # kappa = x' * R + j
        # (the timed product computes the outputs the data-radii spline of the tail reads -- whole tiles of 16 -- and no others)
        nout = min(nrow, prune['outputs_computed']) if prune['active'] else nrow
        assert part.shape[2] >= nout and (not prune['active'] or (part.shape[2] < 16 * lay['ntile'] and nout >= prune['outputs_read_by_the_tail']))
        got_row = part[:, w, :nout].sum(axis=0)
        assert _relerr(got_row, want_row[:nout], np.abs(want_row).max()) < 1e-13
        assert _relerr(got_row, rows[w][:nout], np.abs(rows[w]).max()) < 1e-12
        assert _relerr(got_row, st['map_row'][:nout], np.abs(st['map_row']).max()) < RTOL_STAGE
        if w < 2:
            np.testing.assert_allclose(cf[:, w, 0], y_tap[w], rtol=1e-12, atol=1e-14 * np.abs(y_tap[w]).max())
    assert np.isfinite(lp).sum() >= 0.8 * W


def test_lowrank_form_against_full_form(monkeypatch):
    """The two forms of the contracted route on the same inputs: 'full' (one operator row per distinct map sample: exact),
    'tight' (low-rank form with every singular value above rounding), 'default' (cut 1e-8 at sides >= 400) -- against each
    other, the rocFFT sequence and the oracle; ragged walker counts included."""
    from joxsz_amd import datasets
    modes = {'default': {'JOXSZ_MIX_FORM': 'lowrank'}, 'tight': {'JOXSZ_MIX_FORM': 'lowrank', 'JOXSZ_LOWRANK_TOL': '1e-13'},
             # (every distinct sample with ~48 rows per column: the pieces per column are limited by the LDS hand-over, not by the chip)
             'tight_all': {'JOXSZ_MIX_FORM': 'lowrank', 'JOXSZ_LOWRANK_TOL': '1e-13', 'JOXSZ_MIX_SUBSAMPLE': '0'},
             'full': {'JOXSZ_MIX_FORM': 'full'}}
    for S, N, nw in ((256, 300, 6), (512, 500, 37)):
        pb = datasets.synthetic_problem(S=S, N=N, seed=11)
        th = datasets.walker_ball(pb, nw, spread=0.05, seed=11)
        res = {}
        for mode, env in modes.items():
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            post = _post(pb, conv='custom')
            lay = post.ctx.conv_layout
            assert lay['form'] == ('full' if mode == 'full' else 'lowrank') and (lay['rank'] > 0) == (mode != 'full')
            res[mode] = (post.stage(th, 'map_row'), post.stage(th, 'bright'), post.log_prob(th), lay['rank'])
            post.close()
            for k in env:
                monkeypatch.delenv(k)
        ref = _post(pb, conv='rocfft')
        res['rocfft'] = (ref.stage(th, 'map_row'), ref.stage(th, 'bright'), ref.log_prob(th), 0)
        ref.close()
        for mode in ('tight', 'tight_all', 'full'):
            for a, b in zip(res[mode][:2], res['rocfft'][:2]):
                np.testing.assert_allclose(a, b, rtol=1e-10, atol=1e-11 * np.abs(b).max(), err_msg=mode)
            np.testing.assert_allclose(res[mode][2], res['rocfft'][2], rtol=1e-9, err_msg=mode)
        for a, b in zip(res['default'][:2], res['full'][:2]):             # default cut
            np.testing.assert_allclose(a, b, rtol=1e-7, atol=1e-8 * np.abs(b).max())
        np.testing.assert_allclose(res['default'][2], res['full'][2], rtol=1e-9)
        assert res['default'][3] <= res['tight'][3]
        want = orc.log_posterior_batch(pb, th[:8])
        for mode in modes:
            np.testing.assert_allclose(res[mode][2][:8], want, rtol=RTOL, err_msg=mode)
        st = orc.sz_stages(pb, orc.pars_dict(pb, th[0]))
        for mode in modes:
            assert _relerr(res[mode][0][0], st['map_row']) < RTOL_STAGE, mode


def test_rough_transfer_function_takes_the_full_form(legacy_forms):
    """A transfer function whose weights are not low-rank (here: multiplied by uncorrelated noise, symmetrised) makes the
    low-rank form the dearer one: the library takes the full form by itself and still matches the oracle."""
    from joxsz_amd import datasets
    pb = datasets.synthetic_problem(S=128, N=150, seed=21)
    rng = np.random.default_rng(21)
    noise = 1.0 + 0.5 * rng.random((pb.S, pb.S))
    idx = (-np.arange(pb.S)) % pb.S
    noise = 0.25 * (noise + noise[idx][:, idx] + noise[idx] + noise[:, idx])      # symmetric in each wavenumber: real weights
    pb.filtering = np.ascontiguousarray(pb.filtering * noise)
    th = datasets.walker_ball(pb, 5, spread=0.04, seed=21)
    post = _post(pb, conv='custom')
    lay = post.ctx.conv_layout
    # (nothing truncated, too small a map for a sub-grid of its samples: the guard's figure is that of the radial sub-grid of the spline-array product)
    assert lay['form'] == 'full' and lay['rank'] == 0 and not post.ctx.sampling['active'] and 0 <= post.ctx.truncation['est_rel_row_err'] < 1e-12
    got = post.log_prob(th)
    row = post.stage(th[:1], 'map_row')[0]
    post.close()
    want = orc.log_posterior_batch(pb, th)
    assert np.array_equal(np.isfinite(got), np.isfinite(want)) and np.isfinite(want).sum() >= 3
    fin = np.isfinite(want)
    np.testing.assert_allclose(got[fin], want[fin], rtol=RTOL)
    st = orc.sz_stages(pb, orc.pars_dict(pb, th[0]))
    assert _relerr(row, st['map_row']) < RTOL_STAGE


def _measured_problem(S, N):
    """The bundled (measured) beam profile and transfer function -- the reference's own defaults, beam_approx = tf_approx =
    False (joxsz_main.py:59-60) -- on a map of side S (tests/golden/bundled_inputs.npz holds the two data files' columns)."""
    from joxsz_amd import datasets, setup_host as sh
    z = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'bundled_inputs.npz'))
    pb = datasets.synthetic_problem(S=S, N=N, seed=S)
    prof = sh.clip_beam_profile(z['beam_r'], z['beam_prof'])
    beam_2d, _ = sh.beam_image(2., 116.0, approx=False, profile=prof)
    wn, tf = sh.transfer_function(z['wn_as'], z['tf'], approx=False)
    pb.beam_2d = np.ascontiguousarray(beam_2d)
    pb.filtering = np.ascontiguousarray(sh.filter_image(wn, tf, S, 2.))
    return pb.validate()


@pytest.mark.parametrize('S,N', [(512, 512), (513, 513)])
def test_measured_beam_and_transfer_function_at_the_headline_sides(S, N):
    """The reference's default inputs at the headline sides (measured beam image: not separable; measured transfer function: rough
    from one wavenumber to the next): the exact form is the same one operator product as for any other input -- against the oracle
    and the rocFFT sequence."""
    from joxsz_amd import datasets
    pb = _measured_problem(S, N)
    p0 = orc.pars_dict(pb, datasets.fiducial_theta(pb))
    datasets.fill_data(pb, orc.sz_stages(pb, p0)['bright'], orc.calc_profiles(pb, p0), seed=S)
    th = datasets.walker_ball(pb, 20, spread=0.04, seed=S)
    th[2, 1] = 9.0
    post = _post(pb)
    assert post.ctx.conv == 'custom' and post.ctx.conv_layout['form'] == 'exact'
    got = post.log_prob(th)
    rows, chisq = post.stage(th[:4], 'map_row'), post.stage(th, 'chisq')
    post.close()
    ref = _post(pb, conv='rocfft')
    want_fft, chisq_fft = ref.log_prob(th), ref.stage(th, 'chisq')
    ref.close()
    want = orc.log_posterior_batch(pb, th[:6])
    fin = np.isfinite(want_fft)
    assert fin.sum() >= 15 and np.array_equal(np.isfinite(got), fin)
    np.testing.assert_allclose(got[fin], want_fft[fin], rtol=1e-10)
    assert np.max(np.abs(chisq[fin] - chisq_fft[fin])) / 2 < 1e-7
    f6 = np.isfinite(want)
    np.testing.assert_allclose(got[:6][f6], want[f6], rtol=1e-9)
    for k in range(2):
        assert _relerr(rows[k], orc.sz_stages(pb, orc.pars_dict(pb, th[k]))['map_row']) < RTOL_STAGE


_CALLER_STREAM_SCRIPT = r"""
import sys, numpy as np
import torch                                   # first: torch brings its own HIP runtime and must initialise it itself
torch.cuda.init()
sys.path.insert(0, sys.argv[1])
from joxsz_amd import datasets
from joxsz_amd.posterior import JoxszPosterior
pb = datasets.synthetic_problem(S=64, N=80, seed=2)
th = datasets.walker_ball(pb, 9, spread=0.03, seed=2)
post = JoxszPosterior(pb, device=0)
ctx = post.ctx
want = post.log_prob(th)
side = torch.cuda.Stream()
ctx.set_stream(side.cuda_stream)
th_t = torch.from_numpy(th).cuda()
lp_t = torch.full((len(th),), float('nan'), dtype=torch.float64, device='cuda')
torch.cuda.synchronize()
ctx.eval_device(th_t.data_ptr(), len(th), lp_t.data_ptr())
side.synchronize()
assert np.array_equal(lp_t.cpu().numpy(), want), 'caller stream'
ctx.set_stream(None)
assert np.array_equal(post.log_prob(th), want), 'own stream again'
post.close()
print('ok')
"""


def test_caller_stream():
    """jx_set_stream: the evaluation is enqueued on a stream of the caller (here a torch stream, as bench.py does for
    the RCCL gather) and gives the same numbers; NULL returns to the context's own stream.  In a child process: torch
    has to initialise its HIP runtime before this library is loaded, as in bench.py."""
    import importlib.util
    if importlib.util.find_spec('torch') is None:                  # (imported in the child only: torch carries its own HIP and
        pytest.skip('torch not installed')                         #  RCCL builds, which must not meet this process's /opt/rocm ones)
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, '-c', _CALLER_STREAM_SCRIPT, root], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.strip().endswith('ok'), r.stderr[-2000:]


def test_contracted_route_chunking():
    """Walker counts that are no multiple of the kernels' walker tiles (64 lanes in stage 1, 128 per block in the product),
    split over several launch sequences of different sizes, give the numbers of one launch sequence, walker by walker."""
    from joxsz_amd import datasets
    pb = datasets.synthetic_problem(S=128, N=150, seed=13)
    th = datasets.walker_ball(pb, 77, spread=0.04, seed=13)
    post = _post(pb, conv='custom')
    assert post.ctx.conv_layout['form'] == 'exact'
    one = post.log_prob(th)
    post.close()
    for mb in (16, 20, 50):
        post = _post(pb, conv='custom', max_batch=mb)
        got = post.log_prob(th)
        post.close()
        np.testing.assert_array_equal(got, one, err_msg='max_batch=%d' % mb)
    want = orc.log_posterior_batch(pb, th[:6])
    np.testing.assert_allclose(one[:6], want, rtol=RTOL)


def test_device_sampler_against_host_replay():
    """jx_sample (proposals, evaluation, accept/reject and chain on the device) against the same stretch move replayed
    on the host with the same Philox counters and the device's own log-posterior: identical chain and acceptances."""
    from joxsz_amd import datasets
    from joxsz_amd.sampler import DeviceStretchMove, initial_ball
    pb = datasets.synthetic_problem(S=64, N=80, seed=3)
    p0f = orc.pars_dict(pb, datasets.fiducial_theta(pb))
    datasets.fill_data(pb, orc.sz_stages(pb, p0f)['bright'], orc.calc_profiles(pb, p0f), seed=3)
    post = _post(pb)
    p0 = initial_ball(post.log_prob, datasets.fiducial_theta(pb), 28, spread=0.01, rng=np.random.default_rng(3))
    sm = DeviceStretchMove(post, a=2.0, seed=1234)
    chain, lps, nacc = sm.run(p0, 12)
    rchain, rlps, rnacc = sm.replay(p0, 12)
    post.close()
    assert chain.shape == (12, 28, pb.ndim) and np.isfinite(lps).all()
    assert 0 < nacc.sum() < 12 * 28                            # some proposals accepted, some rejected
    np.testing.assert_array_equal(nacc, rnacc)
    np.testing.assert_allclose(chain, rchain, rtol=1e-13, atol=0)
    np.testing.assert_allclose(lps, rlps, rtol=1e-12)
    # a different seed gives a different chain, the same seed the same one
    assert not np.array_equal(chain, DeviceStretchMove(_post(pb), seed=99).run(p0, 12)[0])


def test_stretch_move_inside_the_likelihood_kernels_against_the_separate_kernels(monkeypatch):
    """On the contracted route jx_sample draws a walker's proposal in the per-walker kernel and accepts or rejects it in the tail (five
    launches per half step); JOXSZ_SAMPLE_FUSED=0 keeps the proposal and the acceptance as kernels of their own (seven): the same random
    numbers, the same arithmetic -- identical chains, log-posteriors and acceptance counts, bit for bit; ragged ensembles and several
    chunks per half included."""
    from joxsz_amd import datasets
    from joxsz_amd.sampler import DeviceStretchMove, initial_ball
    pb = datasets.synthetic_problem(S=128, N=150, seed=5)
    p0f = orc.pars_dict(pb, datasets.fiducial_theta(pb))
    datasets.fill_data(pb, orc.sz_stages(pb, p0f)['bright'], orc.calc_profiles(pb, p0f), seed=5)
    res = {}
    for fused in ('1', '0'):
        monkeypatch.setenv('JOXSZ_SAMPLE_FUSED', fused)
        for nw, mb in ((90, 0), (90, 32)):                        # (max_batch 32: two chunks per half step, the second ragged)
            post = _post(pb, **({'max_batch': mb} if mb else {}))
            p0 = initial_ball(post.log_prob, datasets.fiducial_theta(pb), nw, spread=0.01, rng=np.random.default_rng(5))
            res[(fused, mb)] = DeviceStretchMove(post, a=2.0, seed=77).run(p0, 10)
            post.close()
    for mb in (0, 32):
        for a, b in zip(res[('1', mb)], res[('0', mb)]):
            np.testing.assert_array_equal(a, b)
    for a, b in zip(res[('1', 0)], res[('1', 32)]):              # and the chunking does not matter either
        np.testing.assert_array_equal(a, b)
    assert 0 < res[('1', 0)][2].sum() < 10 * 90


def test_per_walker_kernel_as_two_blocks_per_walker(monkeypatch):
    """The timed sequence launches the per-walker work as 2 n blocks of 128 threads -- n for everything but the X-ray side, n for the X-ray side
    alone; the tail adds the two (radial grids up to 640 points) -- since round 5 as two lean roles of a kernel of their own (jx_walker2_kernel:
    every table of a block copied into LDS in one batch), or inside jx_prep_kernel (JOXSZ_PREP_LEAN=0).  Against the one-block form of
    jx_prep_kernel (JOXSZ_PREP_SPLIT=0, what every call with taps runs): the same bits for every walker in all three, the rejected ones (box,
    NaN, r_c > r_s, mass veto, non-positive count rates) included, both density models, ragged chunks; the taps' parts add up to the timed
    value; and the headline shape (two trips of the grid pass per lane)."""
    from joxsz_amd import datasets
    modes = (('one block', {'JOXSZ_PREP_SPLIT': '0'}), ('two blocks in jx_prep_kernel', {'JOXSZ_PREP_LEAN': '0'}), ('two lean roles', {}))

    def run(pb, th, **kw):
        res = {}
        for name, env in modes:
            for k in ('JOXSZ_PREP_SPLIT', 'JOXSZ_PREP_LEAN'):
                monkeypatch.delenv(k, raising=False)
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            post = _post(pb, **kw)
            res[name] = (post.log_prob(th), post.stage(th, 'parts'))
            post.close()
        return res

    for kw in ({}, dict(ne_mode='double')):
        pb = datasets.synthetic_problem(S=128, N=150, seed=13, **kw)
        p0 = orc.pars_dict(pb, datasets.fiducial_theta(pb))
        datasets.fill_data(pb, orc.sz_stages(pb, p0)['bright'], orc.calc_profiles(pb, p0), seed=13)
        th = datasets.walker_ball(pb, 300, spread=0.08, seed=13)
        th[7, 2] = np.nan
        res = run(pb, th, max_batch=128)                              # (three chunks, the last ragged)
        a = res['one block'][0]
        for name, (got, _) in res.items():
            np.testing.assert_array_equal(got, a, err_msg=name)
        fin = np.isfinite(a)
        assert 30 < fin.sum() < 300
        parts = res['two lean roles'][1]
        np.testing.assert_allclose(a[fin], parts[fin, 0] + parts[fin, 1] + parts[fin, 2], rtol=1e-13)
        want = orc.log_posterior_batch(pb, th[:12])
        ok = np.isfinite(want)
        assert np.array_equal(np.isfinite(a[:12]), ok)
        np.testing.assert_allclose(a[:12][ok], want[ok], rtol=1e-9)
    pb = datasets.synthetic_problem(S=512, N=500, seed=12)
    th = datasets.walker_ball(pb, 200, spread=0.05, seed=12)
    res = run(pb, th)
    for name, (got, _) in res.items():
        np.testing.assert_array_equal(got, res['one block'][0], err_msg=name)
    assert np.isfinite(res['one block'][0]).sum() >= 100


def test_largest_config_shape():
    """BASELINE configs[4] shape (S=1024, N=1000): two walkers against the oracle, both back ends."""
    from joxsz_amd import datasets
    pb = datasets.synthetic_problem(S=1024, N=1000, seed=4)
    th = datasets.walker_ball(pb, 2, spread=0.02, seed=4)
    want = orc.log_posterior_batch(pb, th)
    assert np.isfinite(want).all()
    for conv in ('custom', 'rocfft'):
        post = _post(pb, conv=conv, max_batch=2)
        got = post.log_prob(th)
        post.close()
        np.testing.assert_allclose(got, want, rtol=RTOL)


@pytest.mark.parametrize('fwhm,B', [(6.5, 19), (9.0, 27), (21.0, 63)])
def test_other_beam_widths(fwhm, B):
    """Other beam widths (B = 19, 27, 63): nothing in the contracted route is specialised to one; the beam-convolved-map tap
    goes through the context's reference facility.  All against the oracle."""
    from joxsz_amd import datasets
    pb = datasets.synthetic_problem(S=128, N=150, seed=11, fwhm=fwhm)
    assert pb.B == B
    p0 = orc.pars_dict(pb, datasets.fiducial_theta(pb))
    datasets.fill_data(pb, orc.sz_stages(pb, p0)['bright'], orc.calc_profiles(pb, p0), seed=11)
    th = datasets.walker_ball(pb, 6, spread=0.04, seed=11)
    post = _post(pb, conv='custom')
    got, conv = post.log_prob(th), post.stage(th[:2], 'conv_2d')
    post.close()
    want = orc.log_posterior_batch(pb, th)
    fin = np.isfinite(want)
    assert fin.sum() >= 4 and np.array_equal(np.isfinite(got), fin)
    np.testing.assert_allclose(got[fin], want[fin], rtol=RTOL)
    st = orc.sz_stages(pb, orc.pars_dict(pb, th[0]))
    assert _relerr(conv[0], st['conv_2d']) < RTOL_STAGE


def test_full_size_determinism_and_permutation():
    """Size-independent properties at the headline size (S=512, N=500, 1024+ walkers): two
    evaluations are bitwise equal (partials are summed in a fixed order, no float atomics), a
    walker's value does not depend on its position in the batch or on the chunking, and the
    scalar entry point agrees bitwise with the batched one."""
    from joxsz_amd import datasets
    pb = datasets.synthetic_problem(S=512, N=500, seed=0)
    th = datasets.walker_ball(pb, 1100, spread=0.02, seed=21)     # not a multiple of the chunk
    post = _post(pb)
    a = post.log_prob(th)
    b = post.log_prob(th)
    perm = np.random.default_rng(0).permutation(len(th))
    c = post.log_prob(th[perm])
    one = post.getLikelihood(th[17])
    post.close()
    assert np.isfinite(a).sum() > 1000
    np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(c, a[perm])
    assert one == a[17] or (np.isinf(one) and np.isinf(a[17]))
    small = _post(pb, max_batch=96)
    d = small.log_prob(th[:300])
    small.close()
    np.testing.assert_array_equal(d, a[:300])


def test_prep_log_form_against_pow_form(monkeypatch):
    """The prep kernel evaluates the pressure and density profiles through exp/log of their exponents (default) or with
    the pow() calls of joxsz_funcs.py:275-287, 375-395 (JOXSZ_PREP_POW=1): same T profile, X-ray profiles, veto and
    log-posterior, both density modes."""
    from joxsz_amd import datasets
    for kw in (dict(), dict(ne_mode='double')):
        pb = datasets.synthetic_problem(S=64, N=80, seed=6, **kw)
        p0 = orc.pars_dict(pb, datasets.fiducial_theta(pb))
        datasets.fill_data(pb, orc.sz_stages(pb, p0)['bright'], orc.calc_profiles(pb, p0), seed=6)
        th = datasets.walker_ball(pb, 40, spread=0.08, seed=6)
        out = {}
        for form in ('log', 'pow'):
            if form == 'pow':
                monkeypatch.setenv('JOXSZ_PREP_POW', '1')
            else:
                monkeypatch.delenv('JOXSZ_PREP_POW', raising=False)
            post = _post(pb)
            out[form] = (post.log_prob(th), post.stage(th, 'tprof'), post.stage(th, 'xprofs'), post.stage(th, 'parts'))
            post.close()
        monkeypatch.delenv('JOXSZ_PREP_POW', raising=False)
        fin = np.isfinite(out['pow'][0])
        assert fin.sum() > 10 and (~fin).sum() > 0                       # the wide ball has vetoed walkers too
        assert np.array_equal(np.isfinite(out['log'][0]), fin)
        np.testing.assert_array_equal(out['log'][3][:, 3], out['pow'][3][:, 3])      # same rejection reasons
        # (the pressure profile of the SZ side comes from this kernel too: the two forms differ by ~1e-14 there, which the
        #  64^2 case -- a small difference of large terms -- shows as ~1e-10 of the log-posterior)
        np.testing.assert_allclose(out['log'][0][fin], out['pow'][0][fin], rtol=1e-9)
        np.testing.assert_allclose(out['log'][1][fin], out['pow'][1][fin], rtol=1e-12)
        np.testing.assert_allclose(out['log'][2][fin], out['pow'][2][fin], rtol=1e-12)


@pytest.mark.parametrize('S,N,W', [(64, 80, 5), (256, 300, 70), (171, 313, 33), (512, 500, 130)])
def test_spline_arrays_matrix_product_against_abel_kernel(S, N, W, monkeypatch, legacy_forms):
    """Contracted route: the spline ordinates and moments (y_k, M_k) of a launch come from one matrix product on the matrix
    cores (``jx_abel_gemm_kernel``: Abel weights, Compton-y scale and spline moments folded into one constant operator;
    joxsz_funcs.py:457-460), walker-minor.  Against the Abel kernel's own phases 1-3 (JOXSZ_ABEL_GEMM=0, which also serves
    the 'ab' / 'y' stage taps that are held to the oracle above): arrays, log-posterior, rejections."""
    from joxsz_amd import datasets
    pb = datasets.synthetic_problem(S=S, N=N, seed=S)
    th = datasets.walker_ball(pb, W, spread=0.03, seed=S)     # ragged: not a multiple of the 32 walkers of a block
    th[1, 1] = 9.0
    res = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('JOXSZ_ABEL_GEMM', mode)
        post = _post(pb, conv='custom')
        lp = post.log_prob(th)
        cf = post.ctx.workspace('splines')[:, :W, :]               # [N][walker][(y, M)]
        y_tap = post.stage(th, 'y')
        post.close()
        res[mode] = (lp, np.ascontiguousarray(cf.transpose(1, 0, 2)).reshape(W, 2 * N), y_tap)
    a, b = res['0'], res['1']
    fin = np.isfinite(a[0])
    assert not fin[1] and np.array_equal(np.isfinite(b[0]), fin)
    ya, yb, ma, mb = a[1][fin, 0:2 * N:2], b[1][fin, 0:2 * N:2], a[1][fin, 1:2 * N:2], b[1][fin, 1:2 * N:2]
    np.testing.assert_array_equal(ya, a[2][fin])                                    # the kernel's ordinates are the 'y' tap
    assert np.max(np.abs(ya - yb) / np.abs(ya).max(axis=1, keepdims=True)) < 1e-13
    assert np.max(np.abs(ma - mb) / np.abs(ma).max(axis=1, keepdims=True)) < 5e-12  # (second differences: cancellation)
    np.testing.assert_allclose(b[0][fin], a[0][fin], rtol=1e-9)


@pytest.mark.parametrize('S,N,usplit', [(512, 500, '2'), (512, 500, '1'), (513, 500, '3')])
def test_stage_1_on_the_matrix_cores_against_the_vector_unit_kernel(S, N, usplit, monkeypatch, legacy_forms):
    """The opt-in form of stage 1 (JOXSZ_MIX_MFMA=1: jx_rowmix_mfma_kernel, the samples of four rows through LDS into
    v_mfma_f64_16x16x4, operator in LDS) against the default jx_rowmix_kernel on the same tables: the rows kept per map column
    (work buffer 'stage1') to rounding, the log-posterior far inside the 1e-6 bar, one, two and three pieces per column (the
    piece lengths 257, 128 + 129, 85 + 86 + 86 exercise full and partial groups of four)."""
    from joxsz_amd import datasets
    pb = datasets.synthetic_problem(S=S, N=N, seed=21)
    th = datasets.walker_ball(pb, 200, spread=0.03, seed=21)
    monkeypatch.setenv('JOXSZ_MIX_USPLIT', usplit)
    out = {}
    for mfma in ('0', '1'):
        monkeypatch.setenv('JOXSZ_MIX_MFMA', mfma)
        post = _post(pb)
        assert post.ctx.conv_layout['form'] == 'lowrank' and post.ctx.conv_layout['R'] <= 16
        assert post.ctx.truncation['stage1_on_matrix_cores'] == (mfma == '1')
        lp = post.log_prob(th)
        d = post.ctx.workspace('stage1')[:, :post.ctx.conv_layout['R'], :200].copy()
        out[mfma] = (lp, d, post.ctx.truncation)
        post.close()
    (a, da, tra), (b, db, trb) = out['0'], out['1']
    fin = np.isfinite(a)
    assert fin.sum() >= 150 and np.array_equal(np.isfinite(b), fin)
    assert np.abs(db - da).max() <= 1e-12 * np.abs(da).max()
    np.testing.assert_allclose(b[fin], a[fin], rtol=1e-10)
    assert abs(trb['est_rel_sz_like_err_box'] - tra['est_rel_sz_like_err_box']) <= 1e-10
