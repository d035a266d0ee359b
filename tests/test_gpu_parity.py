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
    assert post.ctx.conv == 'custom'                  # the reference's odd sides run on the hand-written route
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
    convolution back ends (rocFFT sequence / hand-written mixed-domain passes)."""
    from joxsz_amd import datasets
    pb = datasets.synthetic_problem(S=S, N=N, seed=S)
    p0 = orc.pars_dict(pb, datasets.fiducial_theta(pb))
    datasets.fill_data(pb, orc.sz_stages(pb, p0)['bright'], orc.calc_profiles(pb, p0), seed=S)
    W = 24 if S <= 64 else 6
    th = datasets.walker_ball(pb, W, spread=0.05, seed=S)
    th[1, 1] = 9.0                                    # one rejected walker in the batch
    post = _post(pb, conv=conv)
    # odd sides take the hand-written route as well (real-space transfer-function step) once the padded length fits
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
    # (the default route's singular-value cut moves the row by ~1e-11 of its largest entry: absolute, so relatively more in its tail)
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
def test_custom_conv_stages(S, N, step, fwhm):
    """Hand-written convolution passes: beam-convolved map and extracted row against the
    oracle, and against the rocFFT back end on the same context inputs."""
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


def test_custom_conv_refuses_what_it_cannot_do(golden_tiny):
    """The hand-written route needs a flip-symmetric beam image (what mybeam builds, joxsz_funcs.py:46-76) and the
    library's own padded length: asked for explicitly with anything else it refuses, on 'auto' it takes the rocFFT
    sequence -- and that one still matches the oracle."""
    import copy
    from joxsz_amd.hip_backend import JoxszHipError
    pb, ref = golden_tiny
    pb = copy.deepcopy(pb)
    pb.beam_2d = pb.beam_2d.copy()
    o = pb.B // 2
    pb.beam_2d[o, o + 1] *= 1.5                        # no longer symmetric under the flips
    with pytest.raises(JoxszHipError):
        _post(pb, conv='custom')
    with pytest.raises(JoxszHipError):
        _post(golden_tiny[0], conv='custom', fft_pad=64)
    post = _post(pb)
    assert post.ctx.conv == 'rocfft'
    got = post.log_prob(ref['thetas'][:4])
    post.close()
    want = orc.log_posterior_batch(pb, ref['thetas'][:4])
    np.testing.assert_allclose(got, want, rtol=RTOL)


@pytest.mark.parametrize('S,N', [(65, 80), (171, 313), (257, 300), (513, 500)])
def test_odd_side_custom_route(S, N):
    """Odd map sides -- the only kind the reference itself can run (joxsz_main.py:100-105, joxsz_funcs.py:472-473) -- on
    the hand-written route: rows from the spline, FIR + job combination as matrix products, combined rows back to real
    space, transfer function as real-space circular kernels.  Stage by stage against the oracle and against the rocFFT
    sequence; ragged launches bitwise."""
    from joxsz_amd import datasets
    pb = datasets.synthetic_problem(S=S, N=N, seed=S + 1)
    p0 = orc.pars_dict(pb, datasets.fiducial_theta(pb))
    datasets.fill_data(pb, orc.sz_stages(pb, p0)['bright'], orc.calc_profiles(pb, p0), seed=S)
    th = datasets.walker_ball(pb, 21, spread=0.04, seed=S)
    th[3, 1] = 9.0
    post = _post(pb, conv='custom')
    lay = post.ctx.conv_layout
    assert post.ctx.conv == 'custom' and lay['fused'] == 1 and lay['rank'] > 0
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
        # (the rejected walker's conversion factors grow by 36 orders of magnitude towards the edge, where the row has decayed by
        #  four: what the singular-value cut leaves there, 1e-11 of the row's maximum, is then all that 'bright' shows)
        assert _relerr(bright[k], st['bright']) < (RTOL_STAGE if fin[k] else 1e-6)
        assert abs(chisq[k] - st['chisq']) / 2 < 1e-6 * max(1.0, 1e-3 * st['chisq'])      # absolute near the mode, relative far from it


def test_odd_side_1025():
    """SURVEY 8(d)'s largest odd side (the odd neighbour of BASELINE configs[4]'s 1024^2 map, 1000-point grid) on the
    hand-written route, against the oracle and the rocFFT sequence."""
    from joxsz_amd import datasets
    pb = datasets.synthetic_problem(S=1025, N=1000, seed=7)
    p0 = orc.pars_dict(pb, datasets.fiducial_theta(pb))
    datasets.fill_data(pb, orc.sz_stages(pb, p0)['bright'], orc.calc_profiles(pb, p0), seed=7)
    th = datasets.walker_ball(pb, 6, spread=0.03, seed=7)
    th[2, 1] = 9.0
    post = _post(pb, conv='custom')
    assert post.ctx.conv == 'custom' and post.ctx.conv_layout['fused'] == 1
    got = post.log_prob(th)
    rows = post.stage(th[:2], 'map_row')
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
        assert _relerr(rows[k], orc.sz_stages(pb, orc.pars_dict(pb, th[k]))['map_row']) < RTOL_STAGE


def test_custom_conv_with_and_without_row_symmetry(monkeypatch):
    """The row bookkeeping (distinct map rows, conv jobs) and the real-spectrum form of x-symmetric
    rows must not change any number: identity tables (JOXSZ_CONV_NOSYM=1), mirrored rows with complex
    spectra (JOXSZ_CONV_XSYM=0) and the default (both symmetries) against each other and the oracle."""
    from joxsz_amd import datasets
    modes = {'full': {}, 'rows_only': {'JOXSZ_CONV_XSYM': '0'}, 'none': {'JOXSZ_CONV_NOSYM': '1'}}
    monkeypatch.setenv('JOXSZ_LOWRANK_TOL', '1e-13')          # bookkeeping test: no truncation beyond rounding in any mode
    for S, N, fwhm in ((64, 80, 18.5), (256, 300, 18.5), (512, 300, 9.0)):
        pb = datasets.synthetic_problem(S=S, N=N, seed=9, fwhm=fwhm)
        th = datasets.walker_ball(pb, 4, spread=0.04, seed=9)
        res = {}
        for mode, env in modes.items():
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            post = _post(pb, conv='custom')
            res[mode] = (post.stage(th, 'conv_2d'), post.stage(th, 'map_row'), post.log_prob(th))
            post.close()
            for k in env:
                monkeypatch.delenv(k)
        for mode in ('rows_only', 'none'):
            for a, b in zip(res['full'][:2], res[mode][:2]):
                np.testing.assert_allclose(a, b, rtol=1e-11, atol=1e-13 * np.abs(b).max(), err_msg=mode)
            np.testing.assert_allclose(res['full'][2], res[mode][2], rtol=1e-9, err_msg=mode)     # chi^2 amplifies round-off
        st = orc.sz_stages(pb, orc.pars_dict(pb, th[0]))
        assert _relerr(res['full'][0][0], st['conv_2d']) < RTOL_STAGE
        assert _relerr(res['full'][1][0], st['map_row']) < RTOL_STAGE


@pytest.mark.parametrize('S,N', [(64, 80), (512, 500)])
def test_conv_work_buffers_against_numpy(S, N, monkeypatch):
    """Each hand-written pass on its own (jx_debug_workspace): row spectra, FIR output and the
    column-0 terms against numpy FFTs of the Compton-y map the same evaluation produced.  (The route
    with separate kernels, JOXSZ_FUSED=0: the fused route keeps these intermediates walker-minor.)"""
    from joxsz_amd import datasets
    monkeypatch.setenv('JOXSZ_FUSED', '0')
    pb = datasets.synthetic_problem(S=S, N=N, seed=5)
    th = datasets.walker_ball(pb, 2, spread=0.03, seed=5)
    post = _post(pb, conv='custom', max_batch=2)
    y2d = post.stage(th, 'y_2d')
    post.log_prob(th)
    ctx = post.ctx
    y, quad = ctx.workspace('y_map')
    if quad:                                                  # only the distinct pixels (|iy-c|, |ix-c|) are stored
        idx = np.abs(np.arange(S) - S // 2)
        y = y[:, idx][:, :, idx]
    np.testing.assert_array_equal(y, y2d)
    Y, xsym = ctx.workspace('row_spectra')
    C, _ = ctx.workspace('fir_rows')
    jrow = ctx.workspace('job_rows')[0][0, :, 0]
    umap = ctx.workspace('row_index')[0][0, :, 0]
    assert xsym, 'default mode for a mirror-symmetric d_mat and this beam width'
    col0, _ = ctx.workspace('col0')
    post.close()
    P = ctx.conv_layout['P']; Ph = P // 2 + 1; c = S // 2; B = pb.B; o = (B - 1) // 2
    Y = Y[:, :, :Ph]; C = C[:, :, :Ph]                      # rows are padded to whole cache lines
    phase = np.exp(2j * np.pi * np.arange(Ph) * c / P)
    beam = np.asarray(pb.beam_2d, float)
    # beam spectrum along x per row offset d = -o..o (centred: column o is x offset 0)
    bhat = np.fft.rfft(np.roll(np.pad(beam, ((0, 0), (0, P - B))), -o, axis=1), axis=1) * pb.step ** 2 / P
    first = np.array([np.nonzero(umap == u)[0][0] for u in range(Y.shape[1])])
    for w in range(2):
        m0 = y[w].copy(); m0[:, 0] = 0.0                      # the unpaired column travels separately
        spec = np.fft.rfft(m0, n=P, axis=1)
        want = spec[first] * phase
        scale = np.abs(want).max()
        assert np.abs(want.imag).max() < 1e-13 * scale
        assert np.abs(Y[w] - want.real).max() < 1e-13 * scale
        conv = np.zeros((len(jrow), Ph), complex)
        for q, r in enumerate(jrow):
            for d in range(-o, o + 1):
                if 0 <= r - d < S:
                    conv[q] += bhat[o + d] * spec[r - d]
        want_c = conv * phase
        scale = np.abs(want_c).max()
        assert np.abs(want_c.imag).max() < 1e-12 * scale
        assert np.abs(C[w][:len(jrow)] - want_c.real).max() < 1e-12 * scale
        want0 = np.zeros((len(jrow), o + 1))
        for q, r in enumerate(jrow):
            for d in range(-o, o + 1):
                if 0 <= r - d < S:
                    want0[q] += pb.step ** 2 * beam[o + d, o:] * y[w][r - d, 0]
        assert np.abs(col0[w].T - want0).max() < 1e-13 * np.abs(want0).max()       # stored [x][job]


def test_lowrank_weights_against_full_weights(monkeypatch):
    """The transfer-function weights in low-rank form (truncation at 1e-13 of the largest singular value):
    'fused' = FIR + job combination as one matrix product per column on walker-minor row spectra (default),
    'lowrank' = FIR kernel, then the combination (JOXSZ_FUSED=0), 'full' = one pass-3 row per job
    (JOXSZ_LOWRANK=0).  All three against each other and against the oracle; ragged walker counts included."""
    from joxsz_amd import datasets
    # 'tight': every singular value above rounding (1e-13); the default cut is 1e-8 at sides >= 400, 1e-13 below
    modes = {'fused': {}, 'tight': {'JOXSZ_LOWRANK_TOL': '1e-13'}, 'lowrank': {'JOXSZ_FUSED': '0', 'JOXSZ_LOWRANK_TOL': '1e-13'},
             'full': {'JOXSZ_LOWRANK': '0'}}
    for S, N, nw in ((256, 300, 6), (512, 500, 37)):
        pb = datasets.synthetic_problem(S=S, N=N, seed=11)
        th = datasets.walker_ball(pb, nw, spread=0.05, seed=11)
        res = {}
        for mode, env in modes.items():
            for k, v in env.items():
                monkeypatch.setenv(k, v)
            post = _post(pb, conv='custom')
            lay = post.ctx.conv_layout
            assert (lay['rank'] > 0) == (mode != 'full') and lay['rank'] < lay['NJ'] // 2
            assert bool(lay['fused']) == (mode in ('fused', 'tight'))
            res[mode] = (post.stage(th, 'map_row'), post.stage(th, 'bright'), post.log_prob(th), lay['rank'])
            post.close()
            for k in env:
                monkeypatch.delenv(k)
        for mode in ('tight', 'lowrank'):
            for a, b in zip(res[mode][:2], res['full'][:2]):
                np.testing.assert_allclose(a, b, rtol=1e-10, atol=1e-11 * np.abs(b).max(), err_msg=mode)
            np.testing.assert_allclose(res[mode][2], res['full'][2], rtol=1e-9, err_msg=mode)
        for a, b in zip(res['fused'][:2], res['full'][:2]):               # default cut
            np.testing.assert_allclose(a, b, rtol=1e-7, atol=1e-8 * np.abs(b).max())
        np.testing.assert_allclose(res['fused'][2], res['full'][2], rtol=1e-9)
        assert res['fused'][3] <= res['tight'][3]
        want = orc.log_posterior_batch(pb, th[:8])
        np.testing.assert_allclose(res['fused'][2][:8], want, rtol=RTOL)
        st = orc.sz_stages(pb, orc.pars_dict(pb, th[0]))
        assert _relerr(res['fused'][0][0], st['map_row']) < RTOL_STAGE


def test_rough_transfer_function_falls_back_to_one_row_per_job():
    """A transfer function whose weights are not low-rank (here: multiplied by uncorrelated noise, symmetrised) must
    switch the low-rank / fused routes off by itself and still match the oracle."""
    from joxsz_amd import datasets
    pb = datasets.synthetic_problem(S=128, N=150, seed=21)
    rng = np.random.default_rng(21)
    noise = 1.0 + 0.5 * rng.random((pb.S, pb.S))
    idx = (-np.arange(pb.S)) % pb.S
    noise = 0.5 * (noise + noise[idx][:, idx])               # keep the filter symmetric under k -> -k (real weights)
    pb.filtering = np.ascontiguousarray(pb.filtering * noise)
    th = datasets.walker_ball(pb, 5, spread=0.04, seed=21)
    post = _post(pb, conv='custom')
    lay = post.ctx.conv_layout
    assert lay['rank'] == 0 and lay['fused'] == 0 and lay['xsym'] == 1
    got = post.log_prob(th)
    row = post.stage(th[:1], 'map_row')[0]
    post.close()
    want = orc.log_posterior_batch(pb, th)
    assert np.array_equal(np.isfinite(got), np.isfinite(want)) and np.isfinite(want).sum() >= 3
    fin = np.isfinite(want)
    np.testing.assert_allclose(got[fin], want[fin], rtol=RTOL)
    st = orc.sz_stages(pb, orc.pars_dict(pb, th[0]))
    assert _relerr(row, st['map_row']) < RTOL_STAGE


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


def test_fused_route_chunking():
    """Walker counts that are no multiple of the GEMM's 16-walker tiles, split over several launch sequences of
    different sizes, give the numbers of one launch sequence, walker by walker."""
    from joxsz_amd import datasets
    pb = datasets.synthetic_problem(S=128, N=150, seed=13)
    th = datasets.walker_ball(pb, 77, spread=0.04, seed=13)
    post = _post(pb, conv='custom')
    assert post.ctx.conv_layout['fused'] == 1
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
    """Beam widths with (B=27) and without (B=19, 63) a register-window FIR instance: the log-posterior goes through the
    fused route either way, the beam-convolved-map tap through the register FIR or the plain real-array FIR.  All
    against the oracle."""
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
def test_spline_arrays_matrix_product_against_abel_kernel(S, N, W, monkeypatch):
    """Default route: the spline ordinates and moments (y_k, M_k) of a launch come from one matrix product on the matrix
    cores (``jx_abel_gemm_kernel``: Abel weights, Compton-y scale and spline moments folded into one constant operator;
    joxsz_funcs.py:457-460).  Against the Abel kernel's own phases 1-3 (JOXSZ_ABEL_GEMM=0, which also serves the
    'ab' / 'y' stage taps that are held to the oracle above): arrays, zero padding, log-posterior, rejections."""
    from joxsz_amd import datasets
    pb = datasets.synthetic_problem(S=S, N=N, seed=S)
    th = datasets.walker_ball(pb, W, spread=0.03, seed=S)     # ragged: not a multiple of the 32 walkers of a block
    th[1, 1] = 9.0
    res = {}
    for mode in ('0', '1'):
        monkeypatch.setenv('JOXSZ_ABEL_GEMM', mode)
        post = _post(pb, conv='custom')
        lp = post.log_prob(th)
        cf, _ = post.ctx.workspace('coefs')
        y_tap = post.stage(th, 'y')
        post.close()
        res[mode] = (lp, cf[:W, 0, :].copy(), y_tap)
    a, b = res['0'], res['1']
    fin = np.isfinite(a[0])
    assert not fin[1] and np.array_equal(np.isfinite(b[0]), fin)
    ya, yb, ma, mb = a[1][fin, 0:2 * N:2], b[1][fin, 0:2 * N:2], a[1][fin, 1:2 * N:2], b[1][fin, 1:2 * N:2]
    np.testing.assert_array_equal(ya, a[2][fin])                                    # the kernel's ordinates are the 'y' tap
    assert np.max(np.abs(ya - yb) / np.abs(ya).max(axis=1, keepdims=True)) < 1e-13
    assert np.max(np.abs(ma - mb) / np.abs(ma).max(axis=1, keepdims=True)) < 5e-12  # (second differences: cancellation)
    assert np.all(b[1][:, 2 * N:] == 0)                                             # the slots behind the last knot stay zero
    np.testing.assert_allclose(b[0][fin], a[0][fin], rtol=1e-9)
