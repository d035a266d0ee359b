"""The transforms of the literal route (conv = 'rocfft': joxsz_funcs.py:464-467 executed as written): hand-written rows and columns
(csrc/jx_fft.hpp, default wherever both sides are 2^a 3^b 5^c), hand-written columns beside rocFFT's batched row plans
(JOXSZ_FFT_ROWS=rocfft), and rocFFT's own 2-D plans (JOXSZ_FFT_COLUMNS=rocfft) must give the same convolved map, the same extracted row
and the same log-posterior to rounding; the route as a whole is held to the oracle in tests/test_gpu_parity.py at sides that take the
hand-written transforms (32, 64, 256 ...) and sides that cannot (31, 171)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

MODES = {'rocfft': {'FFT_COLUMNS': 'rocfft'}, 'columns': {'FFT_ROWS': 'rocfft'}, 'all': {}}


def _run(pb, th, opts, **kw):
    from joxsz_amd.posterior import JoxszPosterior
    post = JoxszPosterior(pb, device=0, conv='rocfft', options=opts, **kw)
    out = (post.ctx.eval_stage(th, 'conv_2d'), post.ctx.eval_stage(th, 'map_row'), post.log_prob(th), post.ctx.fft_info())
    post.close()
    return out


# sides: powers of two, 2^a 3^b 5^c mixes, odd smooth sides (row pairs straddle walkers; odd row length), the two bench sides
@pytest.mark.parametrize('S,N,W', [(32, 40, 5), (48, 60, 4), (64, 80, 7), (75, 90, 3), (96, 120, 4), (135, 160, 3), (128, 150, 9), (250, 300, 3),
                                   (256, 300, 6), (360, 400, 2), (512, 500, 5), (640, 700, 2), (1024, 1000, 3)])
def test_hand_written_transforms_against_rocfft_plans(S, N, W):
    from joxsz_amd import datasets
    pb = datasets.synthetic_problem(S=S, N=N, seed=3)
    th = np.ascontiguousarray(datasets.walker_ball(pb, W, spread=0.02, seed=4))
    ref = _run(pb, th, MODES['rocfft'], max_batch=4)          # (max_batch 4: more than one launch, and a last one of another size)
    assert ref[3]['built'] and ref[3]['columns'] == 'rocfft' and ref[3]['rows'] == 'rocfft'
    fin = np.isfinite(ref[2])
    assert fin.any()
    for mode in ('columns', 'all'):
        got = _run(pb, th, MODES[mode], max_batch=4)
        info = got[3]
        assert info['columns'] == 'custom' and info['rows'] == ('custom' if mode == 'all' else 'rocfft'), info
        assert int(np.prod(info['radices_padded'])) == info['fft_pad'] and int(np.prod(info['radices_window'])) == S, info
        assert len(info['radices_padded']) <= 4 and info['ld_padded'] % 8 == 0 and info['ld_window'] % 8 == 0, info
        assert np.max(np.abs(got[0] - ref[0])) <= 1e-13 * np.max(np.abs(ref[0])), mode
        assert np.max(np.abs(got[1] - ref[1])) <= 1e-13 * np.max(np.abs(ref[1])), mode
        assert np.array_equal(np.isfinite(got[2]), fin)
        # (sides below 128: the data radii reach beyond the map, the tail's spline extrapolates and amplifies the row's rounding, as in test_gpu_exact)
        np.testing.assert_allclose(got[2][fin], ref[2][fin], rtol=1e-11 if S >= 128 else 1e-9, err_msg=mode)


@pytest.mark.parametrize('S,N', [(31, 40), (171, 313), (77, 90)])
def test_sides_with_other_prime_factors_stay_on_rocfft_plans(S, N):
    from joxsz_amd import datasets
    pb = datasets.synthetic_problem(S=S, N=N, seed=3)
    th = np.ascontiguousarray(datasets.walker_ball(pb, 3, spread=0.02, seed=4))
    got = _run(pb, th, {})
    assert got[3]['built'] and got[3]['columns'] == 'rocfft' and got[3]['rows'] == 'rocfft', got[3]
    assert np.isfinite(got[2]).any()


def test_reference_facility_of_the_exact_form_uses_the_same_transforms():
    """jx_audit's reference (the literal sequence inside a contracted-route context) runs on the hand-written transforms too; the exact form
    still agrees with it to 1e-12."""
    from joxsz_amd import datasets
    from joxsz_amd.posterior import JoxszPosterior
    pb = datasets.synthetic_problem(S=256, N=300, seed=0)
    th = np.ascontiguousarray(datasets.walker_ball(pb, 24, spread=0.02, seed=1))
    post = JoxszPosterior(pb, device=0)
    assert post.ctx.fft_info() == {'built': False}
    a = post.ctx.audit(th)
    info = post.ctx.fft_info()
    assert info['built'] and info['columns'] == 'custom' and info['rows'] == 'custom', info
    assert a['walkers_compared'] > 0 and a['max_rel_row_diff'] <= 1e-12 and a['max_abs_sz_loglike_diff'] <= 1e-8, a
    post.close()


def test_two_contexts_that_share_a_kernel_instance_keep_their_own_sizes():
    """Sides 540, 512 and 500 (padded 576, 540, 540) share kernel instances (9 rows per lane): the LDS allowance of an instance must not be the
    last context's -- a context built later with a shorter transform would leave the longer one unable to launch.  All alive, evaluated in turn."""
    from joxsz_amd import datasets
    from joxsz_amd.posterior import JoxszPosterior
    posts = []
    for S, N in ((540, 560), (512, 500), (500, 520)):
        pb = datasets.synthetic_problem(S=S, N=N, seed=3)
        th = np.ascontiguousarray(datasets.walker_ball(pb, 3, spread=0.02, seed=4))
        ref = _run(pb, th, MODES['rocfft'])
        posts.append((JoxszPosterior(pb, device=0, conv='rocfft'), th, ref))
    for rep in range(2):
        for post, th, ref in posts:
            assert post.ctx.fft_info()['columns'] == 'custom'
            got = post.log_prob(th)
            fin = np.isfinite(ref[2])
            np.testing.assert_allclose(got[fin], ref[2][fin], rtol=1e-11)
    for post, _, _ in posts:
        post.close()
