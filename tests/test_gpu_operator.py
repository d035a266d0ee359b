"""The collapsed route (``jx_set_route(JX_ROUTE_OPERATOR)``): the SZ side as one constant matrix applied to the pressure
profile.  Same bar as the map route: the reference's golden log-posteriors and the oracle on seeded walkers, through the
C-ABI; plus what ties the two routes together (G pp equals the map route's row, identical rejections, ragged batches).
Needs an MI355X: run with -m gpu."""
import numpy as np
import pytest

from oracle import joxsz_oracle as orc

pytestmark = pytest.mark.gpu

RTOL = 1e-6          # BASELINE.json north_star: <= 1e-6 relative on the log-posterior


def _post(pb, **kw):
    from joxsz_amd.posterior import JoxszPosterior
    return JoxszPosterior(pb, device=0, **kw)


def test_golden_logp_operator_route(golden):
    pb, ref = golden
    post = _post(pb, route='operator')
    assert post.ctx.route == 'operator'
    got = post.log_prob(ref['thetas'])
    post.close()
    want = ref['ref_logp']
    fin = np.isfinite(want)
    assert np.array_equal(np.isfinite(got), fin) and np.all(got[~fin] == -np.inf)
    np.testing.assert_allclose(got[fin], want[fin], rtol=RTOL)
    assert np.max(np.abs(got[fin] - want[fin]) / np.abs(want[fin])) < 1e-9


@pytest.mark.parametrize('S,N,kw', [(31, 40, {}), (64, 80, {}), (48, 60, dict(sz_only=True)), (48, 60, dict(ne_mode='double')),
                                    (171, 313, {}), (256, 300, {}), (513, 500, {})])
def test_random_walkers_vs_oracle_operator_route(S, N, kw):
    """Odd (reference-shaped, rocFFT underneath) and even (hand-written passes underneath) sides."""
    from joxsz_amd import datasets
    pb = datasets.synthetic_problem(S=S, N=N, seed=S, **kw)
    p0 = orc.pars_dict(pb, datasets.fiducial_theta(pb))
    datasets.fill_data(pb, orc.sz_stages(pb, p0)['bright'], orc.calc_profiles(pb, p0), seed=S)
    W = 22 if S <= 64 else 6
    th = datasets.walker_ball(pb, W, spread=0.05, seed=S)
    th[1, 1] = 9.0                                    # one rejected walker in the batch
    post = _post(pb, route='operator')
    got = post.log_prob(th)
    post.close()
    want = orc.log_posterior_batch(pb, th)
    fin = np.isfinite(want)
    assert fin.sum() >= W // 2 and np.array_equal(np.isfinite(got), fin)
    np.testing.assert_allclose(got[fin], want[fin], rtol=RTOL)
    assert np.max(np.abs(got[fin] - want[fin]) / np.abs(want[fin])) < 1e-8


@pytest.mark.parametrize('S,N', [(31, 40), (64, 80)])
def test_operator_matrix_against_the_oracle_operator(S, N):
    """The matrix the library builds with its kernels against the one tabulated from the oracle's own chain
    (``orc.sz_operator``: numpy/scipy steps of funcs:457-472 on the unit profiles), entry by entry."""
    from joxsz_amd import datasets
    pb = datasets.synthetic_problem(S=S, N=N, seed=S)
    post = _post(pb, route='operator')
    G = post.ctx.operator()
    post.close()
    want = orc.sz_operator(pb)
    np.testing.assert_allclose(G, want, rtol=0, atol=1e-9 * np.abs(want).max())


def test_operator_is_the_map_route_applied_to_unit_profiles():
    """G pp against the row the map route extracts for the same walkers; both routes on the same batch."""
    from joxsz_amd import datasets
    pb = datasets.synthetic_problem(S=128, N=150, seed=13)
    post = _post(pb)
    th = datasets.walker_ball(pb, 37, spread=0.05, seed=2)           # ragged: not a multiple of the walkers per block
    th[5, 1] = 9.0
    lp_map = post.log_prob(th)
    pp, row = post.stage(th, 'pp'), post.stage(th, 'map_row')
    with pytest.raises(Exception):
        post.ctx.operator()                                          # not built before the route is first selected
    post.ctx.set_route('operator')
    G = post.ctx.operator()
    assert G.shape == (pb.N, pb.nrow) and np.all(np.isfinite(G))
    np.testing.assert_allclose(pp @ G, row, rtol=0, atol=1e-11 * np.abs(row).max())
    lp_op = post.log_prob(th)
    fin = np.isfinite(lp_map)
    assert not fin[5] and np.array_equal(np.isfinite(lp_op), fin)
    np.testing.assert_allclose(lp_op[fin], lp_map[fin], rtol=1e-10)
    # the stage taps keep running the map route, whatever the route of the log-posterior
    np.testing.assert_array_equal(post.stage(th, 'map_row'), row)
    # and back
    post.ctx.set_route('map')
    np.testing.assert_array_equal(post.log_prob(th), lp_map)
    post.close()


def test_operator_route_chunking_and_sampler():
    from joxsz_amd import datasets
    from joxsz_amd.sampler import DeviceStretchMove
    pb = datasets.synthetic_problem(S=64, N=80, seed=3)
    p0f = orc.pars_dict(pb, datasets.fiducial_theta(pb))
    datasets.fill_data(pb, orc.sz_stages(pb, p0f)['bright'], orc.calc_profiles(pb, p0f), seed=3)
    th = datasets.walker_ball(pb, 75, spread=0.03, seed=4)
    one = _post(pb, route='operator')
    want = one.log_prob(th)
    one.close()
    small = _post(pb, max_batch=16, route='operator')                # 75 walkers in chunks of 16: 4 full, one of 11
    assert small.ctx.chunk == 16
    np.testing.assert_array_equal(small.log_prob(th), want)
    for n in (1, 2, 3, 5):
        np.testing.assert_array_equal(small.log_prob(th[:n]), want[:n])
    fin = th[np.isfinite(want)][:32]
    s = DeviceStretchMove(small, seed=11)
    chain, lp, nacc = s.run(fin, 6)
    rc, rl, rn = s.replay(fin, 6)
    np.testing.assert_array_equal(nacc, rn)
    np.testing.assert_allclose(chain, rc, rtol=1e-13, atol=0)
    np.testing.assert_allclose(lp, rl, rtol=1e-10)          # (a last-bit difference in a position, times the slope of the posterior)
    small.close()


def test_operator_route_full_size():
    """BASELINE configs[2] shape: both routes on the same 256 walkers."""
    from joxsz_amd import datasets
    pb = datasets.synthetic_problem(S=512, N=500, seed=0)
    post = _post(pb)
    t0 = np.repeat(datasets.fiducial_theta(pb)[None, :], 2, axis=0)
    datasets.fill_data(pb, post.stage(t0, 'bright')[0], post.stage(t0, 'xprofs')[0], seed=0)
    post.close()
    post = _post(pb)
    th = datasets.walker_ball(pb, 256, spread=0.02, seed=1)
    lp_map = post.log_prob(th)
    post.ctx.set_route('operator')
    lp_op = post.log_prob(th)
    post.close()
    fin = np.isfinite(lp_map)
    assert fin.sum() > 200 and np.array_equal(np.isfinite(lp_op), fin)
    np.testing.assert_allclose(lp_op[fin], lp_map[fin], rtol=1e-9)


@pytest.mark.parametrize('S,N', [(512, 500), (171, 313), (256, 300)])
def test_operator_route_large_launch_equals_small_launch(S, N, monkeypatch):
    """Launches of 4096 walkers and more take G pp from the matrix-core kernel (32 walkers per block), smaller ones from
    the 4/8/16-walkers-per-block kernel.  Within each of the two a walker's log-posterior is the same bit pattern
    whichever launch it was part of; between them the sums run in different orders and agree to rounding."""
    from joxsz_amd import datasets
    pb = datasets.synthetic_problem(S=S, N=N, seed=0)
    th = datasets.walker_ball(pb, 9000, spread=0.03, seed=8)       # ragged last blocks of 32 and of 16
    th[7, 1] = 9.0
    post = _post(pb, route='operator')
    big = post.log_prob(th)                                                                   # matrix cores
    mid = np.concatenate([post.log_prob(th[k:k + 4500]) for k in range(0, 9000, 4500)])      # matrix cores, other blocks
    small = np.concatenate([post.log_prob(th[k:k + 500]) for k in range(0, 9000, 500)])      # 4 walkers per block
    tiny = np.concatenate([post.log_prob(th[k:k + 3000]) for k in range(0, 9000, 3000)])     # 8 per block
    post.close()
    fin = np.isfinite(big)
    assert fin.sum() > 8000 and big[7] == -np.inf and np.array_equal(np.isfinite(small), fin)
    np.testing.assert_array_equal(big, mid)
    np.testing.assert_array_equal(small, tiny)
    np.testing.assert_allclose(big[fin], small[fin], rtol=1e-12)
    monkeypatch.setenv('JOXSZ_OP_NARROW', '1')                      # the small-launch kernel on the large launch (16 per block)
    post = _post(pb, route='operator')
    np.testing.assert_array_equal(post.log_prob(th), small)
    post.close()
