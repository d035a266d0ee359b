"""The built-in stretch-move sampler: BASELINE configs[0] (bundled CL J1226.9+3332 data shape,
30 walkers, 10 steps) on the CPU oracle, and identical chains from the GPU callable."""
import numpy as np
import pytest

from joxsz_amd.sampler import StretchMoveSampler, initial_ball
from oracle import joxsz_oracle as orc


def test_samples_a_gaussian():
    rng_dim = 3
    logp = lambda t: -0.5 * np.sum((np.atleast_2d(t) / np.array([1., 2., 0.5])) ** 2, axis=1)
    s = StretchMoveSampler(24, rng_dim, logp, seed=1)
    p0 = np.random.default_rng(0).normal(size=(24, rng_dim))
    chain, lp = s.run(p0, 600)
    flat = chain[200:].reshape(-1, rng_dim)
    np.testing.assert_allclose(flat.std(axis=0), [1., 2., 0.5], rtol=0.15)
    assert 0.2 < s.acceptance_fraction.mean() < 0.9
    assert chain.shape == (600, 24, 3) and lp.shape == (600, 24)


def test_config0_bundled_cpu(golden_bundled):
    """30 walkers, 10 steps on the bundled-shape problem with the oracle as log-posterior."""
    pb, ref = golden_bundled
    calls = []

    def logp(t):
        calls.append(len(t))
        return orc.log_posterior_batch(pb, t)

    rng = np.random.default_rng(5)
    p0 = initial_ball(logp, ref['thetas'][0], 30, spread=0.01, rng=rng)
    s = StretchMoveSampler(30, pb.ndim, logp, seed=6)
    chain, lp = s.run(p0, 10)
    assert chain.shape == (10, 30, pb.ndim) and np.all(np.isfinite(lp))
    assert set(calls[-20:]) == {15}                       # two batched half-ensemble calls per step
    assert lp[-1].max() >= lp[0].min()
    # a stored position really has the stored log-posterior
    np.testing.assert_allclose(orc.get_likelihood(pb, chain[-1, 3]), lp[-1, 3], rtol=1e-12)


@pytest.mark.gpu
def test_config0_gpu_chain_equals_cpu_chain(golden_bundled):
    from joxsz_amd.posterior import JoxszPosterior
    pb, ref = golden_bundled
    post = JoxszPosterior(pb, device=0)
    out = {}
    for name, f in (('cpu', lambda t: orc.log_posterior_batch(pb, t)), ('gpu', post.log_prob)):
        rng = np.random.default_rng(5)
        p0 = initial_ball(f, ref['thetas'][0], 30, spread=0.01, rng=rng)
        s = StretchMoveSampler(30, pb.ndim, f, seed=6)
        out[name] = s.run(p0, 10)
    post.close()
    np.testing.assert_allclose(out['gpu'][0], out['cpu'][0], rtol=1e-12)      # same accept decisions, same chain
    np.testing.assert_allclose(out['gpu'][1], out['cpu'][1], rtol=1e-6)


def test_philox_known_answers():
    """The counter-based generator of the device sampler against the Random123 known-answer vectors for Philox4x32-10."""
    from joxsz_amd.sampler import philox4x32
    assert [int(v) for v in philox4x32(0, 0, 0, 0, 0)] == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    m = 0xffffffff
    assert [int(v) for v in philox4x32(m, m, m, m, 0xffffffffffffffff)] == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    r = philox4x32(np.arange(5), 3, 1, 0, 42)
    assert all(a.shape == (5,) for a in r) and len({int(v) for v in r[0]}) == 5


def test_device_sampler_replay_on_a_gaussian():
    """The host replay of the device sampler's algorithm samples a Gaussian correctly (moments), with any log_prob."""
    from joxsz_amd.sampler import DeviceStretchMove
    ndim, W = 3, 40
    sig = np.array([1.0, 0.5, 2.0])
    sm = DeviceStretchMove(None, a=2.0, seed=7)
    rng = np.random.default_rng(0)
    chain, lps, nacc = sm.replay(rng.normal(size=(W, ndim)), 600, log_prob=lambda x: -0.5 * np.sum((x / sig) ** 2, axis=1))
    flat = chain[200:].reshape(-1, ndim)
    assert np.all(np.abs(flat.mean(axis=0)) < 0.15 * sig)
    np.testing.assert_allclose(flat.std(axis=0), sig, rtol=0.12)
    assert 0.2 < nacc.mean() / 600 < 0.8
