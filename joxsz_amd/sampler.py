"""Minimal affine-invariant ensemble sampler (Goodman & Weare stretch move) with emcee's
red/blue half-ensemble structure, so that the walkers of each half step are evaluated in ONE
batched log-posterior call (the caller of the hot path).

emcee is not installed on the target image; this mirrors what ``mcmc.sample`` does with the
reference's callable (joxsz_funcs.py:593, 600, 622) closely enough to run BASELINE configs[0]
("30 walkers, 10 steps") end to end, and keeps emcee's chain layout (nsteps, nwalkers, ndim).
The log-probability function is any batched callable ``theta[W, ndim] -> logp[W]``:
``JoxszPosterior.log_prob`` on a GPU, ``ShardedLogProb`` across GPUs, the oracle on a CPU.
"""
import numpy as np


def initial_ball(log_prob, theta0, nwalkers, spread=0.1, rng=None, max_tries=100):
    """``_generateInitPars`` (joxsz_funcs.py:548-570): theta0*(1+spread*N(0,1)), keeping only
    positions with a finite log-posterior -- evaluated in batches instead of one by one."""
    rng = np.random.default_rng() if rng is None else rng
    theta0 = np.asarray(theta0, dtype=np.float64)
    assert np.all(np.isfinite(theta0))
    p0 = np.empty((0, theta0.size))
    for _ in range(max_tries):
        cand = theta0 * (1 + rng.normal(0., spread, size=(2 * nwalkers, theta0.size)))
        lp = np.asarray(log_prob(cand))
        p0 = np.vstack((p0, cand[np.isfinite(lp)]))
        if len(p0) >= nwalkers:
            return p0[:nwalkers]
    raise RuntimeError('could not find %d walkers with a finite log-posterior' % nwalkers)


class StretchMoveSampler:
    def __init__(self, nwalkers, ndim, log_prob, a=2.0, seed=None):
        if nwalkers < 2 * ndim or nwalkers % 2:
            raise ValueError('need an even number of walkers, at least 2*ndim')
        self.nwalkers, self.ndim, self.log_prob, self.a = nwalkers, ndim, log_prob, float(a)
        self.rng = np.random.default_rng(seed)
        self.naccepted = np.zeros(nwalkers)
        self.iteration = 0

    @property
    def acceptance_fraction(self):
        return self.naccepted / max(self.iteration, 1)

    def run(self, p0, nsteps, thin=1):
        """Returns (chain[nsteps//thin, nwalkers, ndim], log_prob[nsteps//thin, nwalkers])."""
        x = np.array(p0, dtype=np.float64)
        assert x.shape == (self.nwalkers, self.ndim)
        lp = np.asarray(self.log_prob(x), dtype=np.float64)
        if not np.all(np.isfinite(lp)):
            raise ValueError('initial positions must have finite log-probability')
        chain, lps = [], []
        half = self.nwalkers // 2
        idx = np.arange(self.nwalkers)
        for it in range(nsteps):
            for split in (0, 1):
                S1 = idx[split * half:(split + 1) * half]          # walkers being moved
                S2 = idx[(1 - split) * half:(2 - split) * half]    # complementary ensemble
                zz = ((self.a - 1.) * self.rng.random(half) + 1.) ** 2 / self.a
                partner = x[S2[self.rng.integers(0, half, size=half)]]
                q = partner - (partner - x[S1]) * zz[:, None]
                lq = np.asarray(self.log_prob(q), dtype=np.float64)   # ONE batched call per half step
                lnpdiff = (self.ndim - 1.) * np.log(zz) + lq - lp[S1]
                accept = np.log(self.rng.random(half)) < lnpdiff
                accept &= np.isfinite(lq)
                x[S1[accept]] = q[accept]
                lp[S1[accept]] = lq[accept]
                self.naccepted[S1[accept]] += 1
            self.iteration += 1
            if (it + 1) % thin == 0:
                chain.append(x.copy())
                lps.append(lp.copy())
        return np.array(chain), np.array(lps)
