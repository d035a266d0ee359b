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


# ---------------------------------------------------------------------------------------
# Device-resident variant (jx_sample) and its host replay
# ---------------------------------------------------------------------------------------
_M0, _M1, _W0, _W1 = 0xD2511F53, 0xCD9E8D57, 0x9E3779B9, 0xBB67AE85
_MASK = 0xFFFFFFFF


def philox4x32(c0, c1, c2, c3, seed):
    """Philox4x32-10 on arrays of 32-bit counters, key = (seed & 0xffffffff, seed >> 32): the generator of jx_sample."""
    c0, c1, c2, c3 = (np.asarray(v, dtype=np.uint64) & _MASK for v in np.broadcast_arrays(c0, c1, c2, c3))
    k0, k1 = np.uint64(seed & _MASK), np.uint64((seed >> 32) & _MASK)
    for _ in range(10):
        p0, p1 = np.uint64(_M0) * c0, np.uint64(_M1) * c2
        c0, c1, c2, c3 = ((p1 >> np.uint64(32)) ^ c1 ^ k0) & _MASK, p1 & _MASK, ((p0 >> np.uint64(32)) ^ c3 ^ k1) & _MASK, p0 & _MASK
        k0, k1 = (k0 + np.uint64(_W0)) & np.uint64(_MASK), (k1 + np.uint64(_W1)) & np.uint64(_MASK)
    return c0, c1, c2, c3


def _u01(hi, lo):
    return ((((hi << np.uint64(32)) | lo) >> np.uint64(11)).astype(np.float64)) * (1.0 / 9007199254740992.0)


class DeviceStretchMove:
    """The stretch move with the whole loop on the device (``JoxszPosterior.sample`` -> ``jx_sample``): proposals,
    evaluation, accept/reject and the chain never leave the GPU.  ``replay`` runs the same algorithm on the host with
    the same counter-based random numbers and any batched ``log_prob``: with the device's own ``log_prob`` it
    reproduces the device chain exactly, which is how the test checks the kernels."""

    def __init__(self, posterior, a=2.0, seed=0):
        self.post, self.a, self.seed = posterior, float(a), int(seed)

    def run(self, p0, nsteps):
        return self.post.ctx.sample(p0, nsteps, self.a, self.seed)

    def replay(self, p0, nsteps, log_prob=None):
        log_prob = self.post.log_prob if log_prob is None else log_prob
        x = np.array(p0, dtype=np.float64)
        W, ndim = x.shape
        half = W // 2
        lp = np.asarray(log_prob(x), dtype=np.float64)
        nacc = np.zeros(W, np.int64)
        chain, lps = np.empty((nsteps, W, ndim)), np.empty((nsteps, W))
        i = np.arange(half)
        for it in range(nsteps):
            for hs in (0, 1):
                s1, s2 = hs * half, (1 - hs) * half
                r = philox4x32(i, 2 * it + hs, 0, 0, self.seed)
                u1, u2 = _u01(r[0], r[1]), _u01(r[2], r[3])
                t = (self.a - 1.0) * u1 + 1.0
                z = (t * t) / self.a
                j = np.minimum((u2 * half).astype(np.int64), half - 1)
                xp, xi = x[s2 + j], x[s1:s1 + half]
                q = xp - (xp - xi) * z[:, None]
                lq = np.asarray(log_prob(q), dtype=np.float64)
                u3 = _u01(*philox4x32(i, 2 * it + hs, 1, 0, self.seed)[:2])
                with np.errstate(invalid='ignore'):
                    lnpdiff = (ndim - 1) * np.log(z) + (lq - lp[s1:s1 + half])
                    acc = np.isfinite(lq) & (np.log(u3) < lnpdiff)
                x[s1:s1 + half][acc] = q[acc]
                lp[s1:s1 + half][acc] = lq[acc]
                nacc[s1:s1 + half][acc] += 1
            chain[it], lps[it] = x, lp
        return chain, lps, nacc
