"""The callers either side of the log-posterior: the MCMC run that feeds it and the chain
post-processing that loops over it again, both batched over the walker / sample axis.

* ``mcmc_run``            -- joxsz_funcs.py:572-635 (preliminary fit, burn-in, sampling) on the device-resident stretch move
* ``save_chain/load_chain`` -- joxsz_main.py:196-199 + ``add_backend_attrs`` joxsz_funcs.py:637-650 (h5py is not on the
  target image: one ``.npz`` with the same datasets and attributes)
* ``cube_chain/flat_chain`` -- the two layouts joxsz_main.py:213-214 hands to the plotting code
* ``equal_tailed``, ``chain_subset``, ``best_fit_prof`` -- joxsz_plots.py:93-132
"""
import numpy as np

from .sampler import initial_ball, StretchMoveSampler


# ---------------------------------------------------------------------------------------
# chain layouts and storage
# ---------------------------------------------------------------------------------------
def cube_chain(chain):
    """emcee's ``mcmc.chain`` (joxsz_main.py:213): [nsteps, W, ndim] -> [W, nsteps, ndim]."""
    return np.ascontiguousarray(np.swapaxes(np.asarray(chain), 0, 1))


def flat_chain(cube):
    """joxsz_main.py:214: ((W x niter) x ndim), walker index fastest."""
    cube = np.asarray(cube)
    return cube.reshape(-1, cube.shape[2], order='F')


def save_chain(path, chain, log_prob, param_names, burn, thin, accepted=None):
    """One file per run with what emcee's HDF backend plus ``add_backend_attrs`` leave behind: datasets ``chain``
    [nsteps, W, ndim], ``log_prob`` [nsteps, W], ``accepted`` [W]; attributes ``param_names``, ``burn``, ``thin``."""
    chain, log_prob = np.asarray(chain, np.float64), np.asarray(log_prob, np.float64)
    if chain.ndim != 3 or log_prob.shape != chain.shape[:2] or len(param_names) != chain.shape[2]:
        raise ValueError('chain [nsteps, W, ndim], log_prob [nsteps, W] and param_names [ndim] do not agree')
    acc = np.zeros(chain.shape[1]) if accepted is None else np.asarray(accepted, np.float64)
    with open(path, 'wb') as f:                                # (a file object: numpy would append ".npz" to a bare name)
        np.savez(f, chain=chain, log_prob=log_prob, accepted=acc,
                 param_names=np.array([k.encode('utf-8') for k in param_names]), burn=np.int64(burn), thin=np.int64(thin))


def load_chain(path):
    """Returns dict(chain, log_prob, accepted, param_names (str list), burn, thin)."""
    with np.load(path) as z:
        return dict(chain=z['chain'], log_prob=z['log_prob'], accepted=z['accepted'],
                    param_names=[k.decode('utf-8') for k in z['param_names']], burn=int(z['burn']), thin=int(z['thin']))


# ---------------------------------------------------------------------------------------
# the run
# ---------------------------------------------------------------------------------------
class _Runner:
    """``sample(p0, n) -> (chain[n, W, ndim], logp[n, W], naccepted[W])`` on the device (``JoxszPosterior.sample``) or,
    for any other batched callable, with the host stretch move."""

    def __init__(self, post, nwalkers, ndim, a, seed, device):
        self.post, self.a, self.seed, self.calls = post, a, int(seed), 0
        self.device = device
        self.host = None if device else StretchMoveSampler(nwalkers, ndim, getattr(post, 'log_prob', post), a=a, seed=seed)

    def sample(self, p0, n):
        self.calls += 1
        if self.device:                                         # a fresh counter stream per call
            return self.post.sample(p0, n, a=self.a, seed=self.seed + 7919 * self.calls)
        before = self.host.naccepted.copy()
        chain, lp = self.host.run(p0, n)
        return chain, lp, self.host.naccepted - before


class JoxszAuditWarning(UserWarning):
    """The live walkers of a chain, sent through the rocFFT sequence as well (``jx_audit``), differ from the route in use by more than the bound."""


def audit_walkers(post, walkers, bound=1e-6, nmax=64, log=None):
    """Run-time assurance where the chain lives: up to ``nmax`` of the given walkers through the context's own route
    and through the rocFFT sequence held inside it (``HipContext.audit``).  Warns (once per call) when the SZ log-likelihood differs by
    more than ``bound`` ABSOLUTE on any of them; returns the audit's figures, or None for a posterior without a device context."""
    ctx = getattr(post, 'ctx', None)
    if ctx is None or not hasattr(ctx, 'audit'):
        return None
    w = np.asarray(walkers, dtype=np.float64)
    if len(w) > nmax:
        w = w[np.linspace(0, len(w) - 1, nmax).astype(int)]
    res = ctx.audit(w)
    if log is not None:
        log('audit of %d live walkers: |delta SZ log-likelihood| <= %.2e, row <= %.2e of its maximum' % (res['walkers_compared'], res['max_abs_sz_loglike_diff'], res['max_rel_row_diff']))
    if res['max_abs_sz_loglike_diff'] > bound:
        import warnings
        warnings.warn('the SZ log-likelihood of walker %d differs by %.2e from the rocFFT sequence (bound %.0e): the route in use approximates more than '
                      'this chain can take (a contracted form or an fp32 variant? the default exact f64 context reads ~1e-11)'
                      % (res['worst_walker'], res['max_abs_sz_loglike_diff'], bound), JoxszAuditWarning, stacklevel=2)
    return res


def mcmc_run(post, nwalkers, nburn, nsteps, nthin=1, initspread=0.1, a=2.0, seed=0, prelim_iters=1000, max_prelim=20,
             device=None, theta0=None, log=None, audit_every=500, audit_bound=1e-6):
    """joxsz_funcs.py:572-635 with the whole ensemble advanced on the device.

    1. start ball around the current parameters (``_generateInitPars``, joxsz_funcs.py:548-570);
    2. blocks of ``prelim_iters`` steps, repeated while the best log-posterior of the block's last step is at least
       that of the previous one (joxsz_funcs.py:589-599; ``max_prelim`` bounds what the reference leaves open);
    3. ``nburn`` burn-in steps from there (joxsz_funcs.py:600-602), then ``nsteps`` steps kept every ``nthin``-th
       (joxsz_funcs.py:618-624).

    ``post`` is a ``JoxszPosterior`` (device sampler) or any batched callable theta[W, ndim] -> logp[W] (host
    sampler; ``theta0`` is then required).  With a device context the live walkers are AUDITED -- sent through the rocFFT sequence as well,
    ``audit_walkers`` -- behind the burn-in and then every ``audit_every`` kept steps (0: never; the sampling then runs in blocks of that
    many steps, the chain is the same walker-for-walker only for one block: each block draws from its own counter stream).  Returns
    dict(chain [nsteps//nthin, W, ndim], log_prob, accepted [W], acceptance_fraction, burn_best, prelim_blocks, audits)."""
    log = (lambda *_: None) if log is None else log
    device = hasattr(post, 'sample') if device is None else device
    log_prob = getattr(post, 'log_prob', post)
    theta0 = np.asarray(post.thawedParVals() if theta0 is None else theta0, dtype=np.float64)
    ndim = theta0.size
    rng = np.random.default_rng(seed)
    bestprob = float(np.asarray(log_prob(theta0[None, :]))[0])
    newlike = bestprob
    p0 = initial_ball(log_prob, theta0, nwalkers, spread=initspread, rng=rng)
    run = _Runner(post, nwalkers, ndim, a, seed, device)
    blocks = 0
    log('Preliminary fit (%d iterations) to improve likelihood' % prelim_iters)
    while newlike >= bestprob and blocks < max_prelim and prelim_iters > 0:
        bestprob = newlike
        chain, lp, _ = run.sample(p0, prelim_iters)
        newlike = float(lp[-1].max())
        p0 = chain[-1]
        blocks += 1
    if nburn > 0:
        log('Burn-in period')
        chain, lp, _ = run.sample(p0, nburn)
        p0 = chain[-1]
    audits = []
    do_audit = bool(audit_every) and device and hasattr(getattr(post, 'ctx', None), 'audit')
    if do_audit:
        audits.append(audit_walkers(post, p0, audit_bound, log=log))
    log('Starting sampling')
    if do_audit and nsteps > audit_every:
        parts, lps, acc = [], [], 0
        done = 0
        while done < nsteps:
            n = min(audit_every, nsteps - done)
            c_, l_, a_ = run.sample(p0, n)
            parts.append(c_); lps.append(l_); acc = acc + np.asarray(a_, np.float64)
            p0 = c_[-1]
            done += n
            audits.append(audit_walkers(post, p0, audit_bound, log=log))
        chain, lp = np.concatenate(parts), np.concatenate(lps)
    else:
        chain, lp, acc = run.sample(p0, nsteps)
        if do_audit:
            audits.append(audit_walkers(post, chain[-1], audit_bound, log=log))
    keep = slice(nthin - 1, None, nthin)
    out = dict(chain=chain[keep], log_prob=lp[keep], accepted=np.asarray(acc, np.float64),
               acceptance_fraction=float(np.mean(acc) / max(nsteps, 1)), burn_best=max(bestprob, newlike), prelim_blocks=blocks, audits=audits)
    log('Finished sampling')
    log('Acceptance fraction: %s' % out['acceptance_fraction'])
    return out


# ---------------------------------------------------------------------------------------
# posterior-predictive summaries
# ---------------------------------------------------------------------------------------
def equal_tailed(data, ci=95):
    """joxsz_plots.py:93-102: [lower, median, upper] of the equal-tailed interval along axis 0."""
    low, med, upp = map(np.atleast_1d, np.percentile(data, [50 - ci / 2, 50, 50 + ci / 2], axis=0))
    return np.array([low, med, upp])


def chain_subset(cube, num='all', seed=None):
    """The parameter vectors joxsz_plots.py:116-123 (and :261-268, :353-360, :463-470) visit: ``num`` draws without
    replacement from the (W x niter) samples of ``cube`` [W, niter, ndim], in the same order for the same seed."""
    cube = np.asarray(cube)
    nw, nit = cube.shape[:2]
    if num == 'all':
        num = nw * nit
    w, it = np.meshgrid(np.arange(nw), np.arange(nit))
    w, it = w.flatten(), it.flatten()
    np.random.seed(seed)
    rand = np.random.choice(w.size, num, replace=False)
    return cube[w[rand], it[rand], :]


def best_fit_prof(cube, post, num='all', seed=None, ci=95, batch=4096):
    """joxsz_plots.py:104-132: equal-tailed interval of the predicted X-ray count profiles [3, nband, nann] and of the
    SZ surface-brightness profile [3, nrow] over the chain -- the stage taps of the device path in batches instead of
    one ``calcProfiles`` + ``get_sz_like(output='bright')`` per sample."""
    thetas = chain_subset(cube, num, seed)
    px, ps = [], []
    for s in range(0, len(thetas), batch):
        px.append(post.stage(thetas[s:s + batch], 'xprofs'))
        ps.append(post.stage(thetas[s:s + batch], 'bright'))
    return equal_tailed(np.concatenate(px), ci), equal_tailed(np.concatenate(ps), ci)
