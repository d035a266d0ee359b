"""Host-side mirror of the reference's log-posterior callable.

``JoxszPosterior`` plays the role of the ``mb.Fit`` object after
joxsz_main.py:186-188 patched ``getLikelihood`` / ``get_sz_like`` onto it: same
method names, argument meaning and return conventions, evaluated on the GPU
through ``joxsz_amd.hip_backend``.  It can be handed to emcee unchanged::

    sampler = emcee.EnsembleSampler(nwalkers, post.ndim, post.getLikelihood)          # one call per walker
    sampler = emcee.EnsembleSampler(nwalkers, post.ndim, post, vectorize=True)        # one launch per half-step
    sampler = emcee.EnsembleSampler(nwalkers, post.ndim, post.getLikelihood, pool=post.pool())

``multiprocessing.Pool`` must not be used with it (no fork after HIP init); the
walker axis the reference spreads over processes (joxsz_main.py:203-206) is the
batch axis of one kernel launch here.
"""
import numpy as np

from .hip_backend import HipContext


class JoxszPosterior:
    def __init__(self, problem, device=0, max_batch=0, fft_pad=0, map_split=0, conv='auto', route=None, dtype='f64', options=None):
        self.problem = problem
        self.ctx = HipContext(problem, device=device, max_batch=max_batch, fft_pad=fft_pad, map_split=map_split,
                              conv=conv, route=route, dtype=dtype, options=options)
        self.thawed = list(problem.thawed)                 # joxsz_main.py:179
        self.ndim = problem.ndim
        self.exclude_unphy_mass = bool(problem.exclude_unphy_mass)

    # ---- mbproj2.Fit surface used by the reference's drivers ----
    def thawedParVals(self):
        """mbproj2 ``Fit.thawedParVals`` (joxsz_funcs.py:555, 585)."""
        return self.problem.thawed_vals()

    def updateThawed(self, vals):
        """mbproj2 ``Fit.updateThawed`` (joxsz_funcs.py:516): the current
        parameters become ``vals``, on the host table and on the device."""
        vals = np.asarray(vals, dtype=np.float64)
        if vals.shape != (self.ndim,):
            raise ValueError('expected %d thawed values' % self.ndim)
        self.problem.par_vals[self.problem.thawed_idx] = vals
        self.ctx.set_par_vals(self.problem.par_vals)

    @property
    def pars(self):
        return dict(zip(self.problem.par_names, self.problem.par_vals))

    # ---- the hot path ----
    def getLikelihood(self, vals=None):
        """joxsz_funcs.py:507-546: joint X-ray + SZ log-posterior of one
        parameter vector, a Python float, ``-inf`` on rejection.  ``vals=None``
        evaluates the current parameters; otherwise they become current."""
        if vals is not None:
            self.updateThawed(vals)
        return float(self.ctx.eval(self.thawedParVals())[0])

    def log_prob(self, theta):
        """Batched form: theta [W, ndim] -> [W] float64, one launch sequence
        for all walkers.  Does not touch the current parameters."""
        return self.ctx.eval(theta)

    def __call__(self, theta):
        theta = np.asarray(theta, dtype=np.float64)
        if theta.ndim == 1:
            return float(self.ctx.eval(theta)[0])
        return self.ctx.eval(theta)

    def get_sz_like(self, output='ll'):
        """joxsz_funcs.py:439-493 for the current parameters."""
        th = self.thawedParVals()
        if output == 'pp':
            return self.ctx.eval_stage(th, 'pp')[0]
        if output == 'bright':
            return self.ctx.eval_stage(th, 'bright')[0]
        if output == 'chisq':
            return float(self.ctx.eval_stage(th, 'chisq')[0])
        if output == 'll':
            return float(self.ctx.eval_stage(th, 'parts')[0, 1])          # -chisq/2, minus the integrated-Compton term if enabled
        if output == 'integ' and getattr(self.problem, 'calc_integ', False):
            return float(self.ctx.eval_stage(th, 'integ')[0])
        raise RuntimeError('Unrecognised output name (must be "ll", "chisq", "pp", "bright" or "integ")')

    def calcProfiles(self):
        """mbproj2 ``Fit.calcProfiles`` (joxsz_funcs.py:527): predicted counts
        per band and annulus for the current parameters."""
        return list(self.ctx.eval_stage(self.thawedParVals(), 'xprofs')[0])

    def stage(self, theta, name):
        """Batched parity tap (see ``hip_backend.STAGES``)."""
        return self.ctx.eval_stage(theta, name)

    def sample(self, p0, nsteps, a=2.0, seed=0):
        """Device-resident stretch-move run from the start positions ``p0[W, ndim]`` (``jx_sample``; what ``mcmc.sample``
        does with the reference's callable, joxsz_funcs.py:593-622, without a host round trip per step).
        Returns (chain[nsteps, W, ndim], log_prob[nsteps, W], naccepted[W]) in emcee's chain layout."""
        return self.ctx.sample(p0, nsteps, a, seed)

    def pool(self):
        return WalkerPool(self)

    def close(self):
        self.ctx.close()


class WalkerPool:
    """Object with the ``map(func, iterable)`` method emcee expects from
    ``pool=``: evaluates all positions of a half-step in one batch."""

    def __init__(self, posterior):
        self.posterior = posterior

    def map(self, func, iterable):
        thetas = np.array(list(iterable), dtype=np.float64)
        if thetas.size == 0:
            return []
        return list(self.posterior.log_prob(thetas))
