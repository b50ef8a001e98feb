"""Hyper-parameter search on top of the device loglikelihood.

Host-side drivers only - they call ``Interpolator.loglikelihood`` (one kernel-matrix build + one Cholesky on the
GPU per evaluation) repeatedly.  Same entry points and argument meaning as the reference
(mgpi/interpolators.py:347-610): ``optimize_kernel`` (scipy.optimize.minimize, method 'TNC'), ``sample_kernel``
(emcee ensemble sampler).  The reference's latent bug at interpolators.py:365 (``_params`` used before
assignment when ``fixed`` is given) is not reproduced: fixed parameters work.
"""
import time

import numpy as np

DEFAULT_METHOD = 'TNC'           # interpolators.py:25
DEFAULT_TEMPERATURE = 1.0        # :31
DEFAULT_NUM_BURNIN = 100         # :32
DEFAULT_NUM_SAMPLES = 1000       # :33
DEFAULT_NUM_WALKERS = None       # :34  (None -> 2 * number of free parameters)


class LogPosterior(object):
    """params (the FREE kernel parameters, in kernel order) -> tempered loglikelihood + logprior.
    Non-positive or NaN parameters and prior-excluded points return -inf without touching the GPU
    (interpolators.py:369-387)."""

    def __init__(self, interp, source_x, source_f, logprior=None, fixed=None, temperature=DEFAULT_TEMPERATURE,
                 verbose=False, **kwargs):
        self.interp = interp
        self.source_x, self.source_f = source_x, source_f
        self.logprior = logprior if logprior is not None else (lambda params: 0.0)
        self.temperature = float(temperature)
        self.verbose = verbose
        self.kwargs = kwargs
        fixed = dict(fixed) if fixed else {}
        if fixed:
            interp.update(**fixed)
        self.free = [name for name in interp.kernel._params if name not in fixed]
        self._resident = None          # (X, f) on the device, uploaded on first use

    def __call__(self, params):
        """one parameter vector -> float; a (B, num_dim) array -> (B,) array in ONE device call (emcee's vectorize=True:
        the 2 * num_dim walkers of interpolators.py:513-514 are evaluated together, x and f resident on the GPU)"""
        params = np.asarray(params, dtype=float)
        if params.ndim == 2:
            return self.many(params)
        return float(self.many(params[None, :])[0])

    def many(self, params):
        params = np.asarray(params, dtype=float)
        out = np.full(len(params), -np.inf)
        logp = np.zeros(len(params))
        todo = []
        for i, row in enumerate(params):
            if np.any(row <= 0) or np.any(row != row):             # interpolators.py:371-372
                continue
            logp[i] = self.logprior(row)
            if logp[i] == -np.inf:
                continue
            todo.append(i)
        if not todo:
            return out
        if self._resident is None:
            self._resident = self.interp.resident(self.source_x, self.source_f) or ()
        # full kernel parameter vectors: the fixed ones at their values, the free ones from the rows
        names = list(self.interp.kernel._params)
        base = np.array(self.interp.kernel.params, dtype=float)
        cols = [names.index(name) for name in self.free]
        rows = np.tile(base, (len(todo), 1))
        rows[:, cols] = params[todo]
        logl, pivot = self.interp.loglikelihoods(self.source_x, self.source_f, rows, resident=self._resident or None, **self.kwargs)
        for j, i in enumerate(todo):
            if pivot[j] == 0:          # not positive definite at these parameters: an excluded point
                out[i] = logl[j] / self.temperature + logp[i]
                if self.verbose:
                    print('>>> %s\n  logl=%.6e\n  logp=%.6e' % (rows[j], logl[j] / self.temperature, logp[i]))
        if len(todo) == 1 and len(params) == 1:
            self.interp.update(**dict(zip(self.free, params[0])))      # a single evaluation leaves the kernel there, as upstream
        return out


def initial_points(interp, logprior=None, fixed=None, size=1, verbose=False, rng=None):
    """``size`` starting points scattered around the current kernel parameters: each free parameter times
    (1 + N(0,1)), redrawn until positive and allowed by the prior (interpolators.py:391-435)"""
    fixed = fixed or {}
    centre = np.array([val for name, val in zip(interp.kernel._params, interp.kernel.params) if name not in fixed])
    normal = rng.normal if rng is not None else np.random.normal
    if verbose:
        print('initializing %d samples with num_dim = %d (%d params - %d fixed)' %
              (size, len(centre), len(interp.kernel.params), len(fixed)))
        t0 = time.time()
    points, trials = [], 0
    while len(points) < size:
        trials += 1
        cand = centre * (1 + normal(size=len(centre)))
        if np.any(cand <= 0):
            continue
        if logprior is not None and logprior(cand) == -np.inf:
            continue
        points.append(cand)
    if verbose:
        print('    time : %.6f sec (%d/%d trials accepted)' % (time.time() - t0, size, trials))
    return np.array(points)


class HyperParameterSearch(object):
    """mixin for Interpolator: optimize_kernel / sample_kernel"""

    def _construct_logprob(self, source_x, source_f, logprior=None, fixed=None, temperature=DEFAULT_TEMPERATURE,
                           verbose=False, **kwargs):
        return LogPosterior(self, source_x, source_f, logprior=logprior, fixed=fixed, temperature=temperature,
                            verbose=verbose, **kwargs)

    def _construct_initial_params(self, logprior=None, fixed=None, size=1, verbose=False):
        return initial_points(self, logprior=logprior, fixed=fixed, size=size, verbose=verbose)

    def optimize_kernel(self, source_x, source_f, method=DEFAULT_METHOD, logprior=None, fixed=None,
                        temperature=DEFAULT_TEMPERATURE, verbose=False, Verbose=False):
        """maximise loglikelihood (+ logprior) over the free kernel parameters with scipy.optimize.minimize;
        the kernel is left at the optimum and its params_array is returned (interpolators.py:439-487)"""
        try:
            from scipy.optimize import minimize
        except ImportError:
            raise ImportError('could not import scipy.optimize.minimize')
        logprob = self._construct_logprob(source_x, source_f, logprior=logprior, fixed=fixed,
                                          temperature=temperature, verbose=Verbose)
        start = self._construct_initial_params(logprior=logprior, fixed=fixed, size=1, verbose=Verbose)[0]
        if verbose:
            print('extremizing loglikelihood')
            t0 = time.time()
        result = minimize(lambda params: -logprob(params), start, method=method)
        if verbose:
            print('    time : %.6f sec' % (time.time() - t0))
        self.update(**dict(zip(logprob.free, result.x)))
        return self.kernel.params_array

    def sample_kernel(self, source_x, source_f, logprior=None, fixed=None, temperature=DEFAULT_TEMPERATURE,
                      num_burnin=DEFAULT_NUM_BURNIN, num_samples=DEFAULT_NUM_SAMPLES, num_walkers=DEFAULT_NUM_WALKERS,
                      verbose=False, Verbose=False):
        """MCMC over the free kernel parameters with emcee's EnsembleSampler (interpolators.py:491-610);
        returns (samples (num_samples, num_walkers, num_dim), logprob (num_samples, num_walkers), sampler)"""
        try:
            import emcee
        except ImportError:
            raise ImportError('could not import emcee')
        logprob = self._construct_logprob(source_x, source_f, logprior=logprior, fixed=fixed,
                                          temperature=temperature, verbose=Verbose)
        num_dim = len(logprob.free)
        if num_walkers is None:
            num_walkers = 2 * num_dim
        state = self._construct_initial_params(logprior=logprior, fixed=fixed, size=num_walkers, verbose=Verbose)
        # vectorize=True: emcee hands over all walkers of a half-step at once -> one batched device call
        sampler = emcee.EnsembleSampler(num_walkers, num_dim, logprob, vectorize=True)
        if verbose:
            print('running sampler for %d steps of burn-in' % num_burnin)
        state = sampler.run_mcmc(state, num_burnin, progress=Verbose)
        sampler.reset()
        if verbose:
            print('running sampler for %d steps' % num_samples)
        sampler.run_mcmc(state, num_samples, progress=Verbose)
        # (num_samples, num_walkers, num_dim) and (num_samples, num_walkers), unflattened like interpolators.py:606-609
        return sampler.get_chain(), sampler.get_log_prob(), sampler
