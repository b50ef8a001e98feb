"""Dense Gaussian-process regression on one MI355X: the host-side mirror of the reference's
``Interpolator`` (mgpi/interpolators.py:47-343).

Drop-in surface (same names, argument order and meaning, same exceptions):

    Interpolator(kernel, nugget=None)                                   interpolators.py:52
    .update(*args, **kw) / .update_nugget(*args, **kw)                  :56-66
    .compress(source_x, source_f, verbose=False, Verbose=False) -> (N,)           :72
    .predict(target_x, source_x, compressed, verbose=False, Verbose=False) -> (M,) :109
    .condition(target_x, source_x, source_f, verbose=False, Verbose=False) -> (mean (M,), cov (M,M))  :136
    ._x2cov(x1, x2, kernel, verbose=False) -> (n1, n2)   [staticmethod]  :208
    .loglikelihood(source_x, source_f, verbose=False) -> float          :331
    .rvs(target_x, source_x, source_f, size=1) -> (size, M)             :306

Additive (not in the reference): ``condition_diag`` (mean + pointwise variance without the M x M
matrix - the only way to ask for a variance at 1e6 query points), ``fit`` / ``FittedGP`` (keeps x, L and
alpha resident in HBM between calls), ``sample`` (alias of rvs).

Where the reference calls ``np.linalg.inv`` (:92,:179,:343) this module factors K = L L^T on the GPU
(mgpi_potrf_dev) and solves with L; where it materialises n1*n2 row pairs (:247-294) the GPU builds
the matrix tile by tile (mgpi_cov_matrix_dev).  All O(N^2) and O(N^3) arithmetic is in
libmgpi_hip.so; there is no numpy fallback.
"""
import ctypes
import os
import time

import numpy as np

from . import _lib
from .hyper import HyperParameterSearch
from .kernel_objects import Kernel

__all__ = ['Interpolator', 'FittedGP']


def _points(a):
    """(P, D) float64 coordinates; an empty set raises IndexError like the reference (its pair builder reads row 0
    first, interpolators.py:247-249)"""
    a = _lib.as_f64(a, ndim=2)
    if len(a) == 0:
        raise IndexError('index 0 is out of bounds for axis 0 with size 0')
    return a


def _flatten(kernel, nugget=None):
    """ordered summands of kernel (+ nugget): what ``kernel+nugget`` means at interpolators.py:84,169,337"""
    terms = list(kernel.terms())
    if nugget is not None:
        terms.extend(nugget.terms())
    return terms


class _Stage(object):
    """prints the reference's stage banners and timings (interpolators.py:78-102 etc.)"""

    def __init__(self, ctx, enabled):
        self.ctx = ctx
        self.enabled = enabled
        self.t0 = None

    def begin(self, text):
        if self.enabled:
            print(text)
            self.ctx.sync()
            self.t0 = time.time()

    def end(self):
        if self.enabled:
            self.ctx.sync()
            print('    time : %.6f sec' % (time.time() - self.t0))


def riders_pay(n, m):
    """should m query points ride through the factorisation of an n-point GP (mgpi_posv_riders_dev)?  Measured on MI355X
    (profiles/r03_riders_ab.md): up to N ~ 16k the factorisation is bound by its panel chain and the riders' N^2 M flops hide
    beside it (C2, N = 8192, M = 4096: step 14.03 -> 13.07 ms); at N = 32768 the chip is busy with the trailing update anyway
    (261.0 -> 259.2 ms, and the dominant kernel shares the GPU with more look-ahead work), so the two-pass form stays there.
    MGPI_RIDERS=0 / 1 forces either form, MGPI_RIDERS_MAX_N moves the threshold."""
    forced = os.environ.get('MGPI_RIDERS')
    if m < 128 or m > n:
        return False
    if forced is not None:
        return forced != '0'
    return n <= int(os.environ.get('MGPI_RIDERS_MAX_N', '16384'))


class FittedGP(object):
    """x, the Cholesky factor of K(x,x; kernel+nugget) and (optionally) alpha, resident in HBM"""

    def __init__(self, ctx, source_x, kernel_terms, solve_terms, stage=None, source_f=None, riders_x=None):
        self.ctx = ctx
        self.lib = ctx.lib
        x = _points(source_x)
        self.n, self.d = x.shape
        self.kernel_terms = list(kernel_terms)     # K*, K** : kernel only
        self.solve_terms = list(solve_terms)       # K       : kernel + nugget
        self._kdesc = _lib.make_terms(self.kernel_terms)
        self._sdesc = _lib.make_terms(self.solve_terms)
        self.X = _lib.DeviceMatrix.from_host(ctx, x)
        self.alpha = None
        self.logdet = None
        self.riders = None             # (XS, KR) when query points rode through the factorisation (condition_diag)
        stage = stage or _Stage(ctx, False)
        if riders_x is not None:
            self._fit_with_riders(stage, source_f, riders_x)
            return
        self.L = _lib.DeviceMatrix(ctx, self.n, self.n, wide=True)

        stage.begin('constructing %d x %d source-source covariance matrix' % (self.n, self.n))
        # lower tiles only: the factorisation never reads the strict upper triangle
        _lib.check(self.lib.mgpi_cov_matrix_dev(ctx.handle, self.X.handle, None, self._sdesc[0], self._sdesc[1], 1,
                                                self.L.handle))
        stage.end()

        stage.begin('inverting source-source covariance matrix')
        logdet = ctypes.c_double(0.0)
        if source_f is None:
            _lib.check(self.lib.mgpi_potrf_dev(ctx.handle, self.L.handle, ctypes.byref(logdet)))
        else:
            # factor and solve in one call: the forward sweep of alpha = K^-1 f runs beside the factorisation
            f = _lib.as_f64(source_f).reshape(-1)
            assert len(f) == self.n, 'source_f has %d entries for %d source points' % (len(f), self.n)
            self.alpha = _lib.DeviceMatrix.from_host(ctx, f)
            _lib.check(self.lib.mgpi_posv_dev(ctx.handle, self.L.handle, self.alpha.handle, ctypes.byref(logdet)))
        self.logdet = logdet.value
        stage.end()

    def _fit_with_riders(self, stage, source_f, riders_x):
        """condition()'s three steps fused (mgpi_cov_riders_dev / mgpi_posv_riders_dev): K(x*, x) sits below K as extra rows and
        comes out of the factorisation as V^T = K* L^-T - the forward solve of the posterior variance done by the panels'
        substitutions and inside the trailing-update launches, not as a blocked triangular solve of its own afterwards"""
        ctx, lib = self.ctx, self.lib
        XS = self._targets(riders_x)
        pn = (self.n + 127) // 128 * 128
        KR = _lib.DeviceMatrix(ctx, pn + XS.rows, self.n, wide=True)
        stage.begin('constructing %d x %d source-source covariance matrix' % (self.n, self.n))
        _lib.check(lib.mgpi_cov_riders_dev(ctx.handle, self.X.handle, XS.handle, self._kdesc[0], self._kdesc[1], self._sdesc[0],
                                           self._sdesc[1], KR.handle))
        stage.end()
        stage.begin('inverting source-source covariance matrix')
        logdet = ctypes.c_double(0.0)
        bh = None
        if source_f is not None:
            f = _lib.as_f64(source_f).reshape(-1)
            assert len(f) == self.n, 'source_f has %d entries for %d source points' % (len(f), self.n)
            self.alpha = _lib.DeviceMatrix.from_host(ctx, f)
            bh = self.alpha.handle
        _lib.check(lib.mgpi_posv_riders_dev(ctx.handle, KR.handle, bh, ctypes.byref(logdet)))
        self.logdet = logdet.value
        stage.end()
        h = _lib._vp()
        _lib.check(lib.mgpi_mat_square_view(KR.handle, ctypes.byref(h)))
        self.L = _lib.DeviceMatrix(ctx, self.n, self.n, handle=h, parent=KR)      # the factor, for everything that takes L
        self.riders = (XS, KR)

    def riders_var_dev(self):
        """posterior variance at the points that rode through the factorisation"""
        XS, KR = self.riders
        out = _lib.DeviceMatrix(self.ctx, XS.rows, 1)
        _lib.check(self.lib.mgpi_riders_var_dev(self.ctx.handle, KR.handle, XS.handle, self._kdesc[0], self._kdesc[1], out.handle))
        return out

    # alpha = K^-1 f
    def solve(self, source_f):
        f = _lib.as_f64(source_f).reshape(-1)
        assert len(f) == self.n, 'source_f has %d entries for %d source points' % (len(f), self.n)
        self.alpha = _lib.DeviceMatrix.from_host(self.ctx, f)
        _lib.check(self.lib.mgpi_potrs_dev(self.ctx.handle, self.L.handle, self.alpha.handle))
        return self.alpha

    def set_alpha(self, compressed):
        a = _lib.as_f64(compressed).reshape(-1)
        assert len(a) == self.n, 'compressed has %d entries for %d source points' % (len(a), self.n)
        self.alpha = _lib.DeviceMatrix.from_host(self.ctx, a)

    def _targets(self, target_x):
        xs = _points(target_x)
        assert xs.shape[1] == self.d, 'conflicting number of dimensions for x1 and x2'   # interpolators.py:219
        return _lib.DeviceMatrix.from_host(self.ctx, xs)

    def mean_dev(self, XS):
        out = _lib.DeviceMatrix(self.ctx, XS.rows, 1)
        _lib.check(self.lib.mgpi_predict_mean_dev(self.ctx.handle, XS.handle, self.X.handle, self._kdesc[0],
                                                  self._kdesc[1], self.alpha.handle, out.handle))
        return out

    def var_dev(self, XS, m_slab=0):
        out = _lib.DeviceMatrix(self.ctx, XS.rows, 1)
        _lib.check(self.lib.mgpi_predict_var_dev(self.ctx.handle, XS.handle, self.X.handle, self._kdesc[0],
                                                 self._kdesc[1], self.L.handle, int(m_slab), out.handle))
        return out

    def cov_dev(self, XS):
        m = XS.rows
        out = _lib.DeviceMatrix(self.ctx, m, m, wide=True)
        _lib.check(self.lib.mgpi_condition_cov_dev(self.ctx.handle, XS.handle, self.X.handle, self._kdesc[0],
                                                   self._kdesc[1], self.L.handle, out.handle))
        return out

    def mean(self, target_x):
        return self.mean_dev(self._targets(target_x)).to_host().reshape(-1)

    def variance(self, target_x, m_slab=0):
        return self.var_dev(self._targets(target_x), m_slab=m_slab).to_host().reshape(-1)

    def covariance(self, target_x):
        return self.cov_dev(self._targets(target_x)).to_host()


class Interpolator(HyperParameterSearch):
    """general Gaussian-process regression, no structure assumed in the covariance
    (the reference's class of the same name, interpolators.py:47)"""

    def __init__(self, kernel, nugget=None, context=None):
        self.kernel = kernel
        self.nugget = nugget
        self._context = context
        self._ll_data = None           # (context, x copy, f copy, X device, f device) of the last loglikelihood data

    @property
    def context(self):
        return self._context if self._context is not None else _lib.default_context()

    def update(self, *args, **kwargs):
        """update the kernel's parameters (interpolators.py:56-59)"""
        return self.kernel.update(*args, **kwargs)

    def update_nugget(self, *args, **kwargs):
        """update the nugget's parameters (interpolators.py:61-66)"""
        if self.nugget is None:
            raise RuntimeError('cannot update nugget=None')
        self.nugget.update(*args, **kwargs)

    #--- device-resident fit (additive)

    def fit(self, source_x, source_f=None, verbose=False, riders_x=None):
        """build + factor K(x,x; kernel+nugget) on the GPU and (if source_f is given) solve for alpha.
        riders_x: query points whose K(x*, x) rides through the factorisation (FittedGP.riders_var_dev gives their variance)"""
        stage = _Stage(self.context, verbose)
        gp = FittedGP(self.context, source_x, _flatten(self.kernel), _flatten(self.kernel, self.nugget), stage=stage, source_f=source_f,
                      riders_x=riders_x)
        if source_f is not None:
            stage.begin('compressing observations')       # the banner of interpolators.py:98; the solve rode along with the
            stage.end()                                    # factorisation above (mgpi_posv_dev), nothing is left to do here
        return gp

    def _fit_checked(self, source_x, source_f, verbose, riders_x=None):
        try:
            return self.fit(source_x, source_f, verbose=verbose, riders_x=riders_x)
        except _lib.NotPositiveDefinite as err:
            # the reference's inv() raises LinAlgError on a singular K; a Cholesky also rejects
            # indefinite K, which inv() would silently accept
            raise np.linalg.LinAlgError('source-source covariance is not positive definite (pivot %d)' % err.pivot)

    #--- compress / predict (interpolators.py:72-130)

    def compress(self, source_x, source_f, verbose=False, Verbose=False):
        """inv(Cov(source_x, source_x)) @ source_f  ->  (N,)"""
        verbose |= Verbose
        gp = self._fit_checked(source_x, source_f, verbose)
        return gp.alpha.to_host().reshape(-1)

    def predict(self, target_x, source_x, compressed, verbose=False, Verbose=False):
        """mean at target_x from the compressed representation: K(target, source; kernel) @ compressed"""
        verbose |= Verbose
        ctx = self.context
        stage = _Stage(ctx, verbose)
        xs = _points(target_x)
        x = _points(source_x)
        assert xs.shape[1] == x.shape[1], 'conflicting number of dimensions for x1 and x2'
        a = _lib.as_f64(compressed).reshape(-1)
        assert len(a) == len(x), 'compressed has %d entries for %d source points' % (len(a), len(x))
        desc, count = _lib.make_terms(_flatten(self.kernel))
        stage.begin('constructing %d x %d target-source covariance matrix' % (len(xs), len(x)))
        XS = _lib.DeviceMatrix.from_host(ctx, xs)
        X = _lib.DeviceMatrix.from_host(ctx, x)
        A = _lib.DeviceMatrix.from_host(ctx, a)
        out = _lib.DeviceMatrix(ctx, len(xs), 1)
        stage.end()
        stage.begin('computing conditioned mean')
        # fused: K* is evaluated tile by tile inside the contraction and never stored
        _lib.check(ctx.lib.mgpi_predict_mean_dev(ctx.handle, XS.handle, X.handle, desc, count, A.handle, out.handle))
        stage.end()
        return out.to_host().reshape(-1)

    #--- the train-once / query-many flows of bin/mgpi-compress:60-82 and bin/mgpi-predict:58-94, x resident (additive)

    def compress_many(self, source_x, source_f, kernel_params, verbose=False):
        """one compressed vector per row of kernel_params (kernel parameter vectors in kernel order, what
        bin/mgpi-compress loops over): (P, N).  x and f are uploaded once and the N x N buffer is reused, P kernel-matrix
        builds + factorisations + solves run back to back.  The kernel is left at the LAST row, like the loop upstream."""
        ctx = self.context
        x = _points(source_x)
        f = _lib.as_f64(source_f).reshape(-1)
        assert len(f) == len(x), 'source_f has %d entries for %d source points' % (len(f), len(x))
        rows = np.atleast_2d(np.asarray(kernel_params, dtype=float))
        n = len(x)
        X = _lib.DeviceMatrix.from_host(ctx, x)
        F = _lib.DeviceMatrix.from_host(ctx, f)
        L = _lib.DeviceMatrix(ctx, n, n, wide=True)
        a = _lib.DeviceMatrix(ctx, n, 1)
        out = np.empty((len(rows), n))
        if verbose:
            print('generating compressed data for %d parameter sets' % len(rows))
        for p, row in enumerate(rows):
            self.kernel.update(*row)
            if verbose:
                print('    %s' % self.kernel)
            desc, count = _lib.make_terms(_flatten(self.kernel, self.nugget))
            _lib.check(ctx.lib.mgpi_cov_matrix_dev(ctx.handle, X.handle, None, desc, count, 1, L.handle))
            _lib.check(ctx.lib.mgpi_mat_copy(ctx.handle, F.handle, a.handle))
            try:
                _lib.check(ctx.lib.mgpi_posv_dev(ctx.handle, L.handle, a.handle, None))
            except _lib.NotPositiveDefinite as err:
                raise np.linalg.LinAlgError('source-source covariance is not positive definite (pivot %d) for parameter set %d'
                                            % (err.pivot, p))
            out[p] = a.to_host().reshape(-1)
        return out

    def predict_average(self, target_x, source_x, compressed, kernel_params, verbose=False):
        """mean over the P (compressed vector, kernel parameter row) pairs of predict(): bin/mgpi-predict's loop
        (:75-82) with target_x and source_x uploaded once.  The kernel is left at the last row."""
        ctx = self.context
        xs = _points(target_x)
        x = _points(source_x)
        assert xs.shape[1] == x.shape[1], 'conflicting number of dimensions for x1 and x2'
        comp = np.atleast_2d(np.asarray(compressed, dtype=float))
        rows = np.atleast_2d(np.asarray(kernel_params, dtype=float))
        assert len(comp) == len(rows), '%d compressed datasets for %d parameter sets' % (len(comp), len(rows))
        assert comp.shape[1] == len(x), 'compressed has %d entries for %d source points' % (comp.shape[1], len(x))
        XS = _lib.DeviceMatrix.from_host(ctx, xs)
        X = _lib.DeviceMatrix.from_host(ctx, x)
        A = _lib.DeviceMatrix(ctx, len(x), 1)
        out = _lib.DeviceMatrix(ctx, len(xs), 1)
        mean = 0.0
        if verbose:
            print('generating predictions for %d compressed datasets' % len(comp))
        for alpha, row in zip(comp, rows):
            self.kernel.update(*row)
            desc, count = _lib.make_terms(_flatten(self.kernel))
            A.upload(alpha)
            _lib.check(ctx.lib.mgpi_predict_mean_dev(ctx.handle, XS.handle, X.handle, desc, count, A.handle, out.handle))
            mean = mean + out.to_host().reshape(-1)
        return mean / len(comp)

    #--- condition (interpolators.py:136-204)

    def condition(self, target_x, source_x, source_f, verbose=False, Verbose=False):
        """mean (M,) and covariance (M, M) of the process at target_x given source_f = f(source_x);
        zero-mean prior, Rasmussen & Williams eq 2.19"""
        verbose |= Verbose
        gp = self._fit_checked(source_x, source_f, verbose)
        stage = _Stage(self.context, verbose)
        XS = gp._targets(target_x)
        stage.begin('computing conditioned mean')
        mean = gp.mean_dev(XS)
        stage.end()
        stage.begin('computing conditioned covariance')
        cov = gp.cov_dev(XS)
        stage.end()
        return mean.to_host().reshape(-1), cov.to_host()

    def condition_diag(self, target_x, source_x, source_f, verbose=False, m_slab=0):
        """additive: mean (M,) and pointwise variance (M,) = diag of condition()'s covariance, without
        ever forming the M x M matrix.  While the queries are no more than the sources (and no slab size is asked for) they
        RIDE through the factorisation (FittedGP._fit_with_riders): one pass builds, factors and forward-solves - where that
        pays (riders_pay); otherwise the factor is reused slab by slab (mgpi_predict_var_dev)."""
        xs = _points(target_x)
        ride = m_slab == 0 and riders_pay(len(_points(source_x)), len(xs))
        gp = self._fit_checked(source_x, source_f, verbose, riders_x=xs if ride else None)
        stage = _Stage(self.context, verbose)
        XS = gp.riders[0] if ride else gp._targets(xs)
        stage.begin('computing conditioned mean')
        mean = gp.mean_dev(XS)
        stage.end()
        stage.begin('computing conditioned variance')
        var = gp.riders_var_dev() if ride else gp.var_dev(XS, m_slab=m_slab)
        stage.end()
        return mean.to_host().reshape(-1), var.to_host().reshape(-1)

    #--- pairwise covariance matrix (interpolators.py:208-300)

    @staticmethod
    def _x2cov(x1, x2, kernel, verbose=False):
        """K[i, j] = kernel.cov(x1[i], x2[j])  ->  (n1, n2)"""
        a = _points(x1)
        b = _points(x2)
        assert b.shape[1] == a.shape[1], 'conflicting number of dimensions for x1 and x2'
        ctx = _lib.default_context()
        desc, count = _lib.make_terms(kernel.terms())
        if verbose:
            print('computing covariance matrix')
            t0 = time.time()
        out = np.empty((len(a), len(b)), dtype=np.float64)
        _lib.check(ctx.lib.mgpi_cov_matrix(ctx.handle, a.ctypes.data, len(a), b.ctypes.data, len(b), a.shape[1],
                                           desc, count, out.ctypes.data))
        if verbose:
            print('    time : %.6f sec' % (time.time() - t0))
        return out

    #--- realisations of the conditioned process (interpolators.py:306-324)

    def rvs(self, target_x, source_x, source_f, size=1, rng=None):
        """draw `size` realisations of f(target_x) | source_f: mean + MVN(0, cov/outer(s,s)) * s with
        s = sqrt(diag cov) - the rescaling the reference describes at interpolators.py:319-321.

        The reference's own rvs raises TypeError (its helper lacks @staticmethod), so this
        follows its documented intent.  The correlation matrix is factored on the GPU (Cholesky instead of
        numpy's SVD sampler) and the standard normals come from numpy: `rng` may be a numpy Generator, by
        default the global numpy state is used like np.random.multivariate_normal does."""
        gp = self._fit_checked(source_x, source_f, False)
        ctx = self.context
        XS = gp._targets(target_x)
        m = XS.rows
        mean = gp.mean_dev(XS).to_host().reshape(-1)
        cov = gp.cov_dev(XS)
        svec = _lib.DeviceMatrix(ctx, m, 1)
        _lib.check(ctx.lib.mgpi_mat_diag(ctx.handle, cov.handle, svec.handle))
        var = svec.to_host().reshape(-1)
        if not np.all(var > 0):
            raise np.linalg.LinAlgError('conditioned covariance has a non-positive diagonal entry')
        scales = np.sqrt(var)
        svec.upload(scales)
        _lib.check(ctx.lib.mgpi_mat_unscale(ctx.handle, cov.handle, svec.handle))       # correlation matrix
        corr = _lib.DeviceMatrix(ctx, m, m, wide=True)
        # a conditioned covariance is only positive SEMI-definite where targets sit on (nearly) noise-free
        # sources; like a jittered Cholesky sampler, retry with a growing diagonal shift
        jitter, last = 0.0, None
        for attempt in range(6):
            _lib.check(ctx.lib.mgpi_mat_copy(ctx.handle, cov.handle, corr.handle))
            if jitter > 0.0:
                _lib.check(ctx.lib.mgpi_mat_add_diag(ctx.handle, corr.handle, jitter))
            try:
                _lib.check(ctx.lib.mgpi_potrf_dev(ctx.handle, corr.handle, None))
                last = None
                break
            except _lib.NotPositiveDefinite as err:
                last = err
                jitter = 1e-12 if jitter == 0.0 else jitter * 100.0
        if last is not None:
            raise np.linalg.LinAlgError('conditioned covariance is not positive definite even with jitter %g' % jitter)
        _lib.check(ctx.lib.mgpi_mat_tril(ctx.handle, corr.handle))
        normal = rng.standard_normal if rng is not None else np.random.standard_normal
        z = normal((int(size), m))
        Z = _lib.DeviceMatrix.from_host(ctx, -z, wide=True)           # the gemm computes C -= A B^T
        R = _lib.DeviceMatrix(ctx, int(size), m, wide=True)            # zero-initialised
        _lib.check(ctx.lib.mgpi_gemm_nt_dev(ctx.handle, Z.handle, corr.handle, R.handle, 0))
        return mean + R.to_host() * scales

    sample = rvs

    #--- marginal likelihood (interpolators.py:331-343)

    def loglikelihood(self, source_x, source_f, verbose=False):
        """log marginal likelihood of source_f = f(source_x) under the zero-mean process"""
        value, pivot = self.loglikelihoods(source_x, source_f, [np.array(self.kernel.params, dtype=float)], verbose=verbose)
        if pivot[0] != 0:
            raise AssertionError('covariance is not positive definite!')       # interpolators.py:340
        return float(value[0])

    def resident(self, source_x, source_f):
        """(X, f) uploaded once, for repeated loglikelihoods() calls at fixed data: what the closure of the reference's
        hyper-parameter search holds on to (interpolators.py:369-387)"""
        x = _points(source_x)
        f = _lib.as_f64(source_f).reshape(-1)
        assert len(f) == len(x), 'source_f has %d entries for %d source points' % (len(f), len(x))
        return _lib.DeviceMatrix.from_host(self.context, x), _lib.DeviceMatrix.from_host(self.context, f)

    _LL_CACHE_BYTES = 64 << 20

    def _resident_if_unchanged(self, source_x, source_f):
        """the device copies of the previous call's (x, f) when this call's arrays hold the same values - a hyper-parameter
        search through the reference's own entry point, loglikelihood(source_x, source_f) at changing kernel parameters
        (interpolators.py:331-343, 382), then uploads its data once.  The comparison is by VALUE against a private host
        copy (a caller may mutate its arrays in place between calls; NaNs never compare equal and are uploaded again);
        tables above 64 MB are not kept."""
        x = _points(source_x)
        f = _lib.as_f64(source_f).reshape(-1)
        assert len(f) == len(x), 'source_f has %d entries for %d source points' % (len(f), len(x))
        c = getattr(self, '_ll_data', None)
        if (c is not None and c[0] is self.context and c[1].shape == x.shape and c[2].shape == f.shape
                and np.array_equal(c[1], x) and np.array_equal(c[2], f)):
            return c[3], c[4]
        X, F = _lib.DeviceMatrix.from_host(self.context, x), _lib.DeviceMatrix.from_host(self.context, f)
        self._ll_data = (self.context, x.copy(), f.copy(), X, F) if x.nbytes + f.nbytes <= self._LL_CACHE_BYTES else None
        return X, F

    def loglikelihoods(self, source_x, source_f, kernel_params, resident=None, verbose=False):
        """additive: the log marginal likelihood for EVERY row of kernel_params (kernel parameter vectors in kernel
        order, like Kernel.update(*row)) in one device call - the walkers of sample_kernel, the probes of
        optimize_kernel.  Returns (values (B,), pivots (B,)): a parameter set whose covariance is not positive definite
        gets -inf and its failing pivot, the others are unaffected.  The kernel's own parameters are left as they were.
        `resident` = self.resident(source_x, source_f) skips the upload of x and f."""
        rows = np.atleast_2d(np.asarray(kernel_params, dtype=float))
        X, F = resident if resident is not None else self._resident_if_unchanged(source_x, source_f)
        ctx = self.context
        keep = np.array(self.kernel.params, dtype=float)
        try:
            term_lists = []
            for row in rows:
                self.kernel.update(*row)
                term_lists.append(_flatten(self.kernel, self.nugget))
        finally:
            self.kernel.update(*keep)
        desc, nterms, batch = _lib.make_terms_batch(term_lists)
        out = np.empty(batch, dtype=np.float64)
        info = np.zeros(batch, dtype=np.int32)
        if verbose:
            print('computing %d loglikelihood%s of %d samples' % (batch, '' if batch == 1 else 's', X.rows))
            t0 = time.time()
        _lib.check(ctx.lib.mgpi_loglikelihood_batch(ctx.handle, X.handle, F.handle, desc, nterms, batch, out.ctypes.data,
                                                    info.ctypes.data))
        if verbose:
            print('    time : %.6f sec' % (time.time() - t0))
        return out, info
