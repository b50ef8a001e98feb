#!/usr/bin/env python3
"""developer probe for the panel chain: potrf wall time at N = 2048 / 8192 / 32768 (build excluded), loglikelihood per
call at N = 338 / 1311, checked against numpy's Cholesky at N = 2048.  GPU box only."""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgpi_amd as mgpi
from mgpi_amd import _lib
from bench import make_inputs

ctx = mgpi.default_context()
sizes = [((16, 16, 8), 30), ((32, 16, 16), 20), ((32, 32, 32), 3)]
if len(sys.argv) > 1 and sys.argv[1] == 'small':
    sizes = sizes[:2]
for side, reps in sizes:
    x, f, xs, ell = make_inputs(side, 16, 1)
    n = len(x)
    k = mgpi.SquaredExponentialKernel(1.0, *ell) + mgpi.WhiteNoiseKernel(1e-2)
    desc = _lib.make_terms(k.terms())
    X = _lib.DeviceMatrix.from_host(ctx, x)
    L = _lib.DeviceMatrix(ctx, n, n, wide=True)
    best = 1e30
    for rep in range(reps + 2):
        _lib.check(ctx.lib.mgpi_cov_matrix_dev(ctx.handle, X.handle, None, desc[0], desc[1], 1, L.handle))
        ctx.sync()
        t0 = time.perf_counter()
        ld = ctypes.c_double()
        _lib.check(ctx.lib.mgpi_potrf_dev(ctx.handle, L.handle, ctypes.byref(ld)))
        dt = time.perf_counter() - t0
        if rep >= 2:
            best = min(best, dt)
    msg = 'N=%d potrf %.3f ms (%.1f TFLOP/s)' % (n, best * 1e3, n**3 / 3 / best / 1e12)
    if n <= 2048:
        K = mgpi.Interpolator._x2cov(x, x, k)
        Lref = np.linalg.cholesky(K)
        _lib.check(ctx.lib.mgpi_mat_tril(ctx.handle, L.handle))
        got = L.to_host()
        msg += ' | max |L - L_lapack| / max|L| = %.2e, logdet rel err %.1e' % (np.max(np.abs(got - Lref)) / np.max(np.abs(Lref)),
                                                                         abs(ld.value - 2 * np.sum(np.log(np.diag(Lref)))) / abs(ld.value))
    print(msg, flush=True)
    del L

rng = np.random.default_rng(0)
for n in (338, 1311):
    x = rng.random((n, 2)); f = np.sin(4 * x[:, 0]) + x[:, 1]
    it = mgpi.Interpolator(mgpi.SquaredExponentialKernel(1.0, 0.2, 0.2) + mgpi.WhiteNoiseKernel(1e-2))
    for _ in range(20): it.loglikelihood(x, f)
    t0 = time.perf_counter()
    for _ in range(200): it.loglikelihood(x, f)
    print('N=%d loglikelihood %.1f us per call' % (n, (time.perf_counter() - t0) / 200 * 1e6), flush=True)
