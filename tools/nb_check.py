import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgpi_amd as mgpi
rng = np.random.default_rng(5)
n = 3000
x = rng.random((n, 3)); f = np.sin(3 * x[:, 0]) + x[:, 1]
k = mgpi.SquaredExponentialKernel(1.0, 0.2, 0.2, 0.2); nug = mgpi.WhiteNoiseKernel(1e-2)
K = mgpi.Interpolator._x2cov(x, x, k + nug)
ref = np.linalg.solve(K, f)
for nb in (128, 256, 384, 512, 640, 1024, 2048):
    ctx = mgpi.Context(nb=nb)
    it = mgpi.Interpolator(k, nugget=nug, context=ctx)
    a = it.compress(x, f)
    mean, var = it.condition_diag(x[:200] + 0.01, x, f)
    print('nb=%d alpha err %.2e resid %.2e var[min,max]=%.3e %.3e' % (nb, np.max(np.abs(a - ref)) / np.max(np.abs(ref)), np.max(np.abs(K @ a - f)), var.min(), var.max()), flush=True)
