#!/usr/bin/env python3
"""developer probe: kernel-matrix build rate per kernel type (SE, Matern half-integer closed form, Matern general path)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgpi_amd as mgpi
rng = np.random.default_rng(0)
n = 8192
x = rng.random((n, 2)) * [150, 70] + [0, 30]
for name, k in [('SE', mgpi.SquaredExponentialKernel(0.1, 10, 20)), ('Matern 0.5', mgpi.MaternKernel(0.5, 0.1, 15, 10)),
                ('Matern 1.5', mgpi.MaternKernel(1.5, 0.1, 15, 10)), ('Matern 2', mgpi.MaternKernel(2, 0.1, 15, 10)),
                ('Matern 2.7', mgpi.MaternKernel(2.7, 0.1, 15, 10)), ('Matern 12.3', mgpi.MaternKernel(12.3, 0.1, 15, 10))]:
    it = mgpi.Interpolator(k, nugget=mgpi.WhiteNoiseKernel(1e-3))
    ctx = it.context
    X = mgpi._lib.DeviceMatrix.from_host(ctx, x)
    K = mgpi._lib.DeviceMatrix(ctx, n, n, wide=True)
    desc, count = mgpi._lib.make_terms((k + it.nugget).terms())
    best = 1e9
    for _ in range(3):
        ctx.tic()
        mgpi._lib.check(ctx.lib.mgpi_cov_matrix_dev(ctx.handle, X.handle, None, desc, count, 1, K.handle))
        best = min(best, ctx.toc())
    print('%-12s K build (lower tiles) N=%d: %.3f ms  = %.2f G entries/s' % (name, n, best, n * (n + 1) / 2 / best / 1e6))
