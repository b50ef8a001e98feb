#!/usr/bin/env python3
"""developer probe: fused predict mean at large D (argv: D)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgpi_amd as mgpi
from mgpi_amd import _lib
d = int(sys.argv[1]); n, m = 8192, 65536
ctx = mgpi.default_context(); rng = np.random.default_rng(0)
x, xs, alpha = rng.random((n, d)), rng.random((m, d)), rng.standard_normal(n)
k = mgpi.SquaredExponentialKernel(1.3, *([0.9] * d))
kd = _lib.make_terms(k.terms())
X, XS, A = _lib.DeviceMatrix.from_host(ctx, x), _lib.DeviceMatrix.from_host(ctx, xs), _lib.DeviceMatrix.from_host(ctx, alpha)
out = _lib.DeviceMatrix(ctx, m, 1)
for rep in range(3):
    ctx.tic(); _lib.check(ctx.lib.mgpi_predict_mean_dev(ctx.handle, XS.handle, X.handle, kd[0], kd[1], A.handle, out.handle)); ms = ctx.toc()
print('D=%d predict mean N=%d M=%d: %.3f ms = %.3e pairs/s' % (d, n, m, ms, n * m / ms * 1e3))
