#!/usr/bin/env python3
"""developer probe: kernel-matrix build at large D (argv: D N), register loop (default) vs MGPI_COV_SHUFFLE=1"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgpi_amd as mgpi
from mgpi_amd import _lib
from oracle import mgpi_oracle as oracle
d = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = int(sys.argv[2]) if len(sys.argv) > 2 else 8192
ctx = mgpi.default_context()
rng = np.random.default_rng(0)
x = rng.random((n, d))
x[5] = x[3]
ell = tuple(0.8 + 0.05 * k for k in range(d))
k = mgpi.SquaredExponentialKernel(1.3, *ell) + mgpi.WhiteNoiseKernel(0.1)
desc = _lib.make_terms(k.terms())
X = _lib.DeviceMatrix.from_host(ctx, x)
K = _lib.DeviceMatrix(ctx, n, n, wide=True)
for rep in range(4):
    ctx.tic()
    _lib.check(ctx.lib.mgpi_cov_matrix_dev(ctx.handle, X.handle, None, desc[0], desc[1], 0, K.handle))
    ms = ctx.toc()
got = K.view(0, 0, 128, 128 * ((n + 127) // 128)).to_host()[:, :n]
ref = oracle.x2cov([oracle.term_se(1.3, *ell), oracle.term_white(0.1)], x[:128], x)
print('D=%d N=%d %s: %.3f ms, %.2f TB/s of writes, max rel diff vs oracle %.1e (first 128 rows)'
      % (d, n, 'MGPI_COV_SHUFFLE=' + os.environ.get('MGPI_COV_SHUFFLE', 'default'), ms, n * n * 8 / ms * 1e3 / 1e12,
         np.max(np.abs(got - ref) / np.abs(ref))))
