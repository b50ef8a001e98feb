#!/usr/bin/env python3
"""developer driver: a few factorisations at one size (for rocprofv3 --kernel-trace): python3 tools/potrf_trace.py N [reps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgpi_amd as mgpi
from mgpi_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rng = np.random.default_rng(0)
x = rng.random((n, 3))
ctx = mgpi.default_context()
terms = (mgpi.SquaredExponentialKernel(1.0, 0.2, 0.2, 0.2) + mgpi.WhiteNoiseKernel(1e-2)).terms()
desc, count = _lib.make_terms(terms)
X = _lib.DeviceMatrix.from_host(ctx, x)
K = _lib.DeviceMatrix(ctx, n, n, wide=True)
for r in range(reps):
    _lib.check(ctx.lib.mgpi_cov_matrix_dev(ctx.handle, X.handle, None, desc, count, 1, K.handle))
    ctx.sync()
    t0 = time.perf_counter()
    _lib.check(ctx.lib.mgpi_potrf_dev(ctx.handle, K.handle, None))
    print('potrf N=%d: %.3f ms' % (n, (time.perf_counter() - t0) * 1e3), flush=True)
