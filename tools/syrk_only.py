#!/usr/bin/env python3
"""developer driver: a few trailing-update (syrk) launches at fixed sizes, for rocprofv3 --pmc runs"""
import sys
import numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgpi_amd as mgpi
from mgpi_amd import _lib

n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
k = int(sys.argv[3]) if len(sys.argv) > 3 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
ctx = mgpi.default_context()
rng = np.random.default_rng(0)
A = _lib.DeviceMatrix.from_host(ctx, rng.standard_normal((n, k)), wide=True)
C = _lib.DeviceMatrix(ctx, n, n, wide=True)
for _ in range(reps):
    ctx.tic()
    _lib.check(ctx.lib.mgpi_gemm_nt_dev(ctx.handle, A.handle, A.handle, C.handle, 1))
    ms = ctx.toc()
    print('syrk n=%d k=%d: %.3f ms  %.1f TFLOP/s (algorithmic n(n+1)k)' % (n, k, ms, n * (n + 1) * k / ms / 1e9), flush=True)
