#!/usr/bin/env python3
"""developer probe: NT vs NN gemm rates at the shapes of the variance sweep (m x 4096 x 512)"""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgpi_amd as mgpi
from mgpi_amd import _lib

ctx = mgpi.default_context()
rng = np.random.default_rng(0)
k = int(sys.argv[1]) if len(sys.argv) > 1 else 512
for m, n in [(32768, 4096), (16384, 4096), (4096, 4096), (32768, 512)]:
    A = _lib.DeviceMatrix.from_host(ctx, rng.standard_normal((m, k)), wide=True)
    Bt = _lib.DeviceMatrix.from_host(ctx, rng.standard_normal((n, k)), wide=True)      # NT: B given as n x k
    Bn = _lib.DeviceMatrix.from_host(ctx, rng.standard_normal((k, n)), wide=True)      # NN: B given as k x n
    C = _lib.DeviceMatrix(ctx, m, n, wide=True)
    for name, call in [('NT', lambda: ctx.lib.mgpi_gemm_nt_dev(ctx.handle, A.handle, Bt.handle, C.handle, 0)),
                       ('NN', lambda: ctx.lib.mgpi_gemm_nn_dev(ctx.handle, A.handle, Bn.handle, C.handle))]:
        best = 1e9
        for _ in range(4):
            ctx.tic()
            _lib.check(call())
            best = min(best, ctx.toc())
        print('%s m=%d n=%d k=%d: %.3f ms  %.1f TFLOP/s' % (name, m, n, k, best, 2.0 * m * n * k / best / 1e9), flush=True)
