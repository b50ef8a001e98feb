#!/usr/bin/env python3
"""developer probe: what the vendor library (rocBLAS / hipBLASLt behind torch.matmul) reaches on fp64 GEMMs of the
shapes of the hot path, next to this repository's hand-written kernel.  Evidence for DESIGN.md only - torch is not
part of the product path."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgpi_amd as mgpi
from mgpi_amd import _lib

dev = torch.device('cuda', 0)
ctx = mgpi.default_context()
rng = np.random.default_rng(0)


def vendor(m, n, k, reps=5):
    A = torch.randn(m, k, dtype=torch.float64, device=dev)
    B = torch.randn(n, k, dtype=torch.float64, device=dev)
    C = torch.randn(m, n, dtype=torch.float64, device=dev)
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        torch.addmm(C, A, B.t(), beta=1.0, alpha=-1.0, out=C)       # C -= A B^T, like the trailing update
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    return 2.0 * m * n * k / best / 1e12


def ours(m, n, k, lower, reps=5):
    A = _lib.DeviceMatrix.from_host(ctx, rng.standard_normal((m, k)), wide=True)
    B = A if lower else _lib.DeviceMatrix.from_host(ctx, rng.standard_normal((n, k)), wide=True)
    C = _lib.DeviceMatrix(ctx, m, n, wide=True)
    best = 1e9
    for _ in range(reps):
        ctx.tic()
        _lib.check(ctx.lib.mgpi_gemm_nt_dev(ctx.handle, A.handle, B.handle, C.handle, int(lower)))
        best = min(best, ctx.toc())
    flops = (m * (m + 1.0) * k) if lower else 2.0 * m * n * k
    return flops / best / 1e9


for (m, n, k) in [(16384, 16384, 512), (32768, 4096, 512), (8192, 8192, 512), (8192, 8192, 8192)]:
    line = 'm=%5d n=%5d k=%5d : vendor dgemm %.1f TFLOP/s | this repo, full rectangle %.1f' % (m, n, k, vendor(m, n, k), ours(m, n, k, False))
    if m == n:
        line += ' | lower tiles only (syrk form, algorithmic flops) %.1f' % ours(m, n, k, True)
    print(line, flush=True)
