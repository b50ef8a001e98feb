#!/usr/bin/env python3
"""developer probe (round 3): the 256 x 128 trailing-update tile (MGPI_GEMM_WIDE=1) against the shipped 128 x 128 one.
Run twice, with and without the switch: exact-integer correctness of the lower / general NT and the NN form, then rates."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgpi_amd as mgpi
from mgpi_amd import _lib

ctx = mgpi.default_context()
lib, h = ctx.lib, ctx.handle
rng = np.random.default_rng(0)


def dev(a):
    return _lib.DeviceMatrix.from_host(ctx, a, wide=True)


# correctness: integers are exact in fp64
m, k = 4096, 48
A = rng.integers(-3, 4, size=(m, k)).astype(float)
C0 = rng.integers(-5, 6, size=(m, m)).astype(float)
dA, dC = dev(A), dev(C0)
_lib.check(lib.mgpi_gemm_nt_dev(h, dA.handle, dA.handle, dC.handle, 1))
got, want = dC.to_host(), C0 - A @ A.T
low = np.tril(np.ones((m // 128, m // 128))).repeat(128, 0).repeat(128, 1).astype(bool)
assert np.array_equal(got[low], want[low]), 'lower NT'
m, n, k = 8192, 8192, 32
A = rng.integers(-3, 4, size=(m, k)).astype(float)
B = rng.integers(-3, 4, size=(n, k)).astype(float)
C0 = rng.integers(-5, 6, size=(m, n)).astype(float)
dA, dB, dC = dev(A), dev(B), dev(C0)
_lib.check(lib.mgpi_gemm_nt_dev(h, dA.handle, dB.handle, dC.handle, 0))
assert np.array_equal(dC.to_host(), C0 - A @ B.T), 'general NT'
dBn, dC2 = dev(np.ascontiguousarray(B.T)), dev(C0)
_lib.check(lib.mgpi_gemm_nn_dev(h, dA.handle, dBn.handle, dC2.handle))
assert np.array_equal(dC2.to_host(), C0 - A @ B.T), 'NN'
print('exact-integer checks passed (MGPI_GEMM_WIDE=%s)' % os.environ.get('MGPI_GEMM_WIDE', '0'), flush=True)
del dA, dB, dC, dBn, dC2

for n, k in [(32256, 1024), (32256, 512), (16384, 1024), (16384, 512), (8192, 512)]:
    P = _lib.DeviceMatrix.from_host(ctx, rng.standard_normal((n, k)), wide=True)
    C = _lib.DeviceMatrix(ctx, n, n, wide=True)
    best = 1e9
    for _ in range(4):
        ctx.tic()
        _lib.check(lib.mgpi_gemm_nt_dev(h, P.handle, P.handle, C.handle, 1))
        best = min(best, ctx.toc())
    print('syrk n=%d k=%d: %.3f ms  %.1f TFLOP/s' % (n, k, best, n * (n + 1.0) * k / best / 1e9), flush=True)
    del P, C
for m, n, k in [(32768, 4096, 2048), (32768, 4096, 512), (16384, 4096, 2048)]:
    A = _lib.DeviceMatrix.from_host(ctx, rng.standard_normal((m, k)), wide=True)
    Bn = _lib.DeviceMatrix.from_host(ctx, rng.standard_normal((k, n)), wide=True)
    C = _lib.DeviceMatrix(ctx, m, n, wide=True)
    best = 1e9
    for _ in range(4):
        ctx.tic()
        _lib.check(lib.mgpi_gemm_nn_dev(h, A.handle, Bn.handle, C.handle))
        best = min(best, ctx.toc())
    print('NN m=%d n=%d k=%d: %.3f ms  %.1f TFLOP/s' % (m, n, k, best, 2.0 * m * n * k / best / 1e9), flush=True)
    del A, Bn, C
