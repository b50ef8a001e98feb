#!/usr/bin/env python3
"""developer probe: how much slower does a small factorisation (the panel chain: 64-column step kernels) run while a big
trailing-update gemm occupies the GPU from another context's stream?  argv: n_small (default 1024) n_big (16384) k_big (1024)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgpi_amd as mgpi
from mgpi_amd import _lib

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nbig = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
kbig = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
a, b = mgpi.Context(device=0), mgpi.Context(device=0)
rng = np.random.default_rng(0)
A = _lib.DeviceMatrix.from_host(a, rng.standard_normal((nbig, kbig)), wide=True)
C = _lib.DeviceMatrix(a, nbig, nbig, wide=True)
x = rng.random((ns, 3))
k = mgpi.SquaredExponentialKernel(1.0, 0.3, 0.3, 0.3) + mgpi.WhiteNoiseKernel(0.1)
desc = _lib.make_terms(k.terms())
X = _lib.DeviceMatrix.from_host(b, x)
L = _lib.DeviceMatrix(b, ns, ns, wide=True)


def small():
    _lib.check(b.lib.mgpi_cov_matrix_dev(b.handle, X.handle, None, desc[0], desc[1], 1, L.handle))
    b.sync()
    t0 = time.perf_counter()
    _lib.check(b.lib.mgpi_potrf_dev(b.handle, L.handle, None))
    return (time.perf_counter() - t0) * 1e3


for _ in range(3):
    small()
alone = min(small() for _ in range(10))
a.tic()
_lib.check(a.lib.mgpi_gemm_nt_dev(a.handle, A.handle, A.handle, C.handle, 1))
big_ms = a.toc()
loaded = []
for _ in range(6):
    for _ in range(4):          # keep the big stream busy for the whole small factorisation
        _lib.check(a.lib.mgpi_gemm_nt_dev(a.handle, A.handle, A.handle, C.handle, 1))
    time.sleep(0.002)           # the first big launch is running
    loaded.append(small())
    a.sync()
print('N=%d potrf alone %.3f ms; beside a lower-syrk n=%d K=%d (%.2f ms per launch): %s ms' %
      (ns, alone, nbig, kbig, big_ms, ' '.join('%.3f' % v for v in loaded)))
