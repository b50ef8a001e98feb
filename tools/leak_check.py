#!/usr/bin/env python3
"""developer probe: device memory across a few hundred calls of varying size (pool, cached inverses, workspaces)"""
import os, sys, ctypes
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgpi_amd as mgpi
mgpi._lib.load_library()
hip = ctypes.CDLL(None)      # the HIP runtime the library bound to (one runtime per process, see _lib.py)
def free_gb():
    f, t = ctypes.c_size_t(0), ctypes.c_size_t(0)
    hip.hipMemGetInfo(ctypes.byref(f), ctypes.byref(t))
    return f.value / 2**30
rng = np.random.default_rng(0)
it = mgpi.Interpolator(mgpi.SquaredExponentialKernel(1.0, 0.2, 0.2), nugget=mgpi.WhiteNoiseKernel(1e-2))
nn = mgpi.NearestNeighborInterpolator(it.kernel, nugget=it.nugget)
print('start free %.2f GB' % free_gb())
for rep in range(300):
    n = int(rng.integers(500, 3000))
    x = rng.random((n, 2)); f = np.sin(4 * x[:, 0]) + x[:, 1]; xs = rng.random((200, 2))
    it.loglikelihood(x, f)
    it.condition_diag(xs, x, f)
    if rep % 10 == 0:
        it.condition(xs[:50], x, f); nn.condition_diag(xs, x, f); it.rvs(xs[:20], x, f, size=3)
    if rep % 100 == 99:
        print('rep %d free %.2f GB' % (rep, free_gb()), flush=True)
it.context.trim()
print('after trim free %.2f GB' % free_gb())
