#!/usr/bin/env python3
"""developer probe: `reps` factorisations at one size (for rocprofv3 --kernel-trace --stats).  argv: n_side reps"""
import ctypes, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgpi_amd as mgpi
from mgpi_amd import _lib
from bench import make_inputs
side = tuple(int(v) for v in sys.argv[1].split('x'))
reps = int(sys.argv[2])
ctx = mgpi.default_context()
x, f, xs, ell = make_inputs(side, 16, 1)
n = len(x)
k = mgpi.SquaredExponentialKernel(1.0, *ell) + mgpi.WhiteNoiseKernel(1e-2)
desc = _lib.make_terms(k.terms())
X = _lib.DeviceMatrix.from_host(ctx, x)
L = _lib.DeviceMatrix(ctx, n, n, wide=True)
ts = []
for rep in range(reps):
    _lib.check(ctx.lib.mgpi_cov_matrix_dev(ctx.handle, X.handle, None, desc[0], desc[1], 1, L.handle))
    ctx.sync()
    t0 = time.perf_counter()
    rc = ctx.lib.mgpi_potrf_dev(ctx.handle, L.handle, None)
    if not os.environ.get('MGPI_PROBE_IGNORE_STATUS'):      # timing experiments that break the arithmetic on purpose
        _lib.check(rc)
    ts.append(time.perf_counter() - t0)
print('N=%d potrf best %.3f ms median %.3f ms' % (n, min(ts) * 1e3, sorted(ts)[len(ts) // 2] * 1e3))
