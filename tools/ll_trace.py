#!/usr/bin/env python3
"""developer driver: a few loglikelihood calls at one size (for rocprofv3 --kernel-trace): python3 tools/ll_trace.py N [reps]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgpi_amd as mgpi
n = int(sys.argv[1]) if len(sys.argv) > 1 else 7933
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 4
rng = np.random.default_rng(0)
x = rng.random((n, 2)); f = np.sin(3 * x[:, 0]) + x[:, 1]
it = mgpi.Interpolator(mgpi.SquaredExponentialKernel(1.0, 0.2, 0.2), nugget=mgpi.WhiteNoiseKernel(1e-2))
for r in range(reps):
    t0 = time.perf_counter()
    v = it.loglikelihood(x, f)
    print('loglikelihood N=%d: %.3f ms (%.6g)' % (n, (time.perf_counter() - t0) * 1e3, v), flush=True)
