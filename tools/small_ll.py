#!/usr/bin/env python3
"""developer probe: per-call cost of Interpolator.loglikelihood at the sizes of a hyper-parameter search"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgpi_amd as mgpi
rng = np.random.default_rng(0)
for n in (338, 1311):
    x = rng.random((n, 2)); f = np.sin(4 * x[:, 0]) + x[:, 1]
    it = mgpi.Interpolator(mgpi.SquaredExponentialKernel(1.0, 0.2, 0.2) + mgpi.WhiteNoiseKernel(1e-2))
    for _ in range(20): it.loglikelihood(x, f)
    t0 = time.perf_counter()
    for _ in range(200): it.loglikelihood(x, f)
    print('N=%d loglikelihood %.1f us per call' % (n, (time.perf_counter() - t0) / 200 * 1e6))
