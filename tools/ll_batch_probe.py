#!/usr/bin/env python3
"""developer probe: loglikelihood per call and batched (B walkers in one device call) at the reference's sizes"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgpi_amd as mgpi
rng = np.random.default_rng(0)
for n in (16, 90, 338, 1311, 2613):
    x = rng.random((n, 2)); f = np.sin(4 * x[:, 0]) + x[:, 1]
    k = mgpi.SquaredExponentialKernel(1.0, 0.2, 0.2) + mgpi.WhiteNoiseKernel(1e-2)
    it = mgpi.Interpolator(k)
    base = np.array(k.params, dtype=float)
    res = it.resident(x, f)
    for _ in range(10): it.loglikelihood(x, f)
    reps = 100
    t0 = time.perf_counter()
    for _ in range(reps): it.loglikelihood(x, f)
    single = (time.perf_counter() - t0) / reps
    line = 'N=%4d: loglikelihood %.1f us per call' % (n, single * 1e6)
    for B in (1, 8, 32):
        rows = base[None, :] * (1 + 0.05 * rng.standard_normal((B, len(base))))
        for _ in range(5): it.loglikelihoods(x, f, rows, resident=res)
        t0 = time.perf_counter()
        for _ in range(reps // 2): it.loglikelihoods(x, f, rows, resident=res)
        dt = (time.perf_counter() - t0) / (reps // 2)
        line += ' | B=%d resident: %.1f us (%.1f us per member, %.1fx the serial loop)' % (B, dt * 1e6, dt / B * 1e6, single * B / dt)
    print(line, flush=True)
