#!/usr/bin/env python3
"""developer probe: the fused predict-mean kernel and the lower K build, device-resident, at the C5 / C3 shapes.
MGPI_PREDICT_MFMA=1 selects the matrix-core variant of the contraction, MGPI_COV_FAST=0 the general term loop."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgpi_amd as mgpi
from mgpi_amd import _lib
import bench
ctx = mgpi.default_context()
x, f, xs, ell, sigma, nug = bench.make_workload('c5')
k = mgpi.SquaredExponentialKernel(sigma, *ell)
kd = _lib.make_terms(k.terms())
X = _lib.DeviceMatrix.from_host(ctx, x); XS = _lib.DeviceMatrix.from_host(ctx, xs)
rng = np.random.default_rng(0)
alpha = rng.standard_normal(len(x))
A = _lib.DeviceMatrix.from_host(ctx, alpha)
out = _lib.DeviceMatrix(ctx, len(xs), 1)
for rep in range(3):
    ctx.tic()
    _lib.check(ctx.lib.mgpi_predict_mean_dev(ctx.handle, XS.handle, X.handle, kd[0], kd[1], A.handle, out.handle))
    ms = ctx.toc()
pairs = len(x) * len(xs)
got = out.to_host().reshape(-1)
from oracle import mgpi_oracle as oracle
pick = rng.choice(len(xs), 300, replace=False)
want = oracle.x2cov_chunked([oracle.term_se(sigma, *ell)], xs[pick], x) @ alpha
print('C5 predict mean: %.2f ms for %.3e pairs = %.3e pairs/s = %.1f vector-lane cycles per pair | max rel err vs oracle %.1e'
      % (ms, pairs, pairs / ms * 1e3, 256 * 4 * 16 * 2.4e9 / (pairs / ms * 1e3), np.max(np.abs(got[pick] - want)) / np.max(np.abs(want))), flush=True)
x3, f3, xs3, ell3, s3, n3 = bench.make_workload('c3')
k3 = mgpi.SquaredExponentialKernel(s3, *ell3) + mgpi.WhiteNoiseKernel(n3)
d3 = _lib.make_terms(k3.terms())
X3 = _lib.DeviceMatrix.from_host(ctx, x3)
L = _lib.DeviceMatrix(ctx, len(x3), len(x3), wide=True)
for rep in range(3):
    ctx.tic()
    _lib.check(ctx.lib.mgpi_cov_matrix_dev(ctx.handle, X3.handle, None, d3[0], d3[1], 1, L.handle))
    ms = ctx.toc()
n = len(x3)
entries = (n // 64) * (n // 64 + 1) // 2 * 4096
print('C3 lower K build: %.3f ms, %.3e entries, %.2f TB/s of writes, %.1f vector-lane cycles per entry' % (ms, entries, entries * 8 / ms * 1e3 / 1e12, 256 * 4 * 16 * 2.4e9 / (entries / ms * 1e3)), flush=True)
