#!/usr/bin/env python3
"""developer probe: repeated launches of the 64-wide block solve, to separate cold (instruction fetch) from warm time"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgpi_amd as mgpi
from mgpi_amd import _lib
ctx = mgpi.default_context()
rng = np.random.default_rng(0)
n, m = 64, int(sys.argv[1]) if len(sys.argv) > 1 else 4096
K = np.eye(n) * 4 + 0.1 * rng.random((n, n)); K = K @ K.T
L = _lib.DeviceMatrix.from_host(ctx, K, wide=True)
_lib.check(ctx.lib.mgpi_potrf_dev(ctx.handle, L.handle, None))
B = _lib.DeviceMatrix.from_host(ctx, rng.random((n, m)), wide=True)
for trans in (0, 1):
    ts = []
    for i in range(12):
        ctx.tic(); _lib.check(ctx.lib.mgpi_trsm_dev(ctx.handle, L.handle, B.handle, trans)); ts.append(ctx.toc() * 1e3)
    print('trsm_cols trans=%d, 64 x %d: us per launch:' % (trans, m), ' '.join('%.1f' % t for t in ts))
ts = []
for i in range(12):
    L2 = _lib.DeviceMatrix.from_host(ctx, K, wide=True)
    ctx.sync(); ctx.tic(); st = ctx.lib.mgpi_potrf_dev(ctx.handle, L2.handle, None); ts.append(ctx.toc() * 1e3)
print('potrf n=64 (potf2 + logdet + sync): us:', ' '.join('%.1f' % t for t in ts))
