#!/usr/bin/env python3
"""developer driver: factor once, then a few single-rhs solves (for rocprofv3 --kernel-trace --stats)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgpi_amd as mgpi
from mgpi_amd import _lib
from bench import make_inputs, WORKLOADS

n_side, d, m, seed = WORKLOADS[sys.argv[1] if len(sys.argv) > 1 else 'c3']
x, f, xs, ell = make_inputs(n_side, m, seed)
it = mgpi.Interpolator(mgpi.SquaredExponentialKernel(1.0, *ell), nugget=mgpi.WhiteNoiseKernel(1e-2))
gp = it.fit(x)
ctx = it.context
for _ in range(4):
    ctx.tic()
    gp.solve(f)
    print('potrs %.3f ms' % ctx.toc(), flush=True)
