#!/usr/bin/env python3
"""C3 posterior variance (V = L^-1 K*^T, column norms) with and without the look-ahead of the blocked solve (developer probe)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgpi_amd as mgpi
from bench import make_workload

x, f, xs, ell, sigma, nugget_sigma = make_workload(sys.argv[1] if len(sys.argv) > 1 else 'c3')
it = mgpi.Interpolator(mgpi.SquaredExponentialKernel(sigma, *ell), nugget=mgpi.WhiteNoiseKernel(nugget_sigma))
gp = it.fit(x, f)
ctx = it.context
for on in (True, False, True, False):
    ctx.set_lookahead(on)
    ctx.tic()
    v = gp.variance(xs)
    print('look-ahead %s: variance %.2f ms' % (on, ctx.toc()), flush=True)
