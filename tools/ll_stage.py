#!/usr/bin/env python3
"""developer probe: stage timings of one loglikelihood call at N = argv[1]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgpi_amd as mgpi
from loglike_timing import table
n = int(sys.argv[1]) if len(sys.argv) > 1 else 14074
x, f, ell = table(n)
f = f - f.mean()
sig = 1.5 * np.std(f)
it = mgpi.Interpolator(mgpi.CombinedKernel(mgpi.SquaredExponentialKernel(sig, *ell), mgpi.WhiteNoiseKernel(sig / 1000)))
it.loglikelihood(x, f)
for _ in range(2):
    t0 = time.perf_counter()
    it.loglikelihood(x, f, verbose=True)
    print('total %.4f s' % (time.perf_counter() - t0))
