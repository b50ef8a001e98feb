#!/usr/bin/env python3
"""developer probe: build + factor at N = argv[1] (default 65536, D = 4: BASELINE config C4 on ONE GPU), twice"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgpi_amd as mgpi
from bench import make_inputs

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
side = {131072: (16, 16, 16, 32), 98304: (16, 16, 16, 24), 65536: (16, 16, 16, 16), 52224: (32, 34, 48), 49152: (16, 16, 16, 12), 40960: (32, 32, 40), 32768: (32, 32, 32)}[n]
x, f, xs, ell = make_inputs(side, 4096, 3)
it = mgpi.Interpolator(mgpi.SquaredExponentialKernel(1.0, *ell), nugget=mgpi.WhiteNoiseKernel(1e-2))
ctx = it.context
for rep in range(1 if n > 65536 else 2):
    ctx.set_profile(True)
    t0 = time.perf_counter()
    gp = it.fit(x)
    ctx.sync()
    dt = time.perf_counter() - t0
    syrk_ms, launches = ctx.get_profile(0)
    flops = ctx.get_profile_flops(0)
    potrf_ms, _ = ctx.get_profile(2)
    print('N=%d rep %d: fit %.3f s | potrf %.1f ms (%.1f TFLOP/s on N^3/3) | lower-syrk launches %.1f ms, %d, %.1f TFLOP/s'
          % (n, rep, dt, potrf_ms, n**3 / 3 / potrf_ms / 1e9, syrk_ms, launches, flops / syrk_ms / 1e9), flush=True)
    if n > 65536:       # a residual check at the size nothing else runs: alpha = K^-1 f, |K alpha - f| through one more build
        alpha = gp.solve(f).to_host().reshape(-1)
        Kr = mgpi.Interpolator._x2cov(x[:2048], x, it.kernel + it.nugget)
        print('   residual of the first 2048 rows: %.2e' % (np.max(np.abs(Kr @ alpha - f[:2048])) / np.max(np.abs(f))), flush=True)
    del gp
