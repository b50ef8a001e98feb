#!/usr/bin/env python3
"""developer probe: MFMA/HBM peaks and stage timings of condition+predict at a few N (GPU box only)"""
import ctypes
import sys
import time

import numpy as np

import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgpi_amd as mgpi
from mgpi_amd import _lib


def grid(n_side, seed):
    rng = np.random.default_rng(seed)
    axes = [(np.arange(s) + 0.5) / s for s in n_side]
    x = np.transpose([a.ravel() for a in np.meshgrid(*axes, indexing='ij')])
    x = x + rng.uniform(-0.1, 0.1, x.shape) / np.array(n_side)
    f = np.sin(3 * x[:, 0]) * np.cos(2 * x[:, 1]) + 0.5 * np.sin(4 * x[:, 2])
    return x, f, tuple(2.0 / np.array(n_side))


def main():
    ctx = mgpi.default_context()
    tf = ctypes.c_double()
    _lib.check(ctx.lib.mgpi_probe_mfma_f64(ctx.handle, 20000, ctypes.byref(tf)))
    r, w = ctypes.c_double(), ctypes.c_double()
    _lib.check(ctx.lib.mgpi_probe_hbm(ctx.handle, 2 << 30, 5, ctypes.byref(r), ctypes.byref(w)))
    print('probe: fp64 MFMA %.1f TFLOP/s, HBM read %.0f GB/s write %.0f GB/s' % (tf.value, r.value, w.value), flush=True)

    sizes = [((32, 16, 16), 2), ((32, 32, 32), 3)] if len(sys.argv) < 2 else [(tuple(int(v) for v in sys.argv[1].split('x')), 9)]
    for n_side, seed in sizes:
        x, f, ell = grid(n_side, seed)
        n = len(x)
        m = 4096
        xs = np.random.default_rng(1).random((m, 3))
        k = mgpi.SquaredExponentialKernel(1.0, *ell)
        nug = mgpi.WhiteNoiseKernel(1e-2)
        desc = _lib.make_terms(k.terms() + nug.terms())
        kdesc = _lib.make_terms(k.terms())
        X = _lib.DeviceMatrix.from_host(ctx, x)
        XS = _lib.DeviceMatrix.from_host(ctx, xs)
        F = _lib.DeviceMatrix.from_host(ctx, f)
        L = _lib.DeviceMatrix(ctx, n, n, wide=True)
        for rep in range(2):
            ctx.set_profile(rep == 1)
            ctx.tic()
            _lib.check(ctx.lib.mgpi_cov_matrix_dev(ctx.handle, X.handle, None, desc[0], desc[1], 1, L.handle))
            t_cov = ctx.toc()
            t0 = time.time()
            _lib.check(ctx.lib.mgpi_potrf_dev(ctx.handle, L.handle, None))
            t_potrf = (time.time() - t0) * 1e3
            A = _lib.DeviceMatrix.from_host(ctx, f)
            ctx.tic()
            _lib.check(ctx.lib.mgpi_potrs_dev(ctx.handle, L.handle, A.handle))
            t_potrs = ctx.toc()
            mean = _lib.DeviceMatrix(ctx, m, 1)
            ctx.tic()
            _lib.check(ctx.lib.mgpi_predict_mean_dev(ctx.handle, XS.handle, X.handle, kdesc[0], kdesc[1], A.handle, mean.handle))
            t_mean = ctx.toc()
            var = _lib.DeviceMatrix(ctx, m, 1)
            ctx.tic()
            _lib.check(ctx.lib.mgpi_predict_var_dev(ctx.handle, XS.handle, X.handle, kdesc[0], kdesc[1], L.handle, 0, var.handle))
            t_var = ctx.toc()
            flops = n**3 / 3.0
            print('N=%d rep%d: cov %.2f ms | potrf %.1f ms (%.1f TF) | potrs %.2f ms | mean(M=%d) %.2f ms | var %.1f ms (%.1f TF)'
                  % (n, rep, t_cov, t_potrf, flops / t_potrf / 1e9, t_potrs, m, t_mean, t_var, n * n * m / t_var / 1e9), flush=True)
            if rep == 1:
                syrk_ms, syrk_n = ctx.get_profile(0)
                pan_ms, pan_n = ctx.get_profile(1)
                nb = 512
                tflops = sum((n - kk * nb) * (n - kk * nb + 1) * nb for kk in range(1, n // nb))
                print('   profiled: syrk %.1f ms over %d launches (%.1f TF on algorithmic flops) | panels %.1f ms over %d'
                      % (syrk_ms, syrk_n, tflops / syrk_ms / 1e9, pan_ms, pan_n), flush=True)
        del L, X, XS, F


if __name__ == '__main__':
    main()
