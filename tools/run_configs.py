#!/usr/bin/env python3
"""All BASELINE.json configs on one MI355X through the public Python API (host arrays in, host arrays out):
C1 (N=512 D=2 example-interpolator.ini values), C2, C3, C4 (N=65536 D=4, on ONE GPU here), C5 (stellarcollapse-
shaped table, 1e6 queries, mean + variance).  Prints one line per config; used for DESIGN.md's table."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgpi_amd as mgpi
from bench import make_inputs

def timed(fn, reps=2):
    best = None
    for _ in range(reps):
        t0 = time.perf_counter(); out = fn(); dt = time.perf_counter() - t0
        best = dt if best is None or dt < best else best
    return best, out

def grid_case(name, side, m, seed, reps=2):
    x, f, xs, ell = make_inputs(side, m, seed)
    it = mgpi.Interpolator(mgpi.SquaredExponentialKernel(1.0, *ell), nugget=mgpi.WhiteNoiseKernel(1e-2))
    t_fit, gp = timed(lambda: it.fit(x, f), reps)
    t_mean, mean = timed(lambda: gp.mean(xs), reps)
    t_var, var = timed(lambda: gp.variance(xs), reps)
    n = len(x)
    resid = np.max(np.abs(mgpi.Interpolator(it.kernel + it.nugget).predict(x[:4096], x, gp.alpha.to_host().reshape(-1)) - f[:4096])) / np.max(np.abs(f))
    print('%s N=%d D=%d M=%d: fit (build+potrf+potrs) %.3f s (%.1f TFLOP/s on N^3/3) | mean %.4f s | variance %.3f s (%.1f TFLOP/s on N^2 M) | residual %.1e | var in [%.2e, %.2e]'
          % (name, n, x.shape[1], m, t_fit, n**3 / 3 / t_fit / 1e12, t_mean, t_var, n * n * m / t_var / 1e12, resid, var.min(), var.max()), flush=True)

def c1():
    rng = np.random.default_rng(0)
    g0 = np.linspace(0.0, 150.0, 32); g1 = np.linspace(30.0, 100.0, 32)
    G0, G1 = np.meshgrid(g0, g1, indexing='ij')
    full = np.transpose([G0.ravel(), G1.ravel()])
    ff = 0.05 * np.tanh((full[:, 1] - 60.0) / 20.0) + 0.01 * np.sin(full[:, 0] / 30.0) + 1e-4 * rng.standard_normal(len(full))
    x, f = full[::2], ff[::2]
    xs = np.transpose([rng.uniform(0, 150, 256), rng.uniform(30, 100, 256)])
    it = mgpi.Interpolator(mgpi.SquaredExponentialKernel(0.1, 10., 20.), nugget=mgpi.WhiteNoiseKernel(1e-4))
    t, _ = timed(lambda: it.condition(xs, x, f), 3)
    print('C1 N=512 D=2 M=256: condition (mean + full cov) %.4f s through the Python API' % t, flush=True)

def c5(ds):
    # stellarcollapse-shaped: axes ye(50) x logtemp(136) x logrho(234), per-axis decimation as in sc-eos-test:105-109
    ye = np.linspace(0.035, 0.53, 50)[::ds]; lt = np.linspace(-2.0, 2.4, 136)[::ds]; lr = np.linspace(3.0, 15.4, 234)[::ds]
    Y, T, R = np.meshgrid(ye, lt, lr, indexing='ij')
    loge = 17.0 + 1.1 * T + 0.25 * (R - 9.0) + 0.4 * Y + 0.05 * np.sin(R) * T          # smooth, monotone in logT
    x = np.transpose([Y.ravel(), R.ravel(), loge.ravel()]); f = T.ravel() - np.mean(T)
    sig = 2 * np.std(f)
    ell = tuple(2 * (np.ptp(x[:, k]) / (len(ax) - 1)) for k, ax in enumerate([ye, lr, lt]))
    it = mgpi.Interpolator(mgpi.SquaredExponentialKernel(sig, *ell), nugget=mgpi.WhiteNoiseKernel(sig / 100.))
    q = [np.linspace(x[:, k].min(), x[:, k].max(), 100) for k in range(3)]
    xs = np.transpose([a.ravel() for a in np.meshgrid(*q, indexing='ij')])
    t_fit, gp = timed(lambda: it.fit(x, f), 1)
    t_mean, mean = timed(lambda: gp.mean(xs), 1)
    t_var, var = timed(lambda: gp.variance(xs), 1)
    n, m = len(x), len(xs)
    print('C5 (downsample %d) N=%d D=3 M=%d: fit %.3f s | mean %.3f s | variance %.2f s (%.1f TFLOP/s on N^2 M) | var in [%.2e, %.2e] prior %.2e'
          % (ds, n, m, t_fit, t_mean, t_var, n * n * m / t_var / 1e12, var.min(), var.max(), sig**2), flush=True)

which = sys.argv[1:] or ['c1', 'c2', 'c3', 'c5a']
if 'c1' in which: c1()
if 'c2' in which: grid_case('C2', (32, 16, 16), 4096, 2)
if 'c3' in which: grid_case('C3', (32, 32, 32), 4096, 3)
if 'c4' in which: grid_case('C4 (one GPU)', (16, 16, 16, 16), 4096, 4, reps=1)
if 'c5a' in which: c5(10)
if 'c5b' in which: c5(5)
