#!/usr/bin/env python3
"""developer probe: Interpolator.loglikelihood wall-clock at the training-set sizes of the reference's only
published timings (test/uiuc/uiuc-test-nngp-timing.txt: dense GP, SE(D=2) + WhiteNoise sigma/1000 inside the
kernel, checkerboard-thinned 2-D grid).  Synthetic (muB, T) grid of the same sizes; host arrays in, scalar out."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgpi_amd as mgpi

PUBLISHED = {16: 0.0002, 60: 0.0006, 90: 0.001, 160: 0.003, 338: 0.02, 595: 0.05, 1311: 0.22, 2613: 1.60, 5160: 11.87,
             7933: 38.81, 14074: None, 31527: None}


def table(n):
    """~n training points: 3 of every 4 nodes of a g0 x g1 grid (uiuc-test-nngp-timing:117-150)"""
    side = int(np.ceil(np.sqrt(n / 0.75)))
    g0 = np.linspace(0.0, 600.0, side)
    g1 = np.linspace(30.0, 400.0, side)
    G0, G1 = np.meshgrid(g0, g1, indexing='ij')
    f = 0.3 + 0.1 * np.tanh((G1 - 150.0) / 40.0) + 0.02 * np.sin(G0 / 90.0)
    keep = ~((np.arange(side)[:, None] % 2 == 1) & (np.arange(side)[None, :] % 2 == 1))
    x = np.transpose([G0[keep], G1[keep]])[:n]
    return x, f[keep][:n], (2 * (g0[1] - g0[0]), 2 * (g1[1] - g1[0]))


def main():
    for n, pub in PUBLISHED.items():
        x, f, ell = table(n)
        f = f - f.mean()
        sig = 1.5 * np.std(f)
        it = mgpi.Interpolator(mgpi.CombinedKernel(mgpi.SquaredExponentialKernel(sig, *ell), mgpi.WhiteNoiseKernel(sig / 1000)))
        it.loglikelihood(x, f)
        ts = []
        for _ in range(5):
            t0 = time.perf_counter()
            ll = it.loglikelihood(x, f)
            ts.append(time.perf_counter() - t0)
        print('N=%6d  loglikelihood %.6f s (best of 5; logL %.6e)   published %s' %
              (len(x), min(ts), ll, ('%.4g s' % pub) if pub else 'not run'), flush=True)


if __name__ == '__main__':
    main()
