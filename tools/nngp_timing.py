#!/usr/bin/env python3
"""developer probe: NearestNeighborInterpolator.loglikelihood wall-clock at the training-set sizes of the reference's
published NNGP timings (test/uiuc/uiuc-test-nngp-timing.txt, produced with `--num-neighbors 20 --order-by-index 0`
per test/uiuc/README.md:37-45: k = 20 neighbours, ordered by coordinate 0, SE(D=2) + WhiteNoise sigma/1000; argv[1]
overrides k; columns "NNGP" = sort + neighbour search + likelihood, "(presorted)NNGP" = likelihood only).
Synthetic checkerboard-thinned (muB, T) grid of the same sizes; host arrays in, scalar out."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import mgpi_amd as mgpi
from loglike_timing import table

PUBLISHED = {16: (0.0031, 0.0027), 338: (0.12, 0.08), 1311: (0.55, 0.40), 5160: (2.87, 1.62), 7933: (5.59, 2.55),
             14074: (15.13, 4.62), 31527: (68.59, 10.36), 250000: None, 1000000: None}


def main():
    k = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    print('num_neighbors = %d, order_by_index = 0' % k, flush=True)
    for n, pub in PUBLISHED.items():
        x, f, ell = table(n)
        f = f - f.mean()
        sig = 1.5 * np.std(f)
        kern = mgpi.CombinedKernel(mgpi.SquaredExponentialKernel(sig, *ell), mgpi.WhiteNoiseKernel(sig / 1000))
        nn = mgpi.NearestNeighborInterpolator(kern, num_neighbors=k, order_by_index=0)
        nn.loglikelihood(x, f)
        full, pre, search = [], [], []
        for _ in range(3):
            t0 = time.perf_counter()
            ll = nn.loglikelihood(x, f)
            full.append(time.perf_counter() - t0)
        sx, sf = nn._2sorted(x, source_f=f)
        for _ in range(3):
            t0 = time.perf_counter()
            nb = nn._2neighbors(sx)
            search.append(time.perf_counter() - t0)
            t0 = time.perf_counter()
            nn.loglikelihood(sx, sf, neighbors=nb)
            pre.append(time.perf_counter() - t0)
        pairs = len(x) * (len(x) - 1) / 2
        print('N=%8d  loglikelihood %.5f s | presorted %.5f s | neighbour search %.5f s (= %.2f Gpair/s of the N(N-1)/2 scan of the reference)   logL %.6e   published %s'
              % (len(x), min(full), min(pre), min(search), pairs / min(search) / 1e9, ll,
                 ('%.4g s | %.4g s' % pub) if pub else 'not run'), flush=True)


if __name__ == '__main__':
    main()
