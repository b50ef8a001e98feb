"""a few factorisations through the block-cyclic driver on a 1 x 1 grid (for rocprofv3 --kernel-trace)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import mgpi_amd as mgpi
from mgpi_amd.distributed import DistributedGP, HipTileBackend, LocalTransport, ProcessGrid
x, f, xs, ell = bench.make_inputs((32, 32, 40), 4096, 3)
be = HipTileBackend(mgpi.default_context())
k = mgpi.SquaredExponentialKernel(1.0, *ell)
nug = mgpi.WhiteNoiseKernel(1e-2)
gp = DistributedGP(ProcessGrid(LocalTransport(), 1, 1), be, x, k.terms(), (k + nug).terms(), nb=512)
for rep in range(3):
    gp.build(); be.sync(); t0 = time.perf_counter()
    gp.factor(source_f=f); be.sync()
    print('factor %.1f ms' % ((time.perf_counter() - t0) * 1e3), flush=True)
