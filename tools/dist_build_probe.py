"""where the block-cyclic driver's build() spends its time at C4 size on a 1 x 1 grid (developer probe)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import mgpi_amd as mgpi
from mgpi_amd.distributed import DistributedGP, HipTileBackend, LocalTransport, ProcessGrid

x, f, xs, ell = bench.make_inputs((16, 16, 16, 16), 4096, 4)
print('N', len(x), 'D', x.shape[1], flush=True)
ctx = mgpi.default_context()
be = HipTileBackend(ctx)
k = mgpi.SquaredExponentialKernel(1.0, *ell)
nug = mgpi.WhiteNoiseKernel(1e-2)
gp = DistributedGP(ProcessGrid(LocalTransport(), 1, 1), be, x, k.terms(), (k + nug).terms(), nb=512)
for rep in range(3):
    be.sync(); t0 = time.perf_counter()
    gp.build()
    t1 = time.perf_counter()
    be.sync(); t2 = time.perf_counter()
    print('build: enqueue %.1f ms, drained after %.1f ms' % ((t1 - t0) * 1e3, (t2 - t0) * 1e3), flush=True)
    XR = be.points(np.zeros((gp.mloc, gp.d))); be.sync()
    t0 = time.perf_counter()
    for lj in range(len(gp.col_blocks)):
        rows = gp.mloc - lj * 512
        be.cov_dev(be.window(XR, lj * 512, rows), rows, be.window(XR, lj * 512, 512), 512, gp.solve_terms, be.window(gp.A, lj * 512, rows, lj * 512, 512))
    t1 = time.perf_counter(); be.sync(); t2 = time.perf_counter()
    print('  cov launches alone: enqueue %.1f ms, drained %.1f ms' % ((t1 - t0) * 1e3, (t2 - t0) * 1e3), flush=True)
    t0 = time.perf_counter()
    for kk, j in gp._diag_slot.items():
        be.diag(be.window(gp.A, kk * 512, 512, kk * 512, 512), be.window(gp.diag0, j * 512, 512))
    t1 = time.perf_counter(); be.sync(); t2 = time.perf_counter()
    print('  diag copies: enqueue %.1f ms, drained %.1f ms' % ((t1 - t0) * 1e3, (t2 - t0) * 1e3), flush=True)
