"""where the block-cyclic driver's build() spends its time at C4 size on a 1 x 1 grid, inside full steps (developer probe)"""
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


def timed(name, fn):
    be.sync(); t0 = time.perf_counter()
    out = fn()
    t1 = time.perf_counter(); be.sync(); t2 = time.perf_counter()
    print('  %-10s enqueue %8.1f ms, drained after %8.1f ms' % (name, (t1 - t0) * 1e3, (t2 - t0) * 1e3), flush=True)
    return out


for rep in range(3):
    print('step', rep, flush=True)
    t0 = time.perf_counter()
    # build, piece by piece
    def build_points():
        gi = gp._global_index(gp.row_blocks)
        xr = np.zeros((gp.mloc, gp.d)); xr[:gp.n] = gp.x[gi[:gp.n]]
        return be.points(xr)
    timed('points', build_points)
    timed('build', gp.build)
    timed('factor', lambda: gp.factor(source_f=f))
    timed('solve', lambda: gp.solve(f))
    timed('mean', lambda: gp.mean(xs))
    timed('variance', lambda: gp.variance(xs))
    be.sync()
    print('  step total %.1f ms' % ((time.perf_counter() - t0) * 1e3), flush=True)
for prof in (False, True, False, True):
    if prof:
        ctx.set_profile(True, slots=(0, 5))
    else:
        ctx.set_profile(False)
    be.sync(); t0 = time.perf_counter()
    gp.build(); gp.factor(source_f=f); gp.solve(f); gp.mean(xs); gp.variance(xs)
    be.sync()
    print('un-phased step %.1f ms (event pairs around the trailing updates: %s)' % ((time.perf_counter() - t0) * 1e3, prof), flush=True)
