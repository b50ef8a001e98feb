"""host-side cost of the block-cyclic driver's factor() on a 1 x 1 grid (cProfile; developer probe)"""
import cProfile, os, pstats, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
import mgpi_amd as mgpi
from mgpi_amd.distributed import DistributedGP, HipTileBackend, LocalTransport, ProcessGrid

x, f, xs, ell = bench.make_inputs((32, 32, 40), 4096, 3)
ctx = mgpi.default_context()
be = HipTileBackend(ctx)
k = mgpi.SquaredExponentialKernel(1.0, *ell)
nug = mgpi.WhiteNoiseKernel(1e-2)
gp = DistributedGP(ProcessGrid(LocalTransport(), 1, 1), be, x, k.terms(), (k + nug).terms(), nb=512)
for rep in range(2):
    gp.build(); gp.factor(source_f=f); be.sync()
gp.build(); be.sync()
pr = cProfile.Profile()
t0 = time.perf_counter()
pr.enable()
gp.factor(source_f=f)
pr.disable()
print('factor %.1f ms wall' % ((time.perf_counter() - t0) * 1e3))
st = pstats.Stats(pr)
st.sort_stats('cumulative').print_stats(25)
