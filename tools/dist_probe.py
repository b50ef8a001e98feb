#!/usr/bin/env python3
"""developer probe: the multi-GPU driver on a 1x1 grid vs the single-GPU path (driver overhead)"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, torch.distributed as dist
os.environ.setdefault('MASTER_ADDR', '127.0.0.1'); os.environ.setdefault('MASTER_PORT', '29577')
torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=torch.device('cuda', 0))
import mgpi_amd as mgpi
from mgpi_amd.distributed import DistributedGP, HipTileBackend, ProcessGrid
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import make_inputs
side = tuple(int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '32x32x16').split('x'))
x, f, xs, ell = make_inputs(side, 4096, 3)
k = mgpi.SquaredExponentialKernel(1.0, *ell); nug = mgpi.WhiteNoiseKernel(1e-2)
be = HipTileBackend(torch)
gp = DistributedGP(ProcessGrid(dist, 1, 1), be, x, k.terms(), (k + nug).terms(), nb=512)
for rep in range(2):
    be.fence(); t0 = time.perf_counter(); gp.build(); be.fence(); t1 = time.perf_counter()
    gp.factor(); be.fence(); t2 = time.perf_counter()
    gp.solve(f); be.fence(); t3 = time.perf_counter()
    gp.mean(xs); be.fence(); t4 = time.perf_counter()
    gp.variance(xs); be.fence(); t5 = time.perf_counter()
    n = len(x)
    print('N=%d: build %.1f ms | factor %.1f ms (%.1f TF) | solve %.1f ms | mean %.1f ms | var %.1f ms' %
          (n, (t1 - t0) * 1e3, (t2 - t1) * 1e3, n**3 / 3 / (t2 - t1) / 1e12, (t3 - t2) * 1e3, (t4 - t3) * 1e3, (t5 - t4) * 1e3), flush=True)
dist.destroy_process_group()
