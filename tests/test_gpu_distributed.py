"""The multi-GPU driver with the HIP tile backend on ONE GPU (1 x 1 grid): every tile operation goes through
the wrapped-view C-ABI entry points the N>1 path uses (mgpi_mat_wrap, mgpi_cov_block_dev, mgpi_trsm_right_dev,
mgpi_gemm_nt/nn_dev, mgpi_trsv_dev, mgpi_gemv_dev); only the collectives are no-ops.  The communication
schedule itself is covered by tests/test_distributed_cpu.py (gloo, world size 2 and 4)."""
import os
import socket

import numpy as np
import pytest

import mgpi_amd as mgpi
from conftest import relmax
from oracle import mgpi_oracle as oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def dist1():
    import torch
    import torch.distributed as dist
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=0, world_size=1)
    yield torch, dist
    dist.destroy_process_group()


@pytest.mark.parametrize('n,m,nb', [(1500, 200, 512), (700, 130, 128), (2048, 64, 512)])
def test_hip_tile_backend_single_rank(dist1, n, m, nb):
    torch, dist = dist1
    from mgpi_amd.distributed import DistributedGP, HipTileBackend, ProcessGrid
    rng = np.random.default_rng(n)
    x = rng.random((n, 3))
    f = np.sin(3 * x[:, 0]) * np.cos(2 * x[:, 1]) + x[:, 2]
    xs = rng.random((m, 3))
    ell = (0.15, 0.15, 0.15)
    k = mgpi.SquaredExponentialKernel(1.0, *ell)
    nug = mgpi.WhiteNoiseKernel(1e-2)
    ctx = mgpi.Context(device=0, nb=nb)
    gp = DistributedGP(ProcessGrid(dist, 1, 1), HipTileBackend(torch, ctx), x, k.terms(), (k + nug).terms(), nb=nb)
    gp.build()
    logdet = gp.factor()
    alpha = gp.solve(f)
    mean = gp.mean(xs)
    var = gp.variance(xs)

    kernel, nugget = [oracle.term_se(1.0, *ell)], [oracle.term_white(1e-2)]
    K = oracle.x2cov_chunked(kernel + nugget, x, x)
    L = np.linalg.cholesky(K)
    ref_alpha = oracle.solve_cholesky(L, f)
    assert abs(logdet - 2 * np.sum(np.log(np.diag(L)))) < 1e-10 * abs(logdet)
    assert relmax(alpha, ref_alpha) < 1e-9
    assert relmax(mean, oracle.x2cov_chunked(kernel, xs, x) @ ref_alpha) < 1e-10
    assert np.max(np.abs(var - oracle.diag_variance_cholesky(kernel, L, xs, x))) < 1e-9
    # and it agrees with the single-GPU path of the package
    it = mgpi.Interpolator(k, nugget=nug)
    m1, v1 = it.condition_diag(xs, x, f)
    assert relmax(mean, m1) < 1e-11 and np.max(np.abs(var - v1)) < 1e-11


def test_nccl_collectives_on_device_buffers():
    """RCCL path on one rank: the driver's collectives (broadcast / reduce inside row and column groups, world
    broadcast) are issued for real on device buffers through the nccl backend, on the stream shared with the
    tile kernels.  Run in a subprocess because a process holds one default process group."""
    import subprocess
    import sys
    code = r'''
import os, sys, numpy as np
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import torch, torch.distributed as dist
os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = sys.argv[1]
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
import mgpi_amd as mgpi
from mgpi_amd.distributed import DistributedGP, HipTileBackend, ProcessGrid
from oracle import mgpi_oracle as oracle
rng = np.random.default_rng(0)
n, m = 1100, 70
x = rng.random((n, 2)); f = np.sin(4 * x[:, 0]) + x[:, 1]; xs = rng.random((m, 2))
k = mgpi.SquaredExponentialKernel(1.0, 0.1, 0.1); nug = mgpi.WhiteNoiseKernel(1e-2)
ctx = mgpi.Context(device=0, nb=256)
gp = DistributedGP(ProcessGrid(dist, 1, 1, always_communicate=True), HipTileBackend(torch, ctx), x, k.terms(), (k + nug).terms(), nb=256)
gp.build(); gp.factor(); alpha = gp.solve(f); mean = gp.mean(xs); var = gp.variance(xs)
K = oracle.x2cov_chunked([oracle.term_se(1.0, 0.1, 0.1), oracle.term_white(1e-2)], x, x)
L = np.linalg.cholesky(K); ref = oracle.solve_cholesky(L, f)
assert np.max(np.abs(alpha - ref)) < 1e-9 * np.max(np.abs(ref))
assert np.max(np.abs(mean - oracle.x2cov_chunked([oracle.term_se(1.0, 0.1, 0.1)], xs, x) @ ref)) < 1e-10
dist.destroy_process_group()
print("NCCL-OK")
'''
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    out = subprocess.run([sys.executable, '-c', code, str(port)], cwd=os.path.dirname(os.path.dirname(os.path.abspath(__file__))),
                         capture_output=True, text=True, timeout=300)
    assert 'NCCL-OK' in out.stdout, out.stdout[-2000:] + out.stderr[-4000:]


@pytest.mark.parametrize('P,Q,n,m,nb,d,lookahead', [(1, 2, 1500, 130, 256, 3, 1), (2, 1, 1100, 70, 128, 2, 1),
                                                     (2, 2, 2100, 300, 256, 3, 1), (2, 2, 1024, 64, 128, 4, 0)])
def test_hip_tile_backend_multi_rank_on_one_gpu(P, Q, n, m, nb, d, lookahead):
    """P*Q ranks share GPU 0 (gloo, host-staged collectives - tests/dist_gpu_worker.py): the multi-rank tile
    distribution with the HIP kernels, which a one-GPU box cannot otherwise run.  At most 4 ranks + this process
    touch the card (the box allows 6)."""
    import subprocess
    import sys
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    world = P * Q
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'dist_gpu_worker.py')
    procs = [subprocess.Popen([sys.executable, worker] + [str(v) for v in (r, world, port, P, Q, n, m, nb, d, lookahead)],
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    try:
        for pr in procs:
            outs.append(pr.communicate(timeout=300)[0])
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
    for r, (pr, out) in enumerate(zip(procs, outs)):
        assert pr.returncode == 0 and 'RANK-%d-OK' % r in out, 'rank %d:\n%s' % (r, out[-3000:])
