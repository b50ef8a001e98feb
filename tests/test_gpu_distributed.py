"""The multi-GPU driver with the HIP tile backend, as far as ONE GPU can take it (run with -m gpu):

  * 1 x 1 grid, LocalTransport: every tile operation goes through the windowed C-ABI entry points the N>1 path uses
    (mgpi_mat_view, mgpi_cov_block_dev, mgpi_potrf_enqueue_dev, mgpi_trsm_right_dev, mgpi_gemm_nt_stair/nn_dev,
    mgpi_copy_blocks_dev, mgpi_trsv_dev, mgpi_gemv_dev, mgpi_axpy_dev, mgpi_colsq_dev);
  * RCCL through the C-ABI (mgpi_comm_*): communicator creation, row / column splits and every collective of the
    schedule on a one-rank world - and on a two-rank world when the box has two GPUs;
  * P*Q ranks sharing GPU 0 with host-staged gloo collectives (tests/dist_gpu_worker.py): the real multi-rank tile
    distribution with the HIP kernels.
The communication schedule itself is also covered on the CPU by tests/test_distributed_cpu.py (gloo, world size 2..8).
"""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

import mgpi_amd as mgpi
from conftest import relmax
from oracle import mgpi_oracle as oracle

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


@pytest.mark.parametrize('n,m,nb,lookahead', [(1500, 200, 512, True), (700, 130, 128, True), (2048, 64, 512, False)])
def test_hip_tile_backend_single_rank(n, m, nb, lookahead):
    from mgpi_amd.distributed import DistributedGP, HipTileBackend, LocalTransport, ProcessGrid
    rng = np.random.default_rng(n)
    x = rng.random((n, 3))
    f = np.sin(3 * x[:, 0]) * np.cos(2 * x[:, 1]) + x[:, 2]
    xs = rng.random((m, 3))
    ell = (0.15, 0.15, 0.15)
    k = mgpi.SquaredExponentialKernel(1.0, *ell)
    nug = mgpi.WhiteNoiseKernel(1e-2)
    ctx = mgpi.Context(device=0, nb=nb)
    gp = DistributedGP(ProcessGrid(LocalTransport(), 1, 1), HipTileBackend(ctx), x, k.terms(), (k + nug).terms(), nb=nb)
    gp.build()
    logdet = gp.factor(lookahead=lookahead, source_f=f)
    alpha = gp.solve(f)
    assert relmax(gp.solve(f), alpha) < 1e-11               # both sweeps after the fact
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
    # a singular matrix: the failing pivot travels through the device-resident status
    bad = DistributedGP(ProcessGrid(LocalTransport(), 1, 1), HipTileBackend(ctx), np.vstack([x[:n - 1], x[:1]]), k.terms(),
                        k.terms(), nb=nb)
    bad.build()
    with pytest.raises(np.linalg.LinAlgError) as err:
        bad.factor(lookahead=lookahead)
    assert err.value.pivot == n


def _run_ranks(world, args, timeout=600, extra_env=None):
    """start `world` ranks of tests/dist_gpu_worker.py and return their outputs"""
    s = socket.socket()
    s.bind(('127.0.0.1', 0))
    port = s.getsockname()[1]
    s.close()
    worker = os.path.join(HERE, 'dist_gpu_worker.py')
    rdzv = os.path.join('/tmp', 'mgpi_test_rdzv_%d_%d' % (os.getpid(), port))
    env = dict(os.environ, MGPI_RENDEZVOUS=rdzv, **(extra_env or {}))
    procs = [subprocess.Popen([sys.executable, worker] + [str(v) for v in (r, world, port) + tuple(args)], env=env,
                              stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    outs = []
    try:
        for pr in procs:
            outs.append(pr.communicate(timeout=timeout)[0])
    finally:
        for pr in procs:
            if pr.poll() is None:
                pr.kill()
        if os.path.exists(rdzv):
            os.remove(rdzv)
    for r, (pr, out) in enumerate(zip(procs, outs)):
        assert pr.returncode == 0 and 'RANK-%d-OK' % r in out, 'rank %d:\n%s' % (r, out[-3000:])


def test_rccl_collectives_through_the_c_abi():
    """mgpi_comm_create / split / bcast / reduce_sum / allreduce / allgather / barrier on device buffers, on the ctx
    stream: the whole GP on a one-rank world with every collective issued, plus direct checks of each collective; on a
    box with two or more GPUs the same on a 1 x 2 grid with one rank per GPU (RCCL over xGMI)."""
    _run_ranks(1, ('rccl', 1, 1, 1100, 70, 256, 2, 1))
    from mgpi_amd import _lib
    if _lib.device_count() >= 2:
        _run_ranks(2, ('rccl', 1, 2, 2300, 200, 256, 3, 1))
        _run_ranks(2, ('rccl', 2, 1, 1500, 100, 128, 2, 0))


@pytest.mark.parametrize('P,Q,n,m,nb,d,lookahead', [(1, 2, 1500, 130, 256, 3, 1), (2, 1, 1100, 70, 128, 2, 1),
                                                     (2, 2, 2100, 300, 256, 3, 1), (2, 2, 1024, 64, 128, 4, 0)])
def test_hip_tile_backend_multi_rank_on_one_gpu(P, Q, n, m, nb, d, lookahead):
    """P*Q ranks share GPU 0 (gloo, host-staged collectives - tests/dist_gpu_worker.py): the multi-rank tile
    distribution with the HIP kernels, which a one-GPU box cannot otherwise run.  At most 4 ranks + this process
    touch the card (the box allows 6).  Once with single block columns per trailing update, once with super-panels of two
    (MGPI_DIST_SUPER_MIN=0: contraction 2 nb, slabs side by side) and the merged staircase update wherever it can apply
    (MGPI_DIST_MERGED_MIN=0: the next super-panel's block columns first, counted; the look-ahead context starts on the count) -
    the schedule the large runs use."""
    _run_ranks(P * Q, ('staged', P, Q, n, m, nb, d, lookahead))
    _run_ranks(P * Q, ('staged', P, Q, n, m, nb, d, lookahead), extra_env={'MGPI_DIST_SUPER_MIN': '0', 'MGPI_DIST_MERGED_MIN': '0'})


def test_staged_2x2_at_c2_size():
    """the 2 x 2 grid at N = 8192 (BASELINE.json configs[1]'s size), nb = 512, look-ahead, super-panels: four ranks on GPU 0
    with host-staged collectives, every rank against the oracle [r3: the staged tests stopped at N = 2300]"""
    _run_ranks(4, ('staged', 2, 2, 8192, 256, 512, 3, 1), timeout=900, extra_env={'MGPI_DIST_SUPER_MIN': '2048', 'MGPI_DIST_MERGED_MIN': '1024'})


def _single_rank_gp(x, ell, nb=512, **kw):
    from mgpi_amd.distributed import DistributedGP, HipTileBackend, LocalTransport, ProcessGrid
    k = mgpi.SquaredExponentialKernel(1.0, *ell)
    nug = mgpi.WhiteNoiseKernel(1e-2)
    ctx = mgpi.default_context()
    gp = DistributedGP(ProcessGrid(LocalTransport(), 1, 1), HipTileBackend(ctx), x, k.terms(), (k + nug).terms(), nb=nb, **kw)
    return gp, k, nug


def test_single_rank_driver_at_c3_against_the_single_gpu_path():
    """the block-cyclic driver on a 1 x 1 grid at the C3 recipe (N = 32768, D = 3, nb = 512): alpha, mean and variance against
    the single-GPU path's to 1e-11, logdet to 1e-12 - the same matrix through two schedules (fused tall panels, super-panels of
    two block columns, look-ahead on the side context) [r3: the driver had never run above N = 2300 under a test]"""
    import bench
    x, f, xs, ell, sigma, nugget_sigma = bench.make_workload('c3')
    xq = xs[:1024]
    gp, k, nug = _single_rank_gp(x, ell)
    gp.build()
    logdet = gp.factor(source_f=f)                           # as bench.py runs it: the forward sweep beside the factorisation
    alpha = gp.solve(f)
    mean = gp.mean(xq)
    var = gp.variance(xq, m_slab=512)                        # two slabs
    del gp
    it = mgpi.Interpolator(k, nugget=nug)
    fit = it.fit(x, f)
    a1 = fit.alpha.to_host().reshape(-1)
    assert abs(logdet - fit.logdet) < 1e-12 * abs(fit.logdet)
    assert relmax(alpha, a1) < 1e-11
    assert relmax(mean, fit.mean(xq)) < 1e-11
    assert np.max(np.abs(var - fit.variance(xq))) < 1e-11


def test_single_rank_driver_at_c4_size():
    """N = 65536, D = 4 (BASELINE.json configs[3]'s matrix) through the driver on one GPU: residual of alpha < 1e-10 via the
    fused contraction with kernel + nugget, variance bounds, logdet against the single-GPU path"""
    import bench
    x, f, xs, ell, sigma, nugget_sigma = bench.make_workload('c4')
    gp, k, nug = _single_rank_gp(x, ell)
    gp.build()
    logdet = gp.factor()
    alpha = gp.solve(f)
    q = xs[:512].copy()
    q[:8] = x[:8]
    var = gp.variance(q)
    del gp
    idx = np.linspace(0, len(x) - 1, 2048).astype(int)
    Kalpha = mgpi.Interpolator(k + nug).predict(x[idx], x, alpha)
    assert np.max(np.abs(Kalpha - f[idx])) / np.max(np.abs(f)) < 1e-10
    assert np.all(var > -1e-9) and np.all(var <= 1.0 + 1e-12) and np.all(var[:8] < 1e-4 * 1.1 + 1e-9)
    fit = mgpi.Interpolator(k, nugget=nug).fit(x)
    assert abs(logdet - fit.logdet) < 1e-11 * abs(fit.logdet)
