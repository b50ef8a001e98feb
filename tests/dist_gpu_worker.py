"""One rank of the multi-rank tests of the HIP tile backend (tests/test_gpu_distributed.py).

argv: rank world port mode P Q n m nb d lookahead
  mode 'staged': several ranks share GPU 0; gloo is the transport and the collectives are host-staged
                 (dist_cpu_backend.HostStagedTransport); everything else - tile ownership, local tile arrays, the
                 staircase gemm's grid parameters, look-ahead on the side stream, row-panel reuse in the variance sweep -
                 is the code a real multi-GPU launch runs.  torch is imported HERE (test transport only), before the
                 library, so that both bind one HIP runtime.
  mode 'rccl':   one rank per GPU (rank r on device r), RcclTransport: RCCL communicators and collectives through the
                 C-ABI, no torch in the process.
Each rank checks the replicated results against the oracle."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def check_collectives(grid, be):
    """each collective of the transport on its own, with rank-dependent data"""
    g, w, r = grid, grid.world, grid.rank
    buf = be.alloc(256, 128)
    be.upload(buf, np.full((256, 128), float(r)))
    g.all.bcast(be, buf, w - 1)
    assert np.all(be.download(buf) == w - 1), 'bcast'
    be.upload(buf, np.full((256, 128), float(r + 1)))
    g.all.allreduce(be, buf, 'sum')
    assert np.all(be.download(buf) == w * (w + 1) / 2), 'allreduce sum'
    v = be.alloc(512, 1)
    be.upload(v, np.arange(512) * (r + 1.0))
    g.all.allreduce(be, v, 'max')
    assert np.all(be.download(v).reshape(-1) == np.arange(512) * float(w)), 'allreduce max'
    be.upload(v, np.full(512, r + 1.0))
    g.all.reduce_sum(be, v, 0)
    if r == 0:
        assert np.all(be.download(v) == w * (w + 1) / 2), 'reduce'
    mine, every = be.alloc(128, 1), be.alloc(128 * w, 1)
    be.upload(mine, np.full(128, float(r)))
    g.all.allgather(be, mine, every)
    assert np.array_equal(be.download(every).reshape(w, 128)[:, 0], np.arange(w, dtype=float)), 'allgather'
    # row / column communicators: member index = grid coordinate
    be.upload(v, np.full(512, float(g.q)))
    g.row.bcast(be, be.window(v, 128, 256), g.Q - 1)
    got = be.download(v).reshape(-1)
    assert np.all(got[128:384] == g.Q - 1) and np.all(got[:128] == g.q) and np.all(got[384:] == g.q), 'row bcast of a window'
    be.upload(v, np.full(512, float(g.p)))
    g.col.allreduce(be, v, 'sum')
    assert np.all(be.download(v) == g.P * (g.P - 1) / 2), 'column allreduce'
    g.barrier(be)


def main():
    rank, world, port = [int(v) for v in sys.argv[1:4]]
    mode = sys.argv[4]
    P, Q, n, m, nb, d, lookahead = [int(v) for v in sys.argv[5:12]]
    if mode == 'staged':
        os.environ['MASTER_ADDR'] = '127.0.0.1'
        os.environ['MASTER_PORT'] = str(port)
        import torch
        import torch.distributed as dist
        dist.init_process_group('gloo', rank=rank, world_size=world)
    import mgpi_amd as mgpi
    from mgpi_amd.distributed import DistributedGP, HipTileBackend, ProcessGrid, RcclTransport
    from mgpi_amd.rendezvous import exchange_unique_id
    from oracle import mgpi_oracle as oracle

    if mode == 'staged':
        from dist_cpu_backend import GlooTransport, HostStagedTransport
        ctx = mgpi.Context(device=0, nb=nb)
        transport = HostStagedTransport(GlooTransport(torch, dist))
        always = False
    else:
        assert 'torch' not in sys.modules
        ctx = mgpi.Context(device=rank, nb=nb)
        uid = exchange_unique_id(rank, world, RcclTransport.new_unique_id, timeout=300)
        transport = RcclTransport(ctx, uid, rank, world)
        always = True                      # issue every collective even inside one-rank groups
    grid = ProcessGrid(transport, P, Q, always_communicate=always)
    be = HipTileBackend(ctx)
    if mode == 'rccl':
        check_collectives(grid, be)

    rng = np.random.default_rng(1000 + n)                    # the same inputs on every rank
    x = rng.random((n, d))
    f = np.sin(3 * x[:, 0]) * np.cos(2 * x[:, -1]) + x.sum(axis=1)
    xs = rng.random((m, d))
    ell = (0.2,) * d
    k = mgpi.SquaredExponentialKernel(1.0, *ell)
    nug = mgpi.WhiteNoiseKernel(1e-2)
    gp = DistributedGP(grid, be, x, k.terms(), (k + nug).terms(), nb=nb)
    gp.build()
    logdet = gp.factor(lookahead=bool(lookahead), source_f=f)     # forward substitution beside the block steps
    alpha = gp.solve(f)
    both = gp.solve(f)                                            # both sweeps after the fact
    assert np.max(np.abs(both - alpha)) <= 1e-11 * np.max(np.abs(alpha)), np.max(np.abs(both - alpha))
    mean = gp.mean(xs)
    var = gp.variance(xs)

    kernel, nugget = [oracle.term_se(1.0, *ell)], [oracle.term_white(1e-2)]
    L = np.linalg.cholesky(oracle.x2cov_chunked(kernel + nugget, x, x))
    ref = oracle.solve_cholesky(L, f)
    assert abs(logdet - 2 * np.sum(np.log(np.diag(L)))) < 1e-10 * abs(logdet), 'logdet'
    assert np.max(np.abs(alpha - ref)) < 1e-9 * np.max(np.abs(ref)), 'alpha'
    want = oracle.x2cov_chunked(kernel, xs, x) @ ref
    assert np.max(np.abs(mean - want)) < 1e-10 * np.max(np.abs(want)), 'mean'
    assert np.max(np.abs(var - oracle.diag_variance_cholesky(kernel, L, xs, x))) < 1e-9, 'variance'

    # a non-positive-definite matrix is reported by every rank with the same pivot
    bad = DistributedGP(grid, be, np.vstack([x[:n - 1], x[:1]]), k.terms(), (k + nug).terms(), nb=nb)
    bad.build()
    try:
        bad.factor(lookahead=bool(lookahead))
        raise SystemExit('duplicate rows: no error')
    except np.linalg.LinAlgError as err:
        assert err.pivot == n, (err.pivot, n)
    grid.barrier(be)
    transport.close()
    if mode == 'staged':
        dist.destroy_process_group()
    print('RANK-%d-OK' % rank, flush=True)


if __name__ == '__main__':
    main()
