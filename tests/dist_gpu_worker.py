"""One rank of the multi-rank HIP-backend rehearsal (tests/test_gpu_distributed.py): several ranks share GPU 0,
gloo is the transport and the collectives are host-staged (ProcessGrid(host_staged=True)); everything else - tile
ownership, local tile arrays, the staircase gemm's grid parameters, look-ahead on the side stream, row-panel reuse
in the variance sweep - is the code a real multi-GPU launch runs.  Each rank checks the replicated results against
the oracle.  argv: rank world port P Q n m nb d lookahead"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)


def main():
    rank, world, port, P, Q, n, m, nb, d, lookahead = [int(v) for v in sys.argv[1:11]]
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    import torch
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import mgpi_amd as mgpi
    from mgpi_amd.distributed import DistributedGP, HipTileBackend, ProcessGrid
    from oracle import mgpi_oracle as oracle

    rng = np.random.default_rng(1000 + n)                    # the same inputs on every rank
    x = rng.random((n, d))
    f = np.sin(3 * x[:, 0]) * np.cos(2 * x[:, -1]) + x.sum(axis=1)
    xs = rng.random((m, d))
    ell = (0.2,) * d
    k = mgpi.SquaredExponentialKernel(1.0, *ell)
    nug = mgpi.WhiteNoiseKernel(1e-2)
    ctx = mgpi.Context(device=0, nb=nb)
    gp = DistributedGP(ProcessGrid(dist, P, Q, host_staged=True), HipTileBackend(torch, ctx), x, k.terms(), (k + nug).terms(), nb=nb)
    gp.build()
    logdet = gp.factor(lookahead=bool(lookahead))
    alpha = gp.solve(f)
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
    bad = DistributedGP(ProcessGrid(dist, P, Q, host_staged=True), HipTileBackend(torch, ctx), np.vstack([x[:n - 1], x[:1]]),
                        k.terms(), (k + nug).terms(), nb=nb)
    bad.build()
    try:
        bad.factor(lookahead=bool(lookahead))
        raise SystemExit('duplicate rows: no error')
    except np.linalg.LinAlgError as err:
        assert err.pivot == n, (err.pivot, n)
    dist.barrier()
    dist.destroy_process_group()
    print('RANK-%d-OK' % rank, flush=True)


if __name__ == '__main__':
    main()
