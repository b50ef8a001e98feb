"""Test implementations of the two interfaces the multi-GPU schedule (mgpi_amd/distributed.py) is written against.
TEST INFRASTRUCTURE ONLY - the product never constructs them.

  NumpyTileBackend   every tile operation by numpy/scipy and the oracle's kernel evaluation, buffers = numpy arrays:
                     the block-cyclic schedule (ownership, broadcast order, index bookkeeping) runs at world size
                     2..8 without a GPU
  GlooTransport      the collectives over torch.distributed's gloo backend, on numpy buffers
  HostStagedTransport  GlooTransport under the HIP tile backend: device buffers cross through host copies, so that
                     several ranks can SHARE one GPU (RCCL refuses two ranks on one device) - the rehearsal of the
                     real multi-rank tile distribution on a one-GPU box
"""
import numpy as np
from scipy.linalg import solve_triangular

from oracle import mgpi_oracle as oracle


def _terms(terms):
    return [(int(kind), float(sigma), np.asarray(lengths, dtype=float)) for kind, sigma, lengths in terms]


class NumpyTileBackend(object):
    def __init__(self):
        self._pivot, self._logdiag = 0, 0.0

    # -- streams: on the CPU everything runs in program order
    def side_backend(self):
        # a second handle on the same state: the schedule takes its merged branch only when the look-ahead backend is a
        # different object (two contexts on the GPU); everything runs in program order here either way
        if getattr(self, '_side', None) is None:
            self._side = _SideView(self)
        return self._side

    def wait(self, other):
        pass

    def sync(self):
        pass

    # -- memory
    def alloc(self, rows, cols=1, zero=True):
        a = np.zeros((rows, cols))
        if not zero:
            a.fill(np.nan)         # poison: an element that is read before being written shows up in the results
        return a

    def window(self, buf, r0, rows, c0=0, cols=None):
        cols = buf.shape[1] if cols is None else cols
        v = buf[r0:r0 + rows, c0:c0 + cols]
        assert v.shape == (rows, cols), 'window (%d,%d)+(%d x %d) outside %s' % (r0, c0, rows, cols, buf.shape)
        return v

    def copy(self, dst, src):
        assert dst.shape == src.shape, (dst.shape, src.shape)
        dst[...] = src

    def copy_blocks(self, dst, dst_r0, dst_step, src, src_r0, src_step, rows, count):
        for j in range(count):
            d = dst[dst_r0 + j * dst_step:dst_r0 + j * dst_step + rows]
            s = src[src_r0 + j * src_step:src_r0 + j * src_step + rows]
            assert d.shape == s.shape and d.shape[0] == rows, (d.shape, s.shape, rows)
            d[...] = s

    def fill_diag(self, buf, value):
        n = min(buf.shape)
        buf[np.arange(n), np.arange(n)] = value

    def upload(self, buf, array):
        buf[...] = np.asarray(array, dtype=float).reshape(buf.shape)

    def download(self, buf):
        return np.array(buf, copy=True)

    # -- tile kernels
    def cov(self, x_rows, x_cols, terms, out, n_rows_valid, n_cols_valid):
        out[...] = 0.0
        if n_rows_valid and n_cols_valid:
            out[:n_rows_valid, :n_cols_valid] = oracle.x2cov_chunked(_terms(terms), x_rows[:n_rows_valid], x_cols[:n_cols_valid])

    def points(self, x):
        return np.array(x, dtype=float)

    def cov_dev(self, x_rows, n_rows_valid, x_cols, n_cols_valid, terms, out):
        self.cov(x_rows, x_cols, terms, out, n_rows_valid, n_cols_valid)

    def status_reset(self):
        self._pivot, self._logdiag = 0, 0.0

    def diag(self, block, out):
        out[:, 0] = np.diag(block)

    def potrf_block(self, block, pivot_base, diag0=None):
        """Cholesky with the product's pivot rule: pivot k must exceed 100 eps times the ORIGINAL diagonal entry (diag0)"""
        ref = np.abs(np.diag(block).copy() if diag0 is None else np.asarray(diag0).reshape(-1))
        L = np.tril(block).copy()
        n = L.shape[0]
        for k in range(n):
            piv = L[k, k] - L[k, :k] @ L[k, :k]
            if not piv > 100 * np.finfo(float).eps * ref[k]:
                cand = pivot_base + k + 1
                self._pivot = cand if self._pivot == 0 else min(self._pivot, cand)
                return                          # the block is left as it was; what follows it is undefined, as on the GPU
            L[k, k] = np.sqrt(piv)
            L[k + 1:, k] = (L[k + 1:, k] - L[k + 1:, :k] @ L[k, :k]) / L[k, k]
        block[...] = L
        self._logdiag += float(np.sum(np.log(np.diag(L))))

    def potrf_panel(self, panel, pivot_base, diag0=None):
        w = panel.shape[1]
        self.potrf_block(panel[:w], pivot_base, diag0)
        if panel.shape[0] > w:
            self.trsm_right(panel[:w], panel[w:])

    def status_read(self):
        return self._pivot, self._logdiag

    def trsm_right(self, lkk, rows):
        with np.errstate(all='ignore'):      # after a failed pivot the factor is garbage: keep going like the GPU does
            rows[...] = solve_triangular(np.tril(lkk), rows.T, lower=True, check_finite=False).T

    def gemm_nt_stair(self, c, a, b, P, p, Q, q, bi0, bj0, nb):
        for bj in range(c.shape[1] // nb):
            J = q + Q * (bj0 + bj)
            for bi in range(c.shape[0] // nb):
                I = p + P * (bi0 + bi)
                if I >= J:
                    c[bi * nb:(bi + 1) * nb, bj * nb:(bj + 1) * nb] -= a[bi * nb:(bi + 1) * nb] @ b[bj * nb:(bj + 1) * nb].T

    def gemm_nt_stair_merged(self, c, a, b, P, p, Q, q, bi0, bj0, nb, first_cols):
        self.gemm_nt_stair(c, a, b, P, p, Q, q, bi0, bj0, nb)      # program order: the first columns are done when this returns
        self.merged_launches = getattr(self, 'merged_launches', 0) + 1
        return first_cols

    def count_reset(self):
        pass

    def wait_count(self, owner, target):
        pass

    def overlaps(self, other):
        return True

    def gemm_nn(self, c, a, b):
        c -= a @ b

    def gemm_nn_neg(self, c, a, b):
        c[...] = -(a @ b)

    def trtri_neg(self, a_local, r0, c0, row_step, col_step, w, batch, out):
        out[:batch * w, :w] = 0.0
        for j in range(batch):
            blk = np.tril(a_local[r0 + j * row_step:r0 + j * row_step + w, c0 + j * col_step:c0 + j * col_step + w])
            out[j * w:(j + 1) * w, :w] = -solve_triangular(blk, np.eye(w), lower=True)

    def trsm_left(self, lkk, rhs):
        rhs[...] = solve_triangular(np.tril(lkk), rhs, lower=True)

    def trsv(self, lkk, vec, trans):
        vec[...] = solve_triangular(np.tril(lkk), vec, lower=True, trans='T' if trans else 'N')

    def gemv(self, a, x, y, trans):
        y -= (a.T if trans else a) @ x

    def axpy(self, a, x, y):
        y += a * x

    def colsq(self, v, out):
        out[:, 0] = (v * v).sum(axis=0)[:out.shape[0]]

    def predict_mean(self, xs, x, terms, alpha):
        return oracle.x2cov_chunked(_terms(terms), xs, x) @ alpha

    def prior_diag(self, xs, terms):
        return oracle.paired_cov(_terms(terms), xs, xs)


class _SideView(object):
    """the look-ahead 'context' of the numpy backend: the same arrays, the same status words, another identity"""

    def __init__(self, main):
        object.__setattr__(self, '_main', main)

    def __getattr__(self, name):
        return getattr(object.__getattribute__(self, '_main'), name)

    def __setattr__(self, name, value):
        setattr(object.__getattribute__(self, '_main'), name, value)


class GlooGroup(object):
    """a torch.distributed process group; member index <-> global rank through `ranks`.  Buffers are numpy arrays."""

    def __init__(self, torch, dist, group, ranks):
        self.torch, self.dist, self.group, self.ranks = torch, dist, group, list(ranks)
        self.size = len(self.ranks)

    def _t(self, buf):
        assert buf.flags['C_CONTIGUOUS'], 'a communication buffer must be contiguous'
        return self.torch.from_numpy(buf)

    def bcast(self, be, buf, root):
        self.dist.broadcast(self._t(buf), src=self.ranks[root], group=self.group)

    def reduce_sum(self, be, buf, root):
        # gloo's reduce leaves garbage in the non-root buffers; the schedule does not read them afterwards
        self.dist.reduce(self._t(buf), dst=self.ranks[root], op=self.dist.ReduceOp.SUM, group=self.group)

    def allreduce(self, be, buf, op='sum'):
        red = self.dist.ReduceOp.SUM if op == 'sum' else self.dist.ReduceOp.MAX
        self.dist.all_reduce(self._t(buf), op=red, group=self.group)

    def allgather(self, be, send, recv):
        parts = [self.torch.empty(send.shape, dtype=self.torch.float64) for _ in range(self.size)]
        self.dist.all_gather(parts, self._t(np.ascontiguousarray(send)), group=self.group)
        recv[...] = np.concatenate([p.numpy() for p in parts]).reshape(recv.shape)

    def barrier(self, be):
        self.dist.barrier(group=self.group)


class GlooTransport(object):
    def __init__(self, torch, dist):
        self.torch, self.dist = torch, dist
        self.world, self.rank = dist.get_world_size(), dist.get_rank()

    def world_group(self):
        return GlooGroup(self.torch, self.dist, None, range(self.world))

    def make_groups(self, rank_lists):
        mine = None
        for ranks in rank_lists:           # every rank creates every group, in the same order
            grp = self.dist.new_group(list(ranks))
            if self.rank in ranks:
                mine = GlooGroup(self.torch, self.dist, grp, ranks)
        assert mine is not None
        return mine

    def close(self):
        pass


class _StagedGroup(object):
    """device buffer -> host copy -> gloo collective -> device buffer"""

    def __init__(self, inner):
        self.inner, self.size = inner, inner.size

    def _staged(self, be, buf, fn):
        host = be.download(buf)            # waits for the producing kernels
        fn(host)
        be.upload(buf, host)

    def bcast(self, be, buf, root):
        self._staged(be, buf, lambda h: self.inner.bcast(be, h, root))

    def reduce_sum(self, be, buf, root):
        self._staged(be, buf, lambda h: self.inner.reduce_sum(be, h, root))

    def allreduce(self, be, buf, op='sum'):
        self._staged(be, buf, lambda h: self.inner.allreduce(be, h, op))

    def allgather(self, be, send, recv):
        out = np.empty((recv.rows, recv.cols))
        self.inner.allgather(be, be.download(send), out)
        be.upload(recv, out)

    def barrier(self, be):
        be.sync()
        self.inner.barrier(be)


class HostStagedTransport(object):
    def __init__(self, inner):
        self.inner = inner
        self.world, self.rank = inner.world, inner.rank

    def world_group(self):
        return _StagedGroup(self.inner.world_group())

    def make_groups(self, rank_lists):
        return _StagedGroup(self.inner.make_groups(rank_lists))

    def close(self):
        pass
