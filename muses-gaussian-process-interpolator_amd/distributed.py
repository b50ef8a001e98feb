"""Multi-GPU dense GP: 2-D block-cyclic right-looking Cholesky over the GPUs of one node.

One process per GPU.  Everything that touches device memory - tile kernels, copies, the collectives - goes through
libmgpi_hip.so: the communicators are RCCL communicators created by ``mgpi_comm_create`` / ``mgpi_comm_split``
(csrc/comm.hip; RCCL over xGMI), the buffers are ``mgpi_mat`` handles and views of them, and every call is enqueued on
the stream of an ``mgpi_ctx`` without host synchronisation.  This module only holds the schedule: who owns which block,
what is broadcast to whom in which order.

The schedule is written against two small interfaces so that it can be exercised without GPUs:

  * a *tile backend* (``HipTileBackend`` here; the CPU tests inject a numpy one, tests/dist_cpu_backend.py):
    alloc / window / copy / copy_blocks / upload / download + the tile kernels;
  * a *transport* (``RcclTransport`` here; the CPU tests inject one over gloo): ``make_groups`` and, per group,
    bcast / reduce_sum / allreduce / allgather on the backend's buffers.

The product never constructs the test implementations.

Distribution: K is cut into nb x nb blocks (nb = the outer Cholesky block, 512); block (I, J) lives on grid position
(I mod P, J mod Q) of a P x Q process grid (1x1, 1x2, 2x2, 2x4).  Only a broadcast pattern is needed (no all-reduce on
the data path): per block step k
  1. the owner factors the diagonal block and broadcasts it down its process column,
  2. that column's ranks solve their panel blocks  A[I][k] <- A[I][k] L_kk^-T,
  3. the panel is broadcast along process rows, then (transposed role) along process columns,
  4. every rank updates its local trailing blocks  A[I][J] -= L[I][k] L[J][k]^T  (I >= J > k) on fp64 MFMA.
The failing-pivot word and sum log L_ii stay on the device until the end of the factorisation (no host round trip per
block step).  The solves use a replicated right-hand side: per block step one small reduce inside a process row
(forward) or column (backward) and one world broadcast of the nb solved entries.
"""
import ctypes
import math
import os
import sys

import numpy as np

from . import _lib

__all__ = ['ProcessGrid', 'HipTileBackend', 'RcclTransport', 'LocalTransport', 'DistributedGP', 'choose_grid']


def choose_grid(world):
    """P x Q with P <= Q, as square as possible: 1->1x1, 2->1x2, 4->2x2, 8->2x4"""
    p = int(math.floor(math.sqrt(world)))
    while world % p:
        p -= 1
    return p, world // p


# ---- transport: RCCL through the C-ABI ---------------------------------------------------------------------------

class RcclGroup(object):
    """one RCCL communicator (mgpi_comm); collectives on DeviceMatrix buffers, on the stream of the backend's ctx"""

    def __init__(self, lib, handle, rank, size):
        self.lib, self.handle, self.rank, self.size = lib, handle, rank, size

    @staticmethod
    def _span(buf):
        """(device pointer, number of doubles) of a contiguous buffer: full-width rows of a matrix, or a vector"""
        rows, cols, ld = buf.shape3()
        assert ld == 1 or ld == cols, 'a communication buffer must be contiguous (%d x %d, row stride %d)' % (rows, cols, ld)
        return buf.devptr, rows * ld

    def bcast(self, be, buf, root):
        ptr, count = self._span(buf)
        _lib.check(self.lib.mgpi_comm_bcast(self.handle, be.ctx.handle, ptr, count, int(root)))

    def reduce_sum(self, be, buf, root):
        ptr, count = self._span(buf)
        _lib.check(self.lib.mgpi_comm_reduce_sum(self.handle, be.ctx.handle, ptr, count, int(root)))

    def allreduce(self, be, buf, op='sum'):
        ptr, count = self._span(buf)
        _lib.check(self.lib.mgpi_comm_allreduce(self.handle, be.ctx.handle, ptr, count, 0 if op == 'sum' else 1))

    def allgather(self, be, send, recv):
        sp, count = self._span(send)
        rp, total = self._span(recv)
        assert total == count * self.size
        _lib.check(self.lib.mgpi_comm_allgather(self.handle, be.ctx.handle, sp, rp, count))

    def barrier(self, be):
        _lib.check(self.lib.mgpi_comm_barrier(self.handle, be.ctx.handle))

    def close(self):
        if self.handle is not None:
            self.lib.mgpi_comm_destroy(self.handle)
            self.handle = None


class RcclTransport(object):
    """the world communicator of this process's GPU.  ``unique_id`` is the MGPI_COMM_ID_BYTES blob rank 0 obtained
    from ``RcclTransport.new_unique_id()`` and shipped to the others (mgpi_amd.rendezvous)."""

    def __init__(self, ctx, unique_id, rank, world):
        self.ctx, self.lib = ctx, ctx.lib
        self.rank, self.world = int(rank), int(world)
        h = ctypes.c_void_p()
        blob = ctypes.create_string_buffer(bytes(unique_id), _lib.COMM_ID_BYTES)
        _lib.check(self.lib.mgpi_comm_create(ctx.handle, blob, self.rank, self.world, ctypes.byref(h)))
        self._world = RcclGroup(self.lib, h, self.rank, self.world)
        self._groups = [self._world]

    @staticmethod
    def new_unique_id():
        lib = _lib.load_library()
        blob = ctypes.create_string_buffer(_lib.COMM_ID_BYTES)
        _lib.check(lib.mgpi_comm_unique_id(blob))
        return blob.raw

    def world_group(self):
        return self._world

    def make_groups(self, rank_lists):
        """disjoint rank lists covering the world -> the group this rank belongs to (one ncclCommSplit: color = index of
        the list, key = position in it).  Every rank calls this with the same lists, like dist.new_group."""
        mine = [i for i, ranks in enumerate(rank_lists) if self.rank in ranks]
        assert len(mine) == 1, 'the rank lists must partition the world'
        ranks = list(rank_lists[mine[0]])
        h = ctypes.c_void_p()
        _lib.check(self.lib.mgpi_comm_split(self._world.handle, mine[0], ranks.index(self.rank), ctypes.byref(h)))
        r, s = ctypes.c_int(0), ctypes.c_int(0)
        _lib.check(self.lib.mgpi_comm_info(h, ctypes.byref(r), ctypes.byref(s)))
        assert (r.value, s.value) == (ranks.index(self.rank), len(ranks)), 'RCCL ranked the split differently'
        g = RcclGroup(self.lib, h, r.value, s.value)
        self._groups.append(g)
        return g

    def close(self):
        for g in reversed(self._groups):
            g.close()
        self._groups = []


class _NoGroup(object):
    """the communicator of a one-rank world: nothing to move"""
    rank, size = 0, 1

    def bcast(self, be, buf, root):
        pass

    reduce_sum = bcast

    def allreduce(self, be, buf, op='sum'):
        pass

    def allgather(self, be, send, recv):
        be.copy(recv, send)

    def barrier(self, be):
        be.sync()


class LocalTransport(object):
    """world of one process (the driver on a 1 x 1 grid, without loading RCCL)"""
    rank, world = 0, 1

    def world_group(self):
        return _NoGroup()

    def make_groups(self, rank_lists):
        return _NoGroup()

    def close(self):
        pass


class ProcessGrid(object):
    """rank <-> (p, q) on a P x Q grid (row-major), with one communicator per grid row and per grid column"""

    def __init__(self, transport, P=None, Q=None, always_communicate=False):
        self.t = transport
        self.always = bool(always_communicate)   # tests: issue the collectives even inside size-1 groups
        self.world, self.rank = transport.world, transport.rank
        if P is None:
            P, Q = choose_grid(self.world)
        assert P * Q == self.world, 'grid %dx%d does not match world size %d' % (P, Q, self.world)
        self.P, self.Q = P, Q
        self.p, self.q = divmod(self.rank, Q)
        # member index inside the row group = q, inside the column group = p
        self.row = transport.make_groups([[pp * Q + qq for qq in range(Q)] for pp in range(P)])
        self.col = transport.make_groups([[pp * Q + qq for pp in range(P)] for qq in range(Q)])
        self.all = transport.world_group()

    def rank_of(self, p, q):
        return p * self.Q + q

    def bcast_row(self, be, buf, src_q):
        """broadcast inside this rank's grid row from column src_q"""
        if self.Q > 1 or self.always:
            self.row.bcast(be, buf, src_q)

    def bcast_col(self, be, buf, src_p):
        if self.P > 1 or self.always:
            self.col.bcast(be, buf, src_p)

    def bcast_world(self, be, buf, src_rank):
        if self.world > 1 or self.always:
            self.all.bcast(be, buf, src_rank)

    def reduce_row(self, be, buf, dst_q):
        if self.Q > 1 or self.always:
            self.row.reduce_sum(be, buf, dst_q)

    def reduce_col(self, be, buf, dst_p):
        if self.P > 1 or self.always:
            self.col.reduce_sum(be, buf, dst_p)

    def all_reduce(self, be, buf, op='sum', column_only=False):
        """all-reduce over the world, or inside this rank's process column"""
        (self.col if column_only else self.all).allreduce(be, buf, op)

    def all_gather(self, be, send, recv):
        self.all.allgather(be, send, recv)

    def barrier(self, be):
        self.all.barrier(be)


# ---- tile backend: the HIP kernels of the single-GPU path on windows of device buffers ------------------------------

class HipTileBackend(object):
    """buffers = _lib.DeviceMatrix (and views of them); every operation is a C-ABI call enqueued on ctx's stream"""

    def __init__(self, ctx=None):
        self.ctx = ctx if ctx is not None else _lib.default_context()
        self.lib = self.ctx.lib
        self._side = None
        self._overlap_verdict = {}

    # -- second stream for the look-ahead panel (its own mgpi ctx = its own HIP stream and workspace) -----------
    def side_backend(self):
        if self._side is None:
            self._side = HipTileBackend(_lib.Context(device=self.ctx.device, nb=0))
            # its panels and broadcasts run beside the main context's trailing update: on the high-priority queue
            _lib.check(self.lib.mgpi_ctx_prefer_high_priority(self._side.ctx.handle))
        return self._side

    def wait(self, other):
        """what `other` has queued so far precedes what this backend queues from now on (device-side ordering)"""
        if other is not self:
            _lib.check(self.lib.mgpi_ctx_wait(self.ctx.handle, other.ctx.handle))

    def sync(self):
        self.ctx.sync()
        if self._side is not None:
            self._side.ctx.sync()

    # -- memory -------------------------------------------------------------------------------------------
    def alloc(self, rows, cols=1, zero=True):
        """(rows x cols) device buffer, row stride = cols (cols a multiple of 128, or 1 for a vector); zero-filled"""
        assert cols == 1 or cols % 128 == 0, 'tile buffers are 1 or a multiple of 128 columns wide'
        return _lib.DeviceMatrix(self.ctx, rows, cols, wide=(cols != 1))

    def window(self, buf, r0, rows, c0=0, cols=None):
        return buf.view(int(r0), int(c0), int(rows), int(buf.cols if cols is None else cols))

    def copy(self, dst, src):
        _lib.check(self.lib.mgpi_mat_copy(self.ctx.handle, src.handle, dst.handle))

    def copy_blocks(self, dst, dst_r0, dst_step, src, src_r0, src_step, rows, count):
        _lib.check(self.lib.mgpi_copy_blocks_dev(self.ctx.handle, src.handle, int(src_r0), int(src_step), dst.handle, int(dst_r0),
                                                 int(dst_step), int(rows), int(count)))

    def fill_diag(self, buf, value):
        _lib.check(self.lib.mgpi_mat_fill_diag(self.ctx.handle, buf.handle, float(value)))

    def upload(self, buf, array):
        buf.upload(array)                     # stream-ordered copy on the ctx stream, then a host wait

    def download(self, buf):
        self.sync()
        return buf.to_host()

    # -- tile kernels -------------------------------------------------------------------------------------
    def cov(self, x_rows, x_cols, terms, out, n_rows_valid, n_cols_valid):
        """out[i][j] = k(x_rows[i], x_cols[j]) for i < n_rows_valid, j < n_cols_valid, zero elsewhere"""
        desc, count = _lib.make_terms(terms)
        XR = _lib.DeviceMatrix.from_host(self.ctx, x_rows)
        XC = _lib.DeviceMatrix.from_host(self.ctx, x_cols)
        _lib.check(self.lib.mgpi_cov_block_dev(self.ctx.handle, XR.handle, n_rows_valid, XC.handle, n_cols_valid, desc, count,
                                               out.handle))

    def points(self, x):
        """a (rows x D) coordinate array on the device, windowed by rows afterwards"""
        return _lib.DeviceMatrix.from_host(self.ctx, x)

    def cov_dev(self, XR, n_rows_valid, XC, n_cols_valid, terms, out):
        """`cov` with the coordinates already on the device (windows of `points`)"""
        desc, count = _lib.make_terms(terms)
        _lib.check(self.lib.mgpi_cov_block_dev(self.ctx.handle, XR.handle, int(n_rows_valid), XC.handle, int(n_cols_valid), desc,
                                               count, out.handle))

    def status_reset(self):
        _lib.check(self.lib.mgpi_ctx_status_reset(self.ctx.handle))

    def diag(self, block, out):
        """out[i] = block[i][i]"""
        _lib.check(self.lib.mgpi_mat_diag(self.ctx.handle, block.handle, out.handle))

    def potrf_block(self, block, pivot_base, diag0=None):
        """in-place Cholesky of a diagonal block, no read-back: failing pivot and sum log diag stay on the device.
        diag0: the block's ORIGINAL diagonal (what its pivots are measured against)"""
        st = self.lib.mgpi_potrf_enqueue_dev(self.ctx.handle, block.handle, int(pivot_base), diag0.handle if diag0 is not None else None)
        _lib.check(st)

    def potrf_panel(self, panel, pivot_base, diag0=None):
        """a whole local block column (rows from the diagonal block down): Cholesky of its top square block and the solve of
        the rows below it in the same fused launches, no read-back"""
        _lib.check(self.lib.mgpi_potrf_panel_enqueue_dev(self.ctx.handle, panel.handle, int(pivot_base),
                                                         diag0.handle if diag0 is not None else None))

    def status_read(self):
        """(first failing pivot, 1-based global index, 0 = none; sum of log L_ii) since status_reset.  Synchronises."""
        pivot, logdiag = _lib.c_int64(0), ctypes.c_double(0.0)
        _lib.check(self.lib.mgpi_ctx_status_read(self.ctx.handle, ctypes.byref(pivot), ctypes.byref(logdiag)))
        return int(pivot.value), float(logdiag.value)

    def trsm_right(self, lkk, rows):
        lkk.mark_factored()
        _lib.check(self.lib.mgpi_trsm_right_dev(self.ctx.handle, lkk.handle, rows.handle))

    def gemm_nt_stair(self, c, a, b, P, p, Q, q, bi0, bj0, nb):
        """the whole local trailing update of one block step in ONE launch: tiles above the global diagonal exit"""
        _lib.check(self.lib.mgpi_gemm_nt_stair_dev(self.ctx.handle, a.handle, b.handle, c.handle, P, p, Q, q, bi0, bj0, nb))

    def gemm_nt_stair_merged(self, c, a, b, P, p, Q, q, bi0, bj0, nb, first_cols):
        """the same update with the first `first_cols` local block columns first, stored write-through and counted on this
        context's progress counter; returns how far the counter moves (wait_count is the other half)"""
        live = _lib.c_int64(0)
        _lib.check(self.lib.mgpi_gemm_nt_stair_merged_dev(self.ctx.handle, a.handle, b.handle, c.handle, P, p, Q, q, bi0, bj0, nb,
                                                          int(first_cols), ctypes.byref(live)))
        return int(live.value)

    def count_reset(self):
        _lib.check(self.lib.mgpi_ctx_count_reset(self.ctx.handle))

    def wait_count(self, owner, target):
        """what this backend enqueues from now on starts once `owner`'s progress counter has reached `target`"""
        _lib.check(self.lib.mgpi_ctx_wait_count(self.ctx.handle, owner.ctx.handle, int(target)))

    def overlaps(self, other):
        """do kernels of this backend's stream and of `other`'s run side by side (bounded device handshake)?"""
        cached = self._overlap_verdict.get(id(other))
        if cached is None:                       # asked once per pair of contexts: the answer is a property of the process
            yes = ctypes.c_int(0)
            _lib.check(self.lib.mgpi_ctx_streams_overlap(self.ctx.handle, other.ctx.handle, ctypes.byref(yes)))
            cached = self._overlap_verdict[id(other)] = bool(yes.value)
        return cached

    def gemm_nn(self, c, a, b):
        _lib.check(self.lib.mgpi_gemm_nn_dev(self.ctx.handle, a.handle, b.handle, c.handle))

    def gemm_nn_neg(self, c, a, b):
        """c = -(a b); c is not read"""
        _lib.check(self.lib.mgpi_gemm_nn_neg_dev(self.ctx.handle, a.handle, b.handle, c.handle))

    def trtri_neg(self, a_local, r0, c0, row_step, col_step, w, batch, out):
        """out rows [j w, (j+1) w) = -(a_local[r0 + j row_step : + w, c0 + j col_step : + w])^-1 (lower triangular blocks)"""
        _lib.check(self.lib.mgpi_trtri_neg_dev(self.ctx.handle, a_local.handle, r0, c0, row_step, col_step, w, batch, out.handle))

    def trsm_left(self, lkk, rhs):
        lkk.mark_factored()
        _lib.check(self.lib.mgpi_trsm_dev(self.ctx.handle, lkk.handle, rhs.handle, 0))

    def trsv(self, lkk, vec, trans):
        _lib.check(self.lib.mgpi_trsv_dev(self.ctx.handle, lkk.handle, vec.handle, int(trans)))

    def gemv(self, a, x, y, trans):
        _lib.check(self.lib.mgpi_gemv_dev(self.ctx.handle, a.handle, x.handle, y.handle, int(trans)))

    def axpy(self, a, x, y):
        _lib.check(self.lib.mgpi_axpy_dev(self.ctx.handle, float(a), x.handle, y.handle))

    def colsq(self, v, out):
        """out[c] = sum_r v[r][c]^2"""
        _lib.check(self.lib.mgpi_colsq_dev(self.ctx.handle, v.handle, out.handle))

    def predict_mean(self, xs, x, terms, alpha):
        desc, count = _lib.make_terms(terms)
        XS = _lib.DeviceMatrix.from_host(self.ctx, xs)
        X = _lib.DeviceMatrix.from_host(self.ctx, x)
        A = _lib.DeviceMatrix.from_host(self.ctx, alpha)
        out = _lib.DeviceMatrix(self.ctx, len(xs), 1)
        _lib.check(self.lib.mgpi_predict_mean_dev(self.ctx.handle, XS.handle, X.handle, desc, count, A.handle, out.handle))
        return out.to_host().reshape(-1)

    def prior_diag(self, xs, terms):
        desc, count = _lib.make_terms(terms)
        xs = _lib.as_f64(xs, ndim=2)
        out = np.empty(len(xs))
        _lib.check(self.lib.mgpi_cov_paired(self.ctx.handle, xs.ctypes.data, xs.ctypes.data, len(xs), xs.shape[1], desc, count,
                                            out.ctypes.data))
        return out


# ---- the schedule --------------------------------------------------------------------------------------------------

class DistributedGP(object):
    """K(x,x; kernel+nugget) built, factored and solved on a P x Q grid of GPUs"""

    def __init__(self, grid, backend, source_x, kernel_terms, solve_terms, nb=512, super_blocks=None, super_min_rows=None):
        self.g = grid
        self.be = backend
        self.nb = int(nb)
        self.x = _lib.as_f64(source_x, ndim=2)
        self.n, self.d = self.x.shape
        self.kernel_terms = list(kernel_terms)
        self.solve_terms = list(solve_terms)
        g = grid
        lcm = g.P * g.Q // math.gcd(g.P, g.Q)
        self.nblk = int(math.ceil(self.n / float(self.nb)))
        self.nblk = (self.nblk + lcm - 1) // lcm * lcm            # every rank owns the same number of blocks
        self.np_ = self.nblk * self.nb
        self.row_blocks = list(range(g.p, self.nblk, g.P))          # global block rows owned, ascending
        self.col_blocks = list(range(g.q, self.nblk, g.Q))
        self.mloc = len(self.row_blocks) * self.nb
        self.nloc = len(self.col_blocks) * self.nb
        # Two-level blocking as on one GPU (csrc/chol.hip, super_width): while at least super_min_rows rows trail behind it, the
        # trailing update is applied once per SUPER-PANEL of super_blocks block columns - contraction 1024 instead of 512, the C
        # tiles cross HBM half as often (isolated: 57.7 -> 63.2 TFLOP/s at 16384 rows).  The first super-panel is one block wide
        # (nothing to hide its panels behind).  MGPI_DIST_SUPER / MGPI_DIST_SUPER_MIN override.
        self.super_blocks = int(os.environ.get('MGPI_DIST_SUPER', 2) if super_blocks is None else super_blocks)
        self.super_min_rows = int(os.environ.get('MGPI_DIST_SUPER_MIN', 14336) if super_min_rows is None else super_min_rows)
        self.A = None
        self.logdet = None
        self.alpha = None
        self.trailing_flops = 0.0   # algorithmic flops of this rank's trailing updates on the main stream (bench.py roofline)
        self.column_flops = 0.0     # ... and of its next-panel column updates on the look-ahead stream
        self._fwd_y = self._fwd_acc = self._fwd_f = None   # forward substitution riding along with factor(source_f=...)
        self.row_panels = None        # L[I][k] for the local block rows, every block step k (filled by factor), as super-panel slabs
        self._row_panel_view = None   # k -> (first row, first column) of L[I > k][k] inside row_panels
        self._dinv = None             # {k: -(L_kk)^-1} for the diagonal blocks this rank owns (variance sweep only)

    def super_panels(self):
        """the block columns of every super-panel, in order: [[0], [1, 2], [3, 4], .., [k], [k+1], ..]"""
        out, k = [], 0
        while k < self.nblk:
            left = self.nblk - k
            w = self.super_blocks if (k > 0 and (left - self.super_blocks) * self.nb >= self.super_min_rows) else 1
            w = max(1, min(w, left))
            out.append(list(range(k, k + w)))
            k += w
        return out

    # -- index helpers ------------------------------------------------------------------------------------
    def _global_index(self, blocks):
        return (np.asarray(blocks)[:, None] * self.nb + np.arange(self.nb)[None, :]).reshape(-1)

    def _first_local(self, blocks, bound):
        """number of local blocks with global index <= bound"""
        return int(np.searchsorted(np.asarray(blocks), bound, side='right'))

    # -- build --------------------------------------------------------------------------------------------
    def build(self):
        be, nb = self.be, self.nb
        self._dinv = None
        gi = self._global_index(self.row_blocks)
        gj = self._global_index(self.col_blocks)
        nvr = int(np.sum(gi < self.n))
        nvc = int(np.sum(gj < self.n))
        xr = np.zeros((self.mloc, self.d))
        xc = np.zeros((self.nloc, self.d))
        xr[:nvr] = self.x[gi[:nvr]]
        xc[:nvc] = self.x[gj[:nvc]]
        if self.A is None:
            self.A = be.alloc(self.mloc, self.nloc)        # zero; the blocks above the global diagonal stay so and are never read
        # one launch per local block column, over the local block rows I >= J only (the factorisation never reads the rest:
        # half the kernel evaluations of the full local rectangle)
        XR, XC = be.points(xr), be.points(xc)
        for lj, J in enumerate(self.col_blocks):
            il = self._first_local(self.row_blocks, J - 1)
            rows = self.mloc - il * nb
            if rows <= 0:
                continue
            be.cov_dev(be.window(XR, il * nb, rows), min(max(nvr - il * nb, 0), rows), be.window(XC, lj * nb, nb),
                       min(max(nvc - lj * nb, 0), nb), self.solve_terms, be.window(self.A, il * nb, rows, lj * nb, nb))
        # padded global indices (>= n) get a unit diagonal so the padded matrix stays positive definite
        for li, I in enumerate(self.row_blocks):
            if (I + 1) * nb > self.n and I % self.g.Q == self.g.q:
                lj = self.col_blocks.index(I)
                t0 = max(0, self.n - I * nb)
                be.fill_diag(be.window(self.A, li * nb + t0, nb - t0, lj * nb + t0, nb - t0), 1.0)
        # the ORIGINAL diagonal of the diagonal blocks this rank owns: what their pivots are measured against when the block is
        # factored, many updates later (a duplicated source row's diagonal entry is rounding noise by then)
        owned = [k for k in range(self.nblk) if k % self.g.P == self.g.p and k % self.g.Q == self.g.q]
        self._diag_slot = dict((k, j) for j, k in enumerate(owned))
        self.diag0 = be.alloc(max(1, len(owned)) * nb, 1)
        for k, j in self._diag_slot.items():
            be.diag(be.window(self.A, (k // self.g.P) * nb, nb, (k // self.g.Q) * nb, nb), be.window(self.diag0, j * nb, nb))

    # -- factor -------------------------------------------------------------------------------------------
    def _panel(self, k, be, bufs, wrow, wcol):
        """block step k, everything that is not the trailing update: factor the diagonal block, solve the panel
        below it, and move the panel to every rank that needs it: `wrow` receives L[I][k] for the local block rows I > k,
        `wcol` L[J][k] for the local block columns J > k (both are nb-column windows of a super-panel's slab, or None when
        this rank has no such rows / columns).  Runs on whatever stream `be` enqueues to; nothing is read back (the failing
        pivot and the log-diagonal sum accumulate on `be`'s device status)."""
        g, nb, A = self.g, self.nb, self.A
        W = be.window
        pk, qk = k % g.P, k % g.Q
        kr, kc = k // g.P, k // g.Q                      # local block indices on the owners
        i0 = self._first_local(self.row_blocks, k)       # local row blocks [i0:) have I > k
        j0 = self._first_local(self.col_blocks, k)
        mt = self.mloc - i0 * nb                         # local trailing rows
        nt = self.nloc - j0 * nb
        if g.q == qk:
            if g.P == 1 and not g.always:
                # one process row: the whole block column is local - the fused step kernels of the single-GPU path factor the
                # diagonal block and solve the rows below it in the same launches (no copy of L_kk, no separate trsm)
                be.potrf_panel(W(A, kr * nb, self.mloc - kr * nb, kc * nb, nb), k * nb, W(self.diag0, self._diag_slot[k] * nb, nb))
            else:
                # 1. diagonal block, down its process column
                lkk = bufs['lkk']
                if g.p == pk:
                    blk = W(A, kr * nb, nb, kc * nb, nb)
                    be.potrf_block(blk, k * nb, W(self.diag0, self._diag_slot[k] * nb, nb))
                    be.copy(lkk, blk)
                g.bcast_col(be, lkk, pk)
                # 2. panel trsm on this process column
                if mt > 0:
                    be.trsm_right(lkk, W(A, i0 * nb, mt, kc * nb, nb))
        # 3. panel along process rows (collectives move contiguous buffers: through `rowtmp` when there is anyone to send to)
        if mt > 0:
            if g.Q > 1 or g.always:
                tmp = W(bufs['rowtmp'], 0, mt)
                if g.q == qk:
                    be.copy(tmp, W(A, i0 * nb, mt, kc * nb, nb))
                g.bcast_row(be, tmp, qk)
                be.copy(wrow, tmp)
            elif not self._panels_in_place:
                be.copy(wrow, W(A, i0 * nb, mt, kc * nb, nb))
        # 4. transposed role along process columns.  The blocks L[J][k] this rank needs (J = q mod Q, J > k) sit,
        #    after step 3, in the row panels of the ranks (J mod P, q) of its own process column.  For a source
        #    grid row pr the wanted J form an arithmetic progression (J = r mod lcm(P,Q)), so gathering them on
        #    the source and scattering them into wcol on the receivers are single strided block copies.
        if nt > 0:
            lcm = g.P * g.Q // math.gcd(g.P, g.Q)
            for pr in range(g.P):
                r = next((c for c in range(lcm) if c % g.Q == g.q and c % g.P == pr), None)
                if r is None:
                    continue
                J0 = k + 1 + ((r - (k + 1)) % lcm)                   # smallest J > k with J = r (mod lcm)
                if J0 >= self.nblk:
                    continue
                count = (self.nblk - 1 - J0) // lcm + 1
                si, sj = lcm // g.P, lcm // g.Q                       # strides in local row / column blocks
                jl0 = (J0 - g.q) // g.Q - j0
                if g.P == 1 and not g.always:
                    # the source is this rank: straight from the row panel into the column panel (on a 1 x 1 grid they are one)
                    if wcol is not wrow:
                        be.copy_blocks(wcol, jl0 * nb, sj * nb, wrow, ((J0 - pr) // g.P - i0) * nb, si * nb, nb, count)
                    continue
                buf = W(bufs['gather'], 0, count * nb)
                if g.p == pr:
                    il0 = (J0 - pr) // g.P - i0
                    be.copy_blocks(buf, 0, nb, wrow, il0 * nb, si * nb, nb, count)
                g.bcast_col(be, buf, pr)
                be.copy_blocks(wcol, jl0 * nb, sj * nb, buf, 0, nb, nb, count)
        return dict(wrow=wrow, wcol=wcol, i0=i0, j0=j0, mt=mt, nt=nt, kblocks=1)

    def _panel_super(self, si, be, bufs):
        """super-panel si: its block columns one after the other - apply the earlier panels of the super-panel to the block
        column (contraction nb, on the process column that owns it), then the block step of `_panel` - with the row / column
        panels landing side by side in the super-panel's slabs.  Returns what `_update` needs to apply the whole super-panel
        (contraction len(blocks) * nb) to the local trailing matrix behind it."""
        g, nb = self.g, self.nb
        W = be.window
        blocks = self.super_panels_[si]
        k0, kl = blocks[0], blocks[-1]
        i00 = self._first_local(self.row_blocks, k0)
        j00 = self._first_local(self.col_blocks, k0)
        off, col0 = self._slab_offset[si], self._slab_col[si]
        same = self._panels_in_place and g.P == 1          # 1 x 1: the column panel IS the row panel
        done = []
        for h, k in enumerate(blocks):
            i0 = self._first_local(self.row_blocks, k)
            j0 = self._first_local(self.col_blocks, k)
            mt, nt = self.mloc - i0 * nb, self.nloc - j0 * nb
            if g.q == k % g.Q:                            # the block column of step k is local: bring it up to date
                jl = k // g.Q
                for pan in done:
                    self._update(pan, jl, jl + 1, be)
            wrow = W(self.row_panels, off + (i0 - i00) * nb, mt, col0 + h * nb, nb) if mt > 0 else None
            wcol = (wrow if same else W(bufs['wcol'], (j0 - j00) * nb, nt, h * nb, nb)) if nt > 0 else None
            done.append(self._panel(k, be, bufs, wrow, wcol))
        # the slabs behind the LAST block of the super-panel, all of its columns
        i0 = self._first_local(self.row_blocks, kl)
        j0 = self._first_local(self.col_blocks, kl)
        mt, nt = self.mloc - i0 * nb, self.nloc - j0 * nb
        width = len(blocks) * nb
        wrow = W(self.row_panels, off + (i0 - i00) * nb, mt, col0, width) if mt > 0 else None
        return dict(wrow=wrow,
                    wcol=(wrow if same else W(bufs['wcol'], (j0 - j00) * nb, nt, 0, width)) if nt > 0 else None,
                    i0=i0, j0=j0, mt=mt, nt=nt, kblocks=len(blocks))

    def _update(self, pan, c_lo, c_hi, be, first_cols=0):
        """trailing update of the local block columns [c_lo, c_hi) with the panel(s) `pan` describes:
        A[I][J] -= sum_k L[I][k] L[J][k]^T for the local blocks with I >= J.  One launch (staircase gemm).
        first_cols > 0: the merged form - those first columns first, counted; returns how far the progress counter moves."""
        g, nb = self.g, self.nb
        i0, j0 = pan['i0'], pan['j0']
        c_hi = min(c_hi, len(self.col_blocks))
        if pan['mt'] <= 0 or c_lo >= c_hi:
            return 0
        W = be.window
        kw = pan['wrow'].cols if hasattr(pan['wrow'], 'cols') else pan['wrow'].shape[1]
        live = 0
        cwin = W(self.A, i0 * nb, pan['mt'], c_lo * nb, (c_hi - c_lo) * nb)
        bwin = W(pan['wcol'], (c_lo - j0) * nb, (c_hi - c_lo) * nb, 0, kw)
        if first_cols > 0:
            live = be.gemm_nt_stair_merged(cwin, pan['wrow'], bwin, g.P, g.p, g.Q, g.q, i0, c_lo, nb, first_cols)
        else:
            be.gemm_nt_stair(cwin, pan['wrow'], bwin, g.P, g.p, g.Q, g.q, i0, c_lo, nb)
        flops = 0.0                                              # flop bookkeeping only (bench.py roofline)
        for jl in range(c_lo, c_hi):
            J = self.col_blocks[jl]
            il = max(self._first_local(self.row_blocks, J - 1), i0)
            if il < len(self.row_blocks):
                rows = self.mloc - il * nb
                diag = nb if self.row_blocks[il] == J else 0      # a diagonal block counts as its lower half
                flops += (2.0 * (rows - diag) * nb + diag * (nb + 1.0)) * kw
        if be is self.be:
            self.trailing_flops += flops                         # launches timed by the main context's profile slot 0
        else:
            self.column_flops += flops                           # next-panel column updates on the look-ahead stream
        return live

    def _forward_blocks(self, blocks, be):
        """the forward substitution L y = f for the block steps `blocks`, right-looking: each needs only its own, finished
        block column - run right behind that column's panel, on the stream the panels run on, beside the trailing update
        (the single-GPU path does the same on its third stream, csrc/chol.hip mgpi_posv_dev).  `_acc` (local ROW order)
        collects  - sum_{J < I, J local} L[I][J] y_J  for every local block row I."""
        if self._fwd_y is None:
            return
        g, nb, A, W = self.g, self.nb, self.A, be.window
        y, acc = self._fwd_y, self._fwd_acc
        for k in blocks:
            pk, qk = k % g.P, k % g.Q
            kr, kc = k // g.P, k // g.Q
            yk = W(y, k * nb, nb)
            if g.p == pk:
                s = W(acc, kr * nb, nb)
                g.reduce_row(be, s, qk)
                if g.q == qk:
                    be.axpy(1.0, s, yk)
                    be.trsv(W(A, kr * nb, nb, kc * nb, nb), yk, 0)
            if g.q == qk:                                       # y_k multiplies the blocks L[I][k]: process column qk
                g.bcast_col(be, yk, pk)
                i0 = self._first_local(self.row_blocks, k)      # local row blocks [i0:) have I > k
                if i0 < len(self.row_blocks):
                    mt = self.mloc - i0 * nb
                    be.gemv(W(A, i0 * nb, mt, kc * nb, nb), yk, W(acc, i0 * nb, mt), 0)       # acc_I -= L[I][k] y_k

    def factor(self, lookahead=None, source_f=None):
        """2-D block-cyclic right-looking Cholesky in super-panels.
        source_f: a right-hand side whose forward substitution runs along with the factorisation, block step by block step
        (`solve(source_f)` afterwards is then the backward sweep alone).
  With look-ahead (default; MGPI_DIST_LOOKAHEAD=0 turns it
        off) the update of the block columns of super-panel s+1 and then that super-panel's work - diagonal factorisations,
        panel solves and all their broadcasts - run on a second stream while the rest of the trailing update of super-panel s
        runs on the main stream, so the xGMI transfers and the latency-bound panels hide behind the MFMA work.  Every rank
        issues the same collectives in the same order in both modes, and all of them on ONE stream per mode (the look-ahead
        stream, or the main stream without look-ahead)."""
        g, be, nb = self.g, self.be, self.nb
        self._dinv = None
        if lookahead is None:
            lookahead = os.environ.get('MGPI_DIST_LOOKAHEAD', '1') != '0'
        side = be.side_backend() if lookahead else be
        supers = self.super_panels_ = self.super_panels()
        wmax = max(len(b) for b in supers) * nb
        # panel buffers: two sets (parity of the super-panel), allocated once
        # (on grids that never use one of them - no column panels of their own on 1 x 1, nothing to gather with one process row,
        # nothing to stage with one process column - it is not allocated)
        one_row, one_col = g.P == 1 and not g.always, g.Q == 1 and not g.always
        bufs = [dict(lkk=be.alloc(nb, nb, zero=False),
                     wcol=None if (one_row and one_col) else be.alloc(max(self.nloc, nb), wmax, zero=False),
                     gather=None if one_row else be.alloc(max(self.nloc, nb), nb, zero=False),
                     rowtmp=None if one_col else be.alloc(max(self.mloc, nb), nb, zero=False))
                for _ in range(2)]
        # persistent row panels: for every super-panel a slab with the blocks L[I][k] of the local block rows I > (its first
        # block), the panels of its block columns side by side (sum_k (N - k nb)/P x nb doubles = 8 N^2 / (2P) bytes: 8.6 GB per
        # rank at N = 65536 on 2 x 4).  The variance sweep reuses them, so L crosses xGMI once, not twice.
        # One process column (Q = 1): the block columns of a super-panel are adjacent LOCAL columns and L[I][k] of the local
        # block rows is what the panel solve leaves in the local matrix - the "slab" of a super-panel is then a window of A
        # itself (no copy per block step, no 8 N^2 / 2P bytes of slabs); on a 1 x 1 grid the column panels are the same window.
        self._panels_in_place = g.Q == 1 and not g.always
        offs, cols0, total, views = [], [], 0, {}
        for blocks in supers:
            i00 = self._first_local(self.row_blocks, blocks[0])
            base, col0 = (i00 * nb, (blocks[0] // g.Q) * nb) if self._panels_in_place else (total, 0)
            offs.append(base)
            cols0.append(col0)
            for h, k in enumerate(blocks):
                views[k] = (base + (self._first_local(self.row_blocks, k) - i00) * nb, col0 + h * nb)
            total += self.mloc - i00 * nb
        total = max(total, nb)
        self._slab_offset, self._slab_col, self._row_panel_view = offs, cols0, views
        if self._panels_in_place:
            self.row_panels = self.A
        elif self.row_panels is None or self.row_panels is self.A or self.row_panels.rows != total or self.row_panels.cols != wmax:
            self.row_panels = be.alloc(total, wmax, zero=False)
        self._fwd_y = self._fwd_acc = self._fwd_f = None
        if source_f is not None and os.environ.get('MGPI_DIST_FORWARD', '1') != '0':
            f = np.zeros(self.np_)
            f[:self.n] = _lib.as_f64(source_f).reshape(-1)
            self._fwd_y = be.alloc(self.np_, 1)                 # replicated right-hand side; block k becomes y_k where it is solved
            be.upload(self._fwd_y, f)
            self._fwd_acc = be.alloc(max(self.mloc, nb), 1)
            self._fwd_f = f
        side.status_reset()
        side.wait(be)                                           # the build (and the allocations' fills) precede the first panel
        # Merged trailing update, as on one GPU (csrc/chol.hip): while at least merged_min_rows local rows trail, ONE launch
        # updates the whole local trailing matrix - the next super-panel's local block columns first, stored write-through and
        # counted - and the look-ahead context starts on that count while the rest of the same launch runs, instead of two
        # launches side by side.  Only where the two contexts' kernels do overlap (bounded device handshake; not under
        # counter collection or serialising debug switches); MGPI_DIST_MERGED=0 keeps the two launches.
        merged = lookahead and os.environ.get('MGPI_DIST_MERGED', '1') != '0' and side is not be and be.overlaps(side)
        merged_min = int(os.environ.get('MGPI_DIST_MERGED_MIN', 8192))
        if os.environ.get('MGPI_DIST_DEBUG'):
            sys.stderr.write('factor: lookahead=%s merged=%s\n' % (lookahead, merged))
        target = 0
        if merged:
            be.count_reset()

        ncols = len(self.col_blocks)
        pan = self._panel_super(0, side, bufs[0])
        be.wait(side)
        self._forward_blocks(supers[0], side)                   # behind the point the main stream waits for
        for s in range(len(supers)):
            nxt = s + 1 < len(supers)
            j0 = pan['j0']
            if not lookahead or not nxt:
                self._update(pan, j0, ncols, be)
                if nxt:
                    pan = self._panel_super(s + 1, be, bufs[(s + 1) % 2])
                    self._forward_blocks(supers[s + 1], be)
                continue
            # (a) main stream: the rest of the trailing update of super-panel s.  (b) side stream (high priority): the block
            # columns of the next super-panel, then that super-panel - its diagonal factorisations, panel solves and all of
            # its broadcasts.  The two write disjoint block columns and both only read super-panel s.  The side stream waits
            # for what the main stream had queued BEFORE (a): the rest of step s-1, which wrote the columns (b) touches.
            side.wait(be)
            nown = sum(1 for k in supers[s + 1] if k % g.Q == g.q)   # they are the first local trailing columns
            if merged and nown and ncols - j0 > nown and pan['mt'] >= merged_min:
                target += self._update(pan, j0, ncols, be, first_cols=nown)
                side.wait_count(be, target)
            else:
                self._update(pan, j0 + nown, ncols, be)
                if nown:
                    self._update(pan, j0, j0 + nown, side)
            nxt_pan = self._panel_super(s + 1, side, bufs[(s + 1) % 2])
            be.wait(side)
            self._forward_blocks(supers[s + 1], side)           # the trailing update does not wait for these
            pan = nxt_pan
        if side is not be:
            be.wait(side)                                       # the last forward steps
        # logdet and the first failing pivot, agreed by everyone (one read-back for the whole factorisation)
        info_first, logdiag = side.status_read()
        if g.world > 1 or g.always:
            s = be.alloc(1, 1)
            be.upload(s, np.array([logdiag]))
            g.all_reduce(be, s, 'sum')
            logdiag = float(be.download(s).reshape(-1)[0])
            m = be.alloc(1, 1)
            be.upload(m, np.array([float(-info_first) if info_first else -1e300]))
            g.all_reduce(be, m, 'max')                        # max of (-info) = smallest failing pivot
            worst = float(be.download(m).reshape(-1)[0])
            info_first = int(-worst) if worst > -1e299 else 0
        if info_first:
            self._fwd_y = self._fwd_acc = self._fwd_f = None
            raise _lib.NotPositiveDefinite(info_first, 'covariance is not positive definite! (pivot %d)' % info_first)
        self.logdet = 2.0 * logdiag
        return self.logdet

    # -- solve (replicated right-hand side) -----------------------------------------------------------------
    def solve(self, source_f):
        """alpha = K^-1 f by two block sweeps over the distributed factor, right-looking like the single-GPU sweeps: as soon as
        a block of the solution is known it is applied to everything it feeds - ONE wide, HBM-bound launch over the local
        blocks L[I > k][k] (forward) resp. L[k][J < k] (backward) - into a local accumulator of partial sums.  Per block step
        ONE small reduce of the block's partial sums (inside the process row of the diagonal block going forward, its process
        column going backward) and ONE broadcast of the nb solved entries to the ranks that multiply with them: going
        forward the diagonal block's process COLUMN (they hold L[I][k]), going backward its process ROW - P resp. Q members
        instead of the whole world.  The solution is assembled once at the end (one all-reduce of N doubles, every block
        contributed by its owner alone)."""
        g, be, nb = self.g, self.be, self.nb
        A, W = self.A, be.window
        f = np.zeros(self.np_)
        f[:self.n] = _lib.as_f64(source_f).reshape(-1)
        rode = getattr(self, '_fwd_f', None) is not None and np.array_equal(f, self._fwd_f)
        if rode:
            y = self._fwd_y                                     # factor(source_f=...) left y here, block k on its diagonal owner
            self._fwd_y = self._fwd_acc = self._fwd_f = None    # the backward sweep overwrites it
        else:
            y = be.alloc(self.np_, 1)                           # replicated right-hand side; block k becomes the solution on
            be.upload(y, f)                                     # the ranks that need it
        # forward: L y = f.  acc (local ROW order) collects  - sum_{J < I, J local} L[I][J] y_J  for every local block row I
        acc = be.alloc(max(self.mloc, nb), 1)
        for k in (() if rode else range(self.nblk)):
            pk, qk = k % g.P, k % g.Q
            kr, kc = k // g.P, k // g.Q
            yk = W(y, k * nb, nb)
            if g.p == pk:
                s = W(acc, kr * nb, nb)
                g.reduce_row(be, s, qk)
                if g.q == qk:
                    be.axpy(1.0, s, yk)
                    be.trsv(W(A, kr * nb, nb, kc * nb, nb), yk, 0)
            if g.q == qk:                                       # y_k multiplies the blocks L[I][k]: process column qk
                g.bcast_col(be, yk, pk)
                i0 = self._first_local(self.row_blocks, k)      # local row blocks [i0:) have I > k
                if i0 < len(self.row_blocks):
                    mt = self.mloc - i0 * nb
                    be.gemv(W(A, i0 * nb, mt, kc * nb, nb), yk, W(acc, i0 * nb, mt), 0)       # acc_I -= L[I][k] y_k
        # backward: L^T x = y  (y_k is where it was solved: on the owner of the diagonal block).  accb (local COLUMN order)
        # collects  - sum_{I > J, I local} L[I][J]^T x_I  for every local block column J
        accb = be.alloc(max(self.nloc, nb), 1)
        for k in range(self.nblk - 1, -1, -1):
            pk, qk = k % g.P, k % g.Q
            kr, kc = k // g.P, k // g.Q
            xk = W(y, k * nb, nb)
            if g.q == qk:
                s = W(accb, kc * nb, nb)
                g.reduce_col(be, s, pk)
                if g.p == pk:
                    be.axpy(1.0, s, xk)
                    be.trsv(W(A, kr * nb, nb, kc * nb, nb), xk, 1)
            if g.p == pk:                                       # x_k multiplies the blocks L[k][J]^T: process row pk
                g.bcast_row(be, xk, qk)
                jc = self._first_local(self.col_blocks, k - 1)  # local col blocks [0:jc) have J < k
                if jc > 0:
                    be.gemv(W(A, kr * nb, nb, 0, jc * nb), xk, W(accb, 0, jc * nb), 1)          # accb_J -= L[k][J]^T x_k
        if g.world > 1 or g.always:
            out = be.alloc(self.np_, 1)                         # zero; every block from its owner, then one sum
            for k in range(self.nblk):
                if (g.p, g.q) == (k % g.P, k % g.Q):
                    be.copy(W(out, k * nb, nb), W(y, k * nb, nb))
            g.all_reduce(be, out, 'sum')
            y = out
        self.alpha = be.download(y).reshape(-1)[:self.n].copy()
        return self.alpha

    # -- predict ------------------------------------------------------------------------------------------
    def mean(self, target_x):
        """posterior mean, queries sharded over all ranks (x and alpha are replicated), result replicated"""
        g, be = self.g, self.be
        xs = _lib.as_f64(target_x, ndim=2)
        m = len(xs)
        per = (m + g.world - 1) // g.world
        lo, hi = min(m, g.rank * per), min(m, (g.rank + 1) * per)
        part = np.zeros(per)
        if hi > lo:
            part[:hi - lo] = be.predict_mean(xs[lo:hi], self.x, self.kernel_terms, self.alpha)
        if g.world == 1 and not g.always:
            return part[:m]
        mine = be.alloc(per, 1)
        be.upload(mine, part)
        every = be.alloc(per * g.world, 1)
        g.all_gather(be, mine, every)
        return be.download(every).reshape(-1)[:m].copy()

    def _ensure_dinv(self):
        """-(L_kk)^-1 of the diagonal blocks this rank owns, all in one batched call: they sit at a constant stride in
        the local matrix (k = k0, k0 + lcm(P, Q), ..).  Variance sweep only (error budget: DESIGN.md)."""
        if self._dinv is not None:
            return
        g, be, nb = self.g, self.be, self.nb
        mine = [k for k in range(self.nblk) if k % g.P == g.p and k % g.Q == g.q]
        self._dinv = {}
        if not mine:
            return
        lcm = g.P * g.Q // math.gcd(g.P, g.Q)
        assert all(b - a == lcm for a, b in zip(mine, mine[1:]))
        out = be.alloc(len(mine) * nb, nb, zero=False)
        be.trtri_neg(self.A, (mine[0] // g.P) * nb, (mine[0] // g.Q) * nb, (lcm // g.P) * nb, (lcm // g.Q) * nb, nb, len(mine), out)
        for j, k in enumerate(mine):
            self._dinv[k] = be.window(out, j * nb, nb)

    def variance(self, target_x, m_slab=None):
        """pointwise posterior variance: V = L^-1 K(x, x*) by the same block-cyclic sweep (queries sharded over
        process columns, rows of V distributed like the rows of L), then column norms.  The queries are processed in slabs
        (`m_slab` per process column, default sized so that a slab's V stays near 4 GB per rank; MGPI_DIST_VAR_SLAB), as on
        one GPU: 10^6 queries (BASELINE.json configs[4]) never need an (N/P) x (M/Q) array in one piece."""
        g = self.g
        xs = _lib.as_f64(target_x, ndim=2)
        m = len(xs)
        if m_slab is None:
            m_slab = int(os.environ.get('MGPI_DIST_VAR_SLAB', 0)) or max(128, min(65536, (1 << 29) // max(self.mloc, 1) // 128 * 128))
        m_slab = max(128, (int(m_slab) + 127) // 128 * 128)
        step = m_slab * g.Q                                   # queries per sweep, all process columns together
        out = np.empty(m)
        for lo in range(0, m, step):
            out[lo:lo + step] = self._variance_slab(xs[lo:lo + step])
        return out

    def _variance_slab(self, xs):
        g, be, nb = self.g, self.be, self.nb
        A, W = self.A, be.window
        m = len(xs)
        per = (m + g.Q - 1) // g.Q
        per = (per + 127) // 128 * 128
        lo, hi = min(m, g.q * per), min(m, (g.q + 1) * per)
        gi = self._global_index(self.row_blocks)
        nvr = int(np.sum(gi < self.n))
        xr = np.zeros((self.mloc, self.d))
        xr[:nvr] = self.x[gi[:nvr]]
        xq = np.zeros((per, self.d))
        xq[:hi - lo] = xs[lo:hi]
        V = be.alloc(self.mloc, per)
        be.cov(xr, xq, self.kernel_terms, V, nvr, hi - lo)
        use_inv = os.environ.get('MGPI_DIST_VAR_DINV', '1') != '0'
        if use_inv:
            self._ensure_dinv()
        # Block rows are solved one at a time; the rows behind a SUPER-PANEL of the factorisation are updated once, with all of its
        # block columns at a time (the slabs the factorisation left side by side in row_panels: contraction 2 nb while many rows
        # trail, as in the factorisation itself) - half the update launches, each at the longer contraction's rate.
        supers = self.super_panels_
        wmax = max(len(b) for b in supers)
        vks = be.alloc(wmax * nb, per, zero=False)               # the solved block rows of one super-panel, stacked
        lkk = be.alloc(nb, nb, zero=False)
        for si, blocks in enumerate(supers):
            for h, k in enumerate(blocks):
                pk, qk = k % g.P, k % g.Q
                kr, kc = k // g.P, k // g.Q
                vk = W(vks, h * nb, nb)
                if g.p == pk:
                    Vk = W(V, kr * nb, nb)
                    # bring block row k up to date with the earlier block rows of this super-panel: L[k][k'] is the block row
                    # I = k of the row panel of step k'
                    for hh in range(h):
                        r0, c0 = self._row_panel_view[blocks[hh]]
                        skip = kr - self._first_local(self.row_blocks, blocks[hh])      # local block rows between k' and k
                        be.gemm_nn(Vk, W(self.row_panels, r0 + skip * nb, nb, c0, nb), W(vks, hh * nb, nb))
                    # diagonal block to the whole process row pk, which solves its slice of block row k of V.  With the
                    # inverse diagonal blocks (default) what travels is -(L_kk)^-1 and the solve is one gemm instead of the
                    # 8-step substitution chain, which would otherwise sit on the critical path of every block step
                    if use_inv:
                        inv = lkk
                        if g.Q > 1 or g.always:
                            if g.q == qk:
                                be.copy(lkk, self._dinv[k])
                            g.bcast_row(be, lkk, qk)
                        else:
                            inv = self._dinv[k]                                  # nobody to send it to: use it where it is
                        be.gemm_nn_neg(vk, inv, Vk)                              # -(-(L_kk)^-1) V_k, straight into the buffer
                        be.copy(Vk, vk)                                          # that is broadcast next
                    else:
                        if g.q == qk:
                            be.copy(lkk, W(A, kr * nb, nb, kc * nb, nb))
                        g.bcast_row(be, lkk, qk)
                        be.trsm_left(lkk, Vk)
                        be.copy(vk, Vk)
                g.bcast_col(be, vk, pk)
            # the rows behind the super-panel: every rank kept its row panels from the factorisation (self.row_panels), so the
            # factor is NOT broadcast a second time
            kl = blocks[-1]
            i0 = self._first_local(self.row_blocks, kl)
            mt = self.mloc - i0 * nb
            if mt > 0:
                i00 = self._first_local(self.row_blocks, blocks[0])
                width = len(blocks) * nb
                be.gemm_nn(W(V, i0 * nb, mt), W(self.row_panels, self._slab_offset[si] + (i0 - i00) * nb, mt, self._slab_col[si], width), W(vks, 0, width))
        sq_dev = be.alloc(per, 1)
        be.colsq(V, sq_dev)
        if g.P > 1 or g.always:
            g.all_reduce(be, sq_dev, 'sum', column_only=True)
        sq = be.download(sq_dev).reshape(-1)
        var_q = np.zeros(per)
        if hi > lo:
            var_q[:hi - lo] = be.prior_diag(xs[lo:hi], self.kernel_terms) - sq[:hi - lo]
        if g.world == 1 and not g.always:
            return var_q[:m]
        mine = be.alloc(per, 1)
        be.upload(mine, var_q)
        every = be.alloc(per * g.world, 1)
        g.all_gather(be, mine, every)
        parts = be.download(every).reshape(g.world, per)
        # ranks of one process column hold identical slices; take grid row 0
        return np.concatenate([parts[g.rank_of(0, qq)] for qq in range(g.Q)])[:m].copy()
