"""Multi-GPU dense GP: 2-D block-cyclic right-looking Cholesky over one node.

One process per GPU, launched by ``torch.distributed.run``; ``torch.distributed`` (backend ``nccl`` = RCCL over
xGMI on the GPU box, ``gloo`` in the CPU tests) is the transport, torch tensors are the communication
buffers, and every tile operation goes through a *tile backend*:

  * ``HipTileBackend`` (this file): wraps the tensors' device memory with ``mgpi_mat_wrap`` and runs the same
    HIP kernels as the single-GPU path (cov tile build, fp64-MFMA gemm, potf2/trsm panel, trsv pieces).
  * the CPU tests inject a numpy backend built on the oracle (tests/dist_cpu_backend.py) so that the
    schedule below - ownership, broadcast order, index bookkeeping - is exercised with gloo at world size
    2 and 4 without a GPU.  The product never constructs that backend.

Distribution: K is cut into nb x nb blocks (nb = the outer Cholesky block, 512); block (I, J) lives on grid
position (I mod P, J mod Q) of a P x Q process grid (1x1, 1x2, 2x2, 2x4).  Only a broadcast pattern is needed
(no all-reduce): per block step k
  1. the owner factors the diagonal block and broadcasts it down its process column,
  2. that column's ranks solve their panel blocks  A[I][k] <- A[I][k] L_kk^-T,
  3. the panel is broadcast along process rows, then (transposed role) along process columns,
  4. every rank updates its local trailing blocks  A[I][J] -= L[I][k] L[J][k]^T  (I >= J > k) on fp64 MFMA.
The solves use a replicated right-hand side: per block step one small reduce inside a process row (forward)
or column (backward) and one world broadcast of the nb solved entries.
"""
import ctypes
import math
import os

import numpy as np

from . import _lib

__all__ = ['ProcessGrid', 'HipTileBackend', 'DistributedGP', 'choose_grid']


def choose_grid(world):
    """P x Q with P <= Q, as square as possible: 1->1x1, 2->1x2, 4->2x2, 8->2x4"""
    p = int(math.floor(math.sqrt(world)))
    while world % p:
        p -= 1
    return p, world // p


class ProcessGrid(object):
    """rank <-> (p, q) on a P x Q grid (row-major), with one process group per grid row and per grid column"""

    def __init__(self, dist, P=None, Q=None, always_communicate=False, host_staged=False):
        self.dist = dist
        self.always = bool(always_communicate)   # tests: issue the collectives even inside size-1 groups
        # rehearsal transport: device buffers cross through host copies, so that several ranks can SHARE one GPU
        # under gloo (RCCL refuses two ranks on one device).  tests/test_gpu_distributed.py uses it to run the real
        # multi-rank tile distribution with the HIP backend on a one-GPU box; never set on a multi-GPU launch.
        self.host_staged = bool(host_staged)
        self.world = dist.get_world_size()
        self.rank = dist.get_rank()
        if P is None:
            P, Q = choose_grid(self.world)
        assert P * Q == self.world, 'grid %dx%d does not match world size %d' % (P, Q, self.world)
        self.P, self.Q = P, Q
        self.p, self.q = divmod(self.rank, Q)
        # every rank must create every group, in the same order
        self.row_groups = [dist.new_group([pp * Q + qq for qq in range(Q)]) for pp in range(P)]
        self.col_groups = [dist.new_group([pp * Q + qq for pp in range(P)]) for qq in range(Q)]

    def rank_of(self, p, q):
        return p * self.Q + q

    def _issue(self, collective, tensor):
        if self.host_staged and getattr(tensor, 'is_cuda', False):
            host = tensor.cpu()                  # waits for the producing kernels on the current stream
            collective(host)
            tensor.copy_(host)
        else:
            collective(tensor)

    def bcast_row(self, tensor, p, src_q):
        """broadcast inside grid row p from column src_q (only ranks of that row call this)"""
        if self.Q > 1 or self.always:
            self._issue(lambda t: self.dist.broadcast(t, src=self.rank_of(p, src_q), group=self.row_groups[p]), tensor)

    def bcast_col(self, tensor, q, src_p):
        if self.P > 1 or self.always:
            self._issue(lambda t: self.dist.broadcast(t, src=self.rank_of(src_p, q), group=self.col_groups[q]), tensor)

    def bcast_world(self, tensor, src):
        if self.world > 1 or self.always:
            self._issue(lambda t: self.dist.broadcast(t, src=src), tensor)

    def reduce_row(self, tensor, p, dst_q):
        if self.Q > 1 or self.always:
            self._issue(lambda t: self.dist.reduce(t, dst=self.rank_of(p, dst_q), op=self.dist.ReduceOp.SUM,
                                                   group=self.row_groups[p]), tensor)

    def reduce_col(self, tensor, q, dst_p):
        if self.P > 1 or self.always:
            self._issue(lambda t: self.dist.reduce(t, dst=self.rank_of(dst_p, q), op=self.dist.ReduceOp.SUM,
                                                   group=self.col_groups[q]), tensor)

    def all_reduce(self, tensor, op='sum', col=None):
        """all-reduce over the world, or inside process column `col`"""
        red = self.dist.ReduceOp.SUM if op == 'sum' else self.dist.ReduceOp.MAX
        group = None if col is None else self.col_groups[col]
        self._issue(lambda t: self.dist.all_reduce(t, op=red, group=group), tensor)

    def all_gather(self, parts, mine):
        if self.host_staged and getattr(mine, 'is_cuda', False):
            host = [t.cpu() for t in parts]
            self.dist.all_gather(host, mine.cpu())
            for dst, src in zip(parts, host):
                dst.copy_(src)
        else:
            self.dist.all_gather(parts, mine)


class HipTileBackend(object):
    """tile operations on torch CUDA tensors through libmgpi_hip.so (the same kernels as the 1-GPU path)"""

    def __init__(self, torch, ctx=None, make_current=True):
        self.torch = torch
        self.ctx = ctx if ctx is not None else _lib.default_context()
        self.lib = self.ctx.lib
        self.device = torch.device('cuda', self.ctx.device)
        # make OUR stream torch's current stream: torch's own ops on the communication buffers (zeros, copy_,
        # add_) are then enqueued in order with the tile kernels, and NCCL collectives order themselves against
        # the current stream, so no host-side synchronisation is needed for correctness
        torch.cuda.set_device(self.device)
        self.stream = torch.cuda.ExternalStream(self.lib.mgpi_ctx_stream(self.ctx.handle), device=self.device)
        if make_current:
            torch.cuda.set_stream(self.stream)
        self._side = None
        self._side_done = None

    # -- second stream for the look-ahead panel (its own mgpi ctx = its own HIP stream and workspace) -----------
    def side_backend(self):
        if self._side is None:
            ctx2 = _lib.Context(device=self.ctx.device, nb=0)
            self._side = HipTileBackend(self.torch, ctx2, make_current=False)
        return self._side

    def record(self):
        """event on the main stream"""
        ev = self.torch.cuda.Event()
        ev.record(self.stream)
        return ev

    def on_side(self, after):
        """context: torch's current stream = the side stream, which first waits for the event `after`"""
        side = self.side_backend()
        side.stream.wait_event(after)
        return self.torch.cuda.stream(side.stream)

    def join_side(self):
        """the main stream waits for everything queued on the side stream so far"""
        side = self.side_backend()
        ev = self.torch.cuda.Event()
        ev.record(side.stream)
        self.stream.wait_event(ev)

    # -- memory -------------------------------------------------------------------------------------------
    def zeros(self, rows, cols=None):
        shape = (rows,) if cols is None else (rows, cols)
        return self.torch.zeros(shape, dtype=self.torch.float64, device=self.device)

    def empty(self, rows, cols=None):
        """uninitialised buffer: for panels that are fully overwritten by a copy or a broadcast"""
        shape = (rows,) if cols is None else (rows, cols)
        return self.torch.empty(shape, dtype=self.torch.float64, device=self.device)

    def from_numpy(self, a):
        return self.torch.as_tensor(np.ascontiguousarray(a, dtype=np.float64)).to(self.device)

    def to_numpy(self, t):
        self.fence()
        return t.detach().cpu().numpy()

    def fence(self):
        """order our stream against torch's: all queued tile kernels and all torch/NCCL work are complete"""
        self.ctx.sync()
        self.torch.cuda.synchronize(self.device)

    def _wrap(self, t, factored=0):
        """mgpi_mat view of a 2-D tensor (or a vector): row stride from the tensor, no copy"""
        if t.dim() == 1:
            rows, cols, ld = t.shape[0], 1, 1
            assert t.stride(0) == 1
        else:
            rows, cols = t.shape
            ld = t.stride(0)
            assert t.stride(1) == 1, 'tile views must be row-major'
        h = ctypes.c_void_p()
        _lib.check(self.lib.mgpi_mat_wrap(self.ctx.handle, ctypes.c_void_p(t.data_ptr()), rows, cols, ld, factored,
                                          ctypes.byref(h)))
        return h

    def _free(self, *handles):
        for h in handles:
            self.lib.mgpi_mat_free(h)

    # -- tile kernels -------------------------------------------------------------------------------------
    def cov(self, x_rows, x_cols, terms, out, n_rows_valid, n_cols_valid):
        """out[i][j] = k(x_rows[i], x_cols[j]) for i < n_rows_valid, j < n_cols_valid, zero elsewhere"""
        desc, count = _lib.make_terms(terms)
        XR = _lib.DeviceMatrix.from_host(self.ctx, x_rows)
        XC = _lib.DeviceMatrix.from_host(self.ctx, x_cols)
        h = self._wrap(out)
        try:
            _lib.check(self.lib.mgpi_cov_block_dev(self.ctx.handle, XR.handle, n_rows_valid, XC.handle, n_cols_valid,
                                                   desc, count, h))
        finally:
            self._free(h)
        self.ctx.sync()

    def potrf(self, block):
        """in-place Cholesky of a diagonal block; returns (info, sum log diag)"""
        h = self._wrap(block)
        logdet = ctypes.c_double(0.0)
        try:
            st = self.lib.mgpi_potrf_dev(self.ctx.handle, h, ctypes.byref(logdet))
        finally:
            self._free(h)
        if st < 0:
            _lib.check(st)
        return st, 0.5 * logdet.value

    def trsm_right(self, lkk, rows):
        a, b = self._wrap(lkk, 1), self._wrap(rows)
        try:
            _lib.check(self.lib.mgpi_trsm_right_dev(self.ctx.handle, a, b))
        finally:
            self._free(a, b)

    def gemm_nt(self, c, a, b):
        hc, ha, hb = self._wrap(c), self._wrap(a), self._wrap(b)
        try:
            _lib.check(self.lib.mgpi_gemm_nt_dev(self.ctx.handle, ha, hb, hc, 0))
        finally:
            self._free(hc, ha, hb)

    def gemm_nt_stair(self, c, a, b, P, p, Q, q, bi0, bj0, nb):
        """the whole local trailing update of one block step in ONE launch: tiles above the global diagonal exit"""
        hc, ha, hb = self._wrap(c), self._wrap(a), self._wrap(b)
        try:
            _lib.check(self.lib.mgpi_gemm_nt_stair_dev(self.ctx.handle, ha, hb, hc, P, p, Q, q, bi0, bj0, nb))
        finally:
            self._free(hc, ha, hb)

    def gemm_nn(self, c, a, b):
        hc, ha, hb = self._wrap(c), self._wrap(a), self._wrap(b)
        try:
            _lib.check(self.lib.mgpi_gemm_nn_dev(self.ctx.handle, ha, hb, hc))
        finally:
            self._free(hc, ha, hb)

    def trtri_neg(self, a_local, r0, c0, row_step, col_step, w, batch, out):
        """out rows [j w, (j+1) w) = -(a_local[r0 + j row_step : + w, c0 + j col_step : + w])^-1 (lower triangular blocks)"""
        ha, ho = self._wrap(a_local), self._wrap(out)
        try:
            _lib.check(self.lib.mgpi_trtri_neg_dev(self.ctx.handle, ha, r0, c0, row_step, col_step, w, batch, ho))
        finally:
            self._free(ha, ho)

    def trsm_left(self, lkk, rhs):
        a, b = self._wrap(lkk, 1), self._wrap(rhs)
        try:
            _lib.check(self.lib.mgpi_trsm_dev(self.ctx.handle, a, b, 0))
        finally:
            self._free(a, b)

    def trsv(self, lkk, vec, trans):
        a, b = self._wrap(lkk, 1), self._wrap(vec)
        try:
            _lib.check(self.lib.mgpi_trsv_dev(self.ctx.handle, a, b, int(trans)))
        finally:
            self._free(a, b)

    def gemv(self, a, x, y, trans):
        ha, hx, hy = self._wrap(a), self._wrap(x), self._wrap(y)
        try:
            _lib.check(self.lib.mgpi_gemv_dev(self.ctx.handle, ha, hx, hy, int(trans)))
        finally:
            self._free(ha, hx, hy)

    def predict_mean(self, xs, x, terms, alpha):
        desc, count = _lib.make_terms(terms)
        XS = _lib.DeviceMatrix.from_host(self.ctx, xs)
        X = _lib.DeviceMatrix.from_host(self.ctx, x)
        A = _lib.DeviceMatrix.from_host(self.ctx, alpha)
        out = _lib.DeviceMatrix(self.ctx, len(xs), 1)
        _lib.check(self.lib.mgpi_predict_mean_dev(self.ctx.handle, XS.handle, X.handle, desc, count, A.handle, out.handle))
        return out.to_host().reshape(-1)

    def colsq(self, v):
        """column sums of squares of a local slab, as a numpy vector"""
        self.fence()
        return (v * v).sum(dim=0).cpu().numpy()

    def prior_diag(self, xs, terms):
        desc, count = _lib.make_terms(terms)
        xs = _lib.as_f64(xs, ndim=2)
        out = np.empty(len(xs))
        _lib.check(self.lib.mgpi_cov_paired(self.ctx.handle, xs.ctypes.data, xs.ctypes.data, len(xs), xs.shape[1], desc, count,
                                            out.ctypes.data))
        return out


class DistributedGP(object):
    """K(x,x; kernel+nugget) built, factored and solved on a P x Q grid of GPUs"""

    def __init__(self, grid, backend, source_x, kernel_terms, solve_terms, nb=512):
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
        self.A = None
        self.logdet = None
        self.alpha = None
        self.trailing_flops = 0.0   # algorithmic flops of this rank's trailing updates on the main stream (bench.py roofline)
        self.column_flops = 0.0     # ... and of its next-panel column updates on the look-ahead stream
        self.row_panels = None        # L[I][k] for the local block rows, every block step k (filled by factor)
        self._row_panel_offset = None
        self._dinv = None             # {k: -(L_kk)^-1} for the diagonal blocks this rank owns (variance sweep only)

    # -- index helpers ------------------------------------------------------------------------------------
    def _global_index(self, blocks):
        return (np.asarray(blocks)[:, None] * self.nb + np.arange(self.nb)[None, :]).reshape(-1)

    def _first_local(self, blocks, bound):
        """number of local blocks with global index <= bound"""
        return int(np.searchsorted(np.asarray(blocks), bound, side='right'))

    # -- communication ------------------------------------------------------------------------------------
    # No host synchronisation: with the HIP backend our stream IS torch's current stream, a (non-async) NCCL
    # collective waits for the current stream's earlier work and the current stream then waits for the
    # collective; gloo collectives are synchronous on the host.
    def _bcast_row(self, t, p, src_q):
        self.g.bcast_row(t, p, src_q)

    def _bcast_col(self, t, q, src_p):
        self.g.bcast_col(t, q, src_p)

    def _bcast_world(self, t, src):
        self.g.bcast_world(t, src)

    def _reduce_row(self, t, p, dst_q):
        self.g.reduce_row(t, p, dst_q)

    def _reduce_col(self, t, q, dst_p):
        self.g.reduce_col(t, q, dst_p)

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
            self.A = be.zeros(self.mloc, self.nloc)        # every entry is (re)written by the tile build below
        be.cov(xr, xc, self.solve_terms, self.A, nvr, nvc)
        # padded global indices (>= n) get a unit diagonal so the padded matrix stays positive definite
        for li, I in enumerate(self.row_blocks):
            if (I + 1) * nb > self.n and I % self.g.Q == self.g.q:
                lj = self.col_blocks.index(I)
                t0 = max(0, self.n - I * nb)
                self.A[li * nb + t0:(li + 1) * nb, lj * nb + t0:(lj + 1) * nb].diagonal().fill_(1.0)
        be.fence()

    # -- factor -------------------------------------------------------------------------------------------
    def _panel(self, k, be, bufs):
        """block step k, everything that is not the trailing update: factor the diagonal block, solve the panel
        below it, and move the panel to every rank that needs it (row panel `wrow` for the local block rows I > k,
        column panel `wcol` for the local block columns J > k).  Runs on whatever stream `be` enqueues to."""
        g, nb, A = self.g, self.nb, self.A
        pk, qk = k % g.P, k % g.Q
        kr, kc = k // g.P, k // g.Q                      # local block indices on the owners
        i0 = self._first_local(self.row_blocks, k)       # local row blocks [i0:) have I > k
        j0 = self._first_local(self.col_blocks, k)
        mt = self.mloc - i0 * nb                         # local trailing rows
        nt = self.nloc - j0 * nb
        info, ld = 0, 0.0
        lkk = bufs['lkk']
        # the row panel of step k lives in the persistent store (self.row_panels): the variance sweep reuses it, so
        # L crosses xGMI once, not twice
        off = self._row_panel_offset[k]
        wrow = self.row_panels[off:off + max(mt, 1)]
        wcol = bufs['wcol'][:max(nt, 1)]
        # 1. diagonal block, down its process column
        if g.q == qk:
            if g.p == pk:
                blk = A[kr * nb:(kr + 1) * nb, kc * nb:(kc + 1) * nb]
                info, ld = be.potrf(blk)
                if info > 0:
                    info = k * nb + info
                lkk.copy_(blk)
            self._bcast_col(lkk, qk, pk)
            # 2. panel trsm on this process column
            if mt > 0:
                be.trsm_right(lkk, A[i0 * nb:, kc * nb:(kc + 1) * nb])
        # 3. panel along process rows
        if mt > 0:
            if g.q == qk:
                wrow.copy_(A[i0 * nb:, kc * nb:(kc + 1) * nb])
            self._bcast_row(wrow, g.p, qk)
        # 4. transposed role along process columns.  The blocks L[J][k] this rank needs (J = q mod Q, J > k) sit,
        #    after step 3, in the row panels of the ranks (J mod P, q) of its own process column.  For a source
        #    grid row pr the wanted J form an arithmetic progression (J = r mod lcm(P,Q)), so gathering them on
        #    the source and scattering them into wcol on the receivers are single strided copies.
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
                buf = bufs['gather'][:count * nb]
                if g.p == pr:
                    il0 = (J0 - pr) // g.P - i0
                    buf.view(count, nb, nb).copy_(wrow.view(-1, nb, nb)[il0::si][:count])
                self._bcast_col(buf, g.q, pr)
                jl0 = (J0 - g.q) // g.Q - j0
                wcol.view(-1, nb, nb)[jl0::sj][:count].copy_(buf.view(count, nb, nb))
        return dict(wrow=wrow, wcol=wcol, i0=i0, j0=j0, mt=mt, nt=nt, info=info, logdiag=ld)

    def _update(self, pan, c_lo, c_hi, be):
        """trailing update of the local block columns [c_lo, c_hi) with the panel of one block step:
        A[I][J] -= L[I][k] L[J][k]^T for the local blocks with I >= J.  One launch (staircase gemm)."""
        g, nb = self.g, self.nb
        i0, j0 = pan['i0'], pan['j0']
        c_hi = min(c_hi, len(self.col_blocks))
        if pan['mt'] <= 0 or c_lo >= c_hi:
            return
        be.gemm_nt_stair(self.A[i0 * nb:, c_lo * nb:c_hi * nb], pan['wrow'], pan['wcol'][(c_lo - j0) * nb:(c_hi - j0) * nb],
                         g.P, g.p, g.Q, g.q, i0, c_lo, nb)
        flops = 0.0                                              # flop bookkeeping only (bench.py roofline)
        for jl in range(c_lo, c_hi):
            J = self.col_blocks[jl]
            il = self._first_local(self.row_blocks, J - 1)
            if il < len(self.row_blocks):
                rows = self.mloc - il * nb
                diag = nb if self.row_blocks[il] == J else 0      # a diagonal block counts as its lower half
                flops += 2.0 * (rows - diag) * nb * nb + diag * (nb + 1.0) * nb
        if be is self.be:
            self.trailing_flops += flops                         # launches timed by the main context's profile slot 0
        else:
            self.column_flops += flops                           # next-panel column updates on the look-ahead stream

    def factor(self, lookahead=None):
        """2-D block-cyclic right-looking Cholesky.  With look-ahead (default; MGPI_DIST_LOOKAHEAD=0 turns it off)
        the update of the block column of step k+1 and then step k+1's panel work - diagonal factorisation, panel
        solve and all its broadcasts - run on a second stream while the rest of the trailing update of step k runs on
        the main stream, so the xGMI transfers and the latency-bound panel hide behind the MFMA work.  Every rank
        issues the same collectives in the same order in both modes."""
        g, be, nb = self.g, self.be, self.nb
        self._dinv = None
        if lookahead is None:
            lookahead = os.environ.get('MGPI_DIST_LOOKAHEAD', '1') != '0'
        side = be.side_backend() if lookahead else be
        # panel buffers: two sets (step parity), allocated once on the main stream
        bufs = [dict(lkk=be.empty(nb, nb), wcol=be.empty(max(self.nloc, 1), nb), gather=be.empty(max(self.nloc, 1), nb))
                for _ in range(2)]
        # persistent row panels: for every block step k the blocks L[I][k] of the local block rows I > k
        # (sum_k (N - k nb)/P x nb doubles = 8 N^2 / (2P) bytes: 8.6 GB per rank at N = 65536 on 2 x 4)
        offs, total = [], 0
        for k in range(self.nblk):
            offs.append(total)
            total += max(self.mloc - self._first_local(self.row_blocks, k) * nb, 1)
        self._row_panel_offset = offs
        if self.row_panels is None or self.row_panels.shape[0] != total:
            self.row_panels = be.empty(total, nb)
        info_first, logdiag = 0, 0.0

        def note(pan):
            nonlocal info_first, logdiag
            if pan['info'] and not info_first:
                info_first = pan['info']
            logdiag += pan['logdiag']

        ncols = len(self.col_blocks)
        pan = self._panel(0, be, bufs[0])
        note(pan)
        for k in range(self.nblk):
            nxt = k + 1 < self.nblk
            j0 = pan['j0']
            own_next = nxt and g.q == (k + 1) % g.Q              # the first local trailing column is block column k+1
            if not lookahead or not nxt:
                self._update(pan, j0, ncols, be)
                if nxt:
                    pan = self._panel(k + 1, be, bufs[(k + 1) % 2])
                    note(pan)
                continue
            # Everything queued on the main stream so far (the rest of step k-1 included) is what the side stream has
            # to wait for.  (a) main stream: the rest of the trailing update of step k is QUEUED FIRST, because the
            # panel work below reads the diagonal block's status back and so blocks the host - with the opposite
            # order the main stream sat empty for the length of that round trip at every step (N = 40960 on a 1 x 1
            # grid: factor 427 ms, of which ~55 ms of exactly that).  (b) side stream (high priority): the block column
            # of the next panel, then that panel - its diagonal factorisation, panel solve and all of its broadcasts.
            # The two write disjoint block columns and both only read panel k.
            ready = be.record()
            self._update(pan, j0 + 1 if own_next else j0, ncols, be)
            with be.on_side(ready):
                if own_next:
                    self._update(pan, j0, j0 + 1, side)
                nxt_pan = self._panel(k + 1, side, bufs[(k + 1) % 2])
            be.join_side()
            note(nxt_pan)
            pan = nxt_pan
        # logdet and the first failing pivot, agreed by everyone
        if g.world > 1:
            s = self.be.from_numpy(np.array([logdiag]))
            g.all_reduce(s, 'sum')
            logdiag = float(self.be.to_numpy(s)[0])
            m = self.be.from_numpy(np.array([float(-info_first) if info_first else -1e300]))
            g.all_reduce(m, 'max')                            # max of (-info) = smallest failing pivot
            worst = float(self.be.to_numpy(m)[0])
            info_first = int(-worst) if worst > -1e299 else 0
        if info_first:
            raise _lib.NotPositiveDefinite(info_first, 'covariance is not positive definite! (pivot %d)' % info_first)
        self.logdet = 2.0 * logdiag
        return self.logdet

    # -- solve (replicated right-hand side) -----------------------------------------------------------------
    def solve(self, source_f):
        g, be, nb = self.g, self.be, self.nb
        A = self.A
        f = np.zeros(self.np_)
        f[:self.n] = _lib.as_f64(source_f).reshape(-1)
        y = be.from_numpy(f)                                   # replicated, global order
        ycol = be.zeros(self.nloc)                              # solved entries in local COLUMN order
        # forward: L y = f
        for k in range(self.nblk):
            pk, qk = k % g.P, k % g.Q
            kr, kc = k // g.P, k // g.Q
            if g.p == pk:
                s = be.zeros(nb)
                jc = self._first_local(self.col_blocks, k - 1)  # local col blocks [0:jc) have J < k
                if jc > 0:
                    be.gemv(A[kr * nb:(kr + 1) * nb, :jc * nb], ycol[:jc * nb], s, 0)     # s = -sum_J L[k][J] y_J
                self._reduce_row(s, pk, qk)
                if g.q == qk:
                    yk = y[k * nb:(k + 1) * nb]
                    yk.add_(s)
                    be.trsv(A[kr * nb:(kr + 1) * nb, kc * nb:(kc + 1) * nb], yk, 0)
            yk = y[k * nb:(k + 1) * nb]
            self._bcast_world(yk, g.rank_of(pk, qk))
            if g.q == qk:
                ycol[kc * nb:(kc + 1) * nb].copy_(yk)
        # backward: L^T x = y
        xrow = be.zeros(self.mloc)                              # solved entries in local ROW order
        for k in range(self.nblk - 1, -1, -1):
            pk, qk = k % g.P, k % g.Q
            kr, kc = k // g.P, k // g.Q
            if g.q == qk:
                s = be.zeros(nb)
                i0 = self._first_local(self.row_blocks, k)      # local row blocks [i0:) have I > k
                if i0 < len(self.row_blocks):
                    be.gemv(A[i0 * nb:, kc * nb:(kc + 1) * nb], xrow[i0 * nb:], s, 1)     # s = -sum_I L[I][k]^T x_I
                self._reduce_col(s, qk, pk)
                if g.p == pk:
                    xk = y[k * nb:(k + 1) * nb]
                    xk.add_(s)
                    be.trsv(A[kr * nb:(kr + 1) * nb, kc * nb:(kc + 1) * nb], xk, 1)
            xk = y[k * nb:(k + 1) * nb]
            self._bcast_world(xk, g.rank_of(pk, qk))
            if g.p == pk:
                xrow[kr * nb:(kr + 1) * nb].copy_(xk)
        self.alpha = be.to_numpy(y)[:self.n].copy()
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
        if g.world == 1:
            return part[:m]
        mine = be.from_numpy(part)
        parts = [be.zeros(per) for _ in range(g.world)]
        g.all_gather(parts, mine)
        return np.concatenate([be.to_numpy(t) for t in parts])[:m]

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
        out = be.empty(len(mine) * nb, nb)
        be.trtri_neg(self.A, (mine[0] // g.P) * nb, (mine[0] // g.Q) * nb, (lcm // g.P) * nb, (lcm // g.Q) * nb, nb, len(mine), out)
        for j, k in enumerate(mine):
            self._dinv[k] = out[j * nb:(j + 1) * nb]

    def variance(self, target_x):
        """pointwise posterior variance: V = L^-1 K(x, x*) by the same block-cyclic sweep (queries sharded over
        process columns, rows of V distributed like the rows of L), then column norms"""
        g, be, nb = self.g, self.be, self.nb
        A = self.A
        xs = _lib.as_f64(target_x, ndim=2)
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
        V = be.zeros(self.mloc, per)
        be.cov(xr, xq, self.kernel_terms, V, nvr, hi - lo)
        use_inv = os.environ.get('MGPI_DIST_VAR_DINV', '1') != '0'
        if use_inv:
            self._ensure_dinv()
        for k in range(self.nblk):
            pk, qk = k % g.P, k % g.Q
            kr, kc = k // g.P, k // g.Q
            i0 = self._first_local(self.row_blocks, k)
            mt = self.mloc - i0 * nb
            # diagonal block to the whole process row pk, which solves its slice of block row k of V.  With the
            # inverse diagonal blocks (default) what travels is -(L_kk)^-1 and the solve is one gemm instead of the
            # 8-step substitution chain, which would otherwise sit on the critical path of every block step
            vk = be.empty(nb, per)
            if g.p == pk:
                lkk = be.empty(nb, nb)
                if use_inv:
                    if g.q == qk:
                        lkk.copy_(self._dinv[k])
                    self._bcast_row(lkk, pk, qk)
                    solved = be.zeros(nb, per)
                    be.gemm_nn(solved, lkk, V[kr * nb:(kr + 1) * nb])            # 0 - (-(L_kk)^-1) V_k
                    V[kr * nb:(kr + 1) * nb].copy_(solved)
                else:
                    if g.q == qk:
                        lkk.copy_(A[kr * nb:(kr + 1) * nb, kc * nb:(kc + 1) * nb])
                    self._bcast_row(lkk, pk, qk)
                    be.trsm_left(lkk, V[kr * nb:(kr + 1) * nb])
                vk.copy_(V[kr * nb:(kr + 1) * nb])
            self._bcast_col(vk, g.q, pk)
            # panel blocks L[I][k], I > k: every rank kept its row panel of step k from the factorisation
            # (self.row_panels), so the factor is NOT broadcast a second time
            if mt > 0:
                off = self._row_panel_offset[k]
                be.gemm_nn(V[i0 * nb:], self.row_panels[off:off + mt], vk)
        sq = be.colsq(V)
        if g.P > 1:
            t = be.from_numpy(sq)
            g.all_reduce(t, 'sum', col=g.q)
            sq = be.to_numpy(t)
        var_q = np.zeros(per)
        if hi > lo:
            var_q[:hi - lo] = be.prior_diag(xs[lo:hi], self.kernel_terms) - sq[:hi - lo]
        if g.world == 1:
            return var_q[:m]
        mine = be.from_numpy(var_q)
        parts = [be.zeros(per) for _ in range(g.world)]
        g.all_gather(parts, mine)
        # ranks of one process column hold identical slices; take grid row 0
        out = np.concatenate([be.to_numpy(parts[g.rank_of(0, qq)]) for qq in range(g.Q)])
        return out[:m]
