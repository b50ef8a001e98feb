#!/usr/bin/env python3
"""bench.py - condition+predict throughput of the GP hot path on MI355X.

    python bench.py --gpus N --steps K --warmup W

A *step* is one pass of the hot path over one batch of synthetic input that is already resident in HBM:
    build K(x,x; kernel+nugget) lower tiles -> blocked Cholesky (nb=512) + alpha = K^-1 f (one call, mgpi_posv_dev)
    -> posterior mean at M queries (fused K* alpha) -> posterior variance at M queries (V = L^-1 K*^T, col norms)
i.e. Interpolator.condition_diag() without the host<->device copies (the PCIe-inclusive rate is in DESIGN.md).
Round 3: like condition_diag() itself, the step lets K* ride below K through the factorisation where that pays (N <= 16384 and
M <= N, e.g. --workload c2: one pass builds, factors, solves and forward-solves V^T = K* L^-T; mgpi_posv_riders_dev) - the same
arithmetic work, counted the same way.  The default workload (C3) keeps the two-pass form.

Workload at N=1 (config.workload): BASELINE.json configs[2] = C3, N=32768 D=3 jittered grid, SE + WhiteNoise
nugget, nb=512, M=4096 queries - the configuration north_star quotes its target on ("the Cholesky trailing
update at N=32768 on 1 MI355X").  `--workload c2` runs configs[1] (N=8192).

value = algorithmic fp64 flops per step / wall time (TFLOP/s), flops = N^3/3 (potrf) + 2N^2 (potrs) + N^2 M
(variance trsm) + 2MN (mean); kernel evaluations (exp) are NOT counted.  roofline = the dominant kernel, the
trailing-update syrk: algorithmic flops sum_k n_k(n_k+1) nb over its launches / HIP-event time of those
launches, recorded inside the timed region on the stream the kernel runs on; peak = 78.6 TFLOP/s (MI355X
fp64 matrix, spec).  check = what the last timed pass left in HBM, verified before anything overwrites it: the residual
|(K+nugget) alpha - f| / |f| on 1024 sampled rows (fused device contraction) and the range of the variance; the
script exits 3 when it fails.  cpu_baseline = the numpy oracle (a port of the reference's algorithm) on a bounded
sample of the same recipe, host cores of this box.

N>1: one process per GPU, RCCL over xGMI through the library's own C-ABI (mgpi_comm_*; no PyTorch anywhere).  The ranks
are started either by a launcher that exports RANK / LOCAL_RANK / WORLD_SIZE / MASTER_PORT (the driver's
`python -m torch.distributed.run ... bench.py`; only its environment is used) or, with `python bench.py --gpus N`
alone, by this script itself (N child processes, before anything touches the GPU).  Rank 0's RCCL unique id reaches
the others through a file (mgpi_amd/rendezvous.py).  ONE GP is factored across all ranks by the 2-D block-cyclic
right-looking Cholesky of muses-gaussian-process-interpolator_amd/distributed.py
(P x Q = 1x2, 2x2, 2x4; panel broadcasts along process rows and columns, local trailing updates on fp64 MFMA),
followed by the distributed solve, mean and variance.  Scaling is "weak" in the O(N^3) work: N = 32768 * G^(1/3)
rounded to the block grid (40960, 52224, 65536 = BASELINE.json configs[3]); value = whole-job algorithmic
flops / max-over-ranks wall time.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_FP64_MFMA_TFLOPS = 78.6     # MI355X fp64 matrix peak (spec)
NB = 512

WORKLOADS = {
    # name: (grid, D, M, seed)  - SURVEY.md section 8d recipes
    'c2': ((32, 16, 16), 3, 4096, 2),
    'c3': ((32, 32, 32), 3, 4096, 3),
    'c4': ((16, 16, 16, 16), 4, 4096, 4),   # BASELINE.json configs[3] on ONE GPU (34 GB factor; 288 GB HBM)
    # BASELINE.json configs[4]: stellarcollapse-shaped table, 1e6 queries, mean + variance (make_c5)
    'c5': (5, 3, 1000000, None),            # per-axis downsample 5  -> N = 13160
    'c5s': (10, 3, 1000000, None),          # per-axis downsample 10 -> N = 1680 (sc-eos-test's own setting)
}
# weak-scaling series for N > 1 GPU: N ~ 32768 * G^(1/3), a multiple of nb * lcm(P, Q)
MULTI = {
    2: ((32, 32, 40), 3, 4096, 3),          # N = 40960, grid 1 x 2
    4: ((32, 34, 48), 3, 4096, 3),          # N = 52224, grid 2 x 2
    8: ((16, 16, 16, 16), 4, 4096, 4),      # N = 65536, D = 4, grid 2 x 4  (BASELINE.json configs[3] = C4)
}


def make_inputs(n_side, m, seed):
    """jittered grid in [0,1]^3, f = sin/cos product, SE sigma=1, ell_d = 2*spacing_d, white sigma_n = 1e-2"""
    rng = np.random.default_rng(seed)
    axes = [(np.arange(s) + 0.5) / s for s in n_side]
    x = np.transpose([a.ravel() for a in np.meshgrid(*axes, indexing='ij')])
    x = x + rng.uniform(-0.1, 0.1, x.shape) / np.array(n_side)
    f = np.sin(3 * x[:, 0]) * np.cos(2 * x[:, 1]) + 0.5 * np.sin(4 * x[:, -1])
    xs = rng.random((m, len(n_side)))
    ell = tuple(2.0 / np.array(n_side, dtype=float))
    return x, f, xs, ell


def make_c5(ds, m_side=100):
    """BASELINE.json configs[4] on synthetic data of the stellarcollapse table's shape (the real
    LS180 table is fetched by wget in the reference, etc/stellarcollapse-eos/fetch:13-56, and is not available):
    axes ye(50) x logtemp(136) x logrho(234) (etc/example-stellarcollapse-data.ini:6-8, io.py:242-251), a smooth
    logenergy(ye, logT, logrho) monotone in logT, x = (ye, logrho, logenergy), f = logtemp; per-axis decimation by
    `ds` as test/stellarcollapse-eos/sc-eos-test:105-109; sigma = 2 std f, ell = 2 spacing, nugget sigma/100
    (sc-eos-test:132-148); queries = regular m_side^3 grid over the table's extent (sc-eos-test:172-176)."""
    ye = np.linspace(0.035, 0.53, 50)[::ds]
    lt = np.linspace(-2.0, 2.4, 136)[::ds]
    lr = np.linspace(3.0, 15.4, 234)[::ds]
    Y, T, R = np.meshgrid(ye, lt, lr, indexing='ij')
    loge = 17.0 + 1.1 * T + 0.25 * (R - 9.0) + 0.4 * Y + 0.05 * np.sin(R) * T
    x = np.transpose([Y.ravel(), R.ravel(), loge.ravel()])
    f = T.ravel() - np.mean(T)
    sigma = 2 * np.std(f)
    ell = tuple(2 * (np.ptp(x[:, k]) / (len(ax) - 1)) for k, ax in enumerate([ye, lr, lt]))
    q = [np.linspace(x[:, k].min(), x[:, k].max(), m_side) for k in range(3)]
    xs = np.transpose([a.ravel() for a in np.meshgrid(*q, indexing='ij')])
    return x, f, xs, ell, sigma, sigma / 100.0


def make_workload(name):
    """x, f, xs, ell, sigma, nugget sigma of a single-GPU workload"""
    spec = WORKLOADS[name]
    if name.startswith('c5'):
        return make_c5(spec[0])
    x, f, xs, ell = make_inputs(spec[0], spec[2], spec[3])
    return x, f, xs, ell, 1.0, 1e-2


def finish_check(resid, rows, var, mean, sigma):
    """the `check` object of the bench line: ok = residual < 1e-10, variance within [-1e-9, 1 + 1e-12] sigma^2, everything finite"""
    s2 = sigma * sigma
    ok = bool(np.isfinite(resid) and resid < 1e-10 and np.all(np.isfinite(var)) and np.all(np.isfinite(mean))
              and var.min() > -1e-9 * s2 and var.max() <= s2 * (1.0 + 1e-12))
    return {'residual': resid, 'residual_rows': int(rows), 'var_min': float(var.min()), 'var_max': float(var.max()),
            'prior_var': s2, 'mean_absmax': float(np.max(np.abs(mean))), 'ok': ok,
            'what': 'outputs of the last timed pass: |(K+nugget) alpha - f|_inf / |f|_inf on sampled source rows via the fused '
                    'device contraction; min / max of the posterior variance at the M queries'}


def step_flops(n, m):
    return n**3 / 3.0 + 2.0 * n * n + float(n) * n * m + 2.0 * m * n


def trailing_flops(n, nb=NB):
    """algorithmic lower-syrk flops of the trailing updates: sum_k n_k (n_k + 1) nb, n_k = N - k nb"""
    pn = (n + 127) // 128 * 128
    return float(sum((pn - k * nb) * (pn - k * nb + 1) * nb for k in range(1, (pn + nb - 1) // nb)))


def syrk_launch_flops(n, nb=NB):
    """algorithmic flops of the lower-syrk LAUNCHES of the single-level look-ahead factorisation: at step k the block
    column of panel k+1 is updated by its own gemm first, the syrk covers the remaining t2 = n_k - nb rows/columns:
    sum_k t2 (t2 + 1) nb.  (The column updates carry the rest of trailing_flops.)  The shipped schedule aggregates
    four panels per trailing update while the trailing matrix is large, so the timed run asks the library for the
    count it kept next to each launch (mgpi_ctx_get_profile_flops); this closed form stays as a cross-check bound."""
    pn = (n + 127) // 128 * 128
    total = 0.0
    for k in range(1, (pn + nb - 1) // nb):
        t2 = pn - k * nb - min(nb, pn - k * nb)
        total += t2 * (t2 + 1.0) * nb
    return total


def syrk_schedule_launches(n, nb=NB, panels=2, super_min=14336, merged_min=8192, tail_rows=2048):
    """the trailing-update launches of the shipped two-level schedule (csrc/chol.hip, super_width) as (rows, contraction)
    pairs: the very first update is one panel wide, then `panels` panels per update while at least `super_min` rows would
    remain behind the super-panel.  While at least `merged_min` rows lie behind the NEXT super-panel the update is one
    merged launch over the whole trailing matrix (next block column first, t rows); below that the block column is a gemm
    of its own on the look-ahead stream and the launch covers the rest, t2 rows.  Once at most `tail_rows` rows trail, one
    last update covers them all and the rest is factored as a single panel (no further gemm launches).
    merged_min=None, tail_rows=0: the round-1 schedule."""
    pn = (n + 127) // 128 * 128
    if pn <= tail_rows:
        return []

    def width(c0):
        left = pn - c0
        w = panels * nb if (c0 > 0 and left - panels * nb >= super_min) else nb
        return min(w, left)

    out, r0, wc = [], 0, width(0)
    while True:
        t = pn - r0 - wc
        if t <= 0:
            return out
        r1 = r0 + wc
        if t <= tail_rows and t % 128 == 0:
            out.append((t, wc))
            return out
        w1 = width(r1)
        t2 = t - w1
        if merged_min is not None and t2 > 0 and t2 >= merged_min and w1 % 128 == 0:
            out.append((t, wc))
        elif t2 > 0:
            out.append((t2, wc))
        r0, wc = r1, w1


def syrk_schedule_flops(n, **kw):
    """algorithmic lower-syrk flops of those launches, sum t (t + 1) wc: cross-check for the number the library reports
    (mgpi_ctx_get_profile_flops); tests/ compare the two on the GPU"""
    return float(sum(t * (t + 1.0) * wc for t, wc in syrk_schedule_launches(n, **kw)))


class GpuStep(object):
    """device-resident inputs + one hot-path pass through the C-ABI"""

    def __init__(self, x, f, xs, ell, sigma=1.0, nugget_sigma=1e-2):
        import ctypes
        import mgpi_amd as mgpi
        from mgpi_amd import _lib
        self.ct, self.lib_mod = ctypes, _lib
        self.ctx = mgpi.default_context()
        self.lib = self.ctx.lib
        self.n, self.m = len(x), len(xs)
        self.x_host, self.f_host, self.sigma = np.asarray(x), np.asarray(f), float(sigma)
        k = mgpi.SquaredExponentialKernel(sigma, *ell)
        nug = mgpi.WhiteNoiseKernel(nugget_sigma)
        self.kdesc = _lib.make_terms(k.terms())
        self.sdesc = _lib.make_terms(k.terms() + nug.terms())
        DM = _lib.DeviceMatrix
        self.X = DM.from_host(self.ctx, x)
        self.XS = DM.from_host(self.ctx, xs)
        self.F = DM.from_host(self.ctx, f)
        # condition_diag's form (gp.py, riders_pay): in the chain-bound regime (N <= 16384, no more queries than sources) K(x*, x)
        # rides below K through the factorisation (mgpi_cov_riders_dev / mgpi_posv_riders_dev) and the variance is read off the
        # rider rows; C3 / C4 and the million queries of C5 go through the finished factor (mgpi_predict_var_dev).
        from mgpi_amd.gp import riders_pay
        self.ride = riders_pay(self.n, self.m)              # the rule Interpolator.condition_diag applies
        if self.ride:
            pn = (self.n + 127) // 128 * 128
            self.KR = DM(self.ctx, pn + self.m, self.n, wide=True)
            hv = _lib._vp()
            _lib.check(self.lib.mgpi_mat_square_view(self.KR.handle, self.ct.byref(hv)))
            self.L = DM(self.ctx, self.n, self.n, handle=hv, parent=self.KR)
        else:
            self.L = DM(self.ctx, self.n, self.n, wide=True)
        self.alpha = DM(self.ctx, self.n, 1)
        self.mean = DM(self.ctx, self.m, 1)
        self.var = DM(self.ctx, self.m, 1)
        self.ctx.sync()

    def run(self):
        c, h, lib = self.lib_mod.check, self.ctx.handle, self.lib
        if self.ride:
            c(lib.mgpi_cov_riders_dev(h, self.X.handle, self.XS.handle, self.kdesc[0], self.kdesc[1], self.sdesc[0], self.sdesc[1],
                                      self.KR.handle))
            c(lib.mgpi_mat_copy(h, self.F.handle, self.alpha.handle))
            c(lib.mgpi_posv_riders_dev(h, self.KR.handle, self.alpha.handle, None))   # factor + solve + V^T = K* L^-T in one pass
            c(lib.mgpi_predict_mean_dev(h, self.XS.handle, self.X.handle, self.kdesc[0], self.kdesc[1],
                                        self.alpha.handle, self.mean.handle))
            c(lib.mgpi_riders_var_dev(h, self.KR.handle, self.XS.handle, self.kdesc[0], self.kdesc[1], self.var.handle))
            return
        c(lib.mgpi_cov_matrix_dev(h, self.X.handle, None, self.sdesc[0], self.sdesc[1], 1, self.L.handle))
        c(lib.mgpi_mat_copy(h, self.F.handle, self.alpha.handle))
        c(lib.mgpi_posv_dev(h, self.L.handle, self.alpha.handle, None))      # factor + solve: what Interpolator.fit() calls
        c(lib.mgpi_predict_mean_dev(h, self.XS.handle, self.X.handle, self.kdesc[0], self.kdesc[1],
                                    self.alpha.handle, self.mean.handle))
        c(lib.mgpi_predict_var_dev(h, self.XS.handle, self.X.handle, self.kdesc[0], self.kdesc[1], self.L.handle, 0,
                                   self.var.handle))

    def check(self, rows=1024):
        """what the LAST timed pass left in HBM, verified on the device: the residual |(K + nugget) alpha - f| / |f| (max norm)
        on `rows` evenly spaced source rows through the fused contraction (kernel + nugget evaluated at the sources
        themselves, never a stored K), and the range of the posterior variance."""
        DM, c = self.lib_mod.DeviceMatrix, self.lib_mod.check
        idx = np.unique(np.linspace(0, self.n - 1, min(rows, self.n)).astype(np.int64))
        XR = DM.from_host(self.ctx, self.x_host[idx])
        out = DM(self.ctx, len(idx), 1)
        c(self.lib.mgpi_predict_mean_dev(self.ctx.handle, XR.handle, self.X.handle, self.sdesc[0], self.sdesc[1], self.alpha.handle,
                                         out.handle))
        resid = float(np.max(np.abs(out.to_host().reshape(-1) - self.f_host[idx])) / np.max(np.abs(self.f_host)))
        return finish_check(resid, len(idx), self.var.to_host().reshape(-1), self.mean.to_host().reshape(-1), self.sigma)

    def run_staged(self):
        """one pass with a tic/toc pair around every stage (untimed extra of bench.py): ms per stage"""
        c, h, lib, ctx = self.lib_mod.check, self.ctx.handle, self.lib, self.ctx
        out = {}

        def stage(name, fn):
            ctx.tic()
            fn()
            out[name] = ctx.toc()
        if self.ride:
            stage('build_K_lower_and_Kstar', lambda: c(lib.mgpi_cov_riders_dev(h, self.X.handle, self.XS.handle, self.kdesc[0], self.kdesc[1],
                                                                                 self.sdesc[0], self.sdesc[1], self.KR.handle)))
            c(lib.mgpi_mat_copy(h, self.F.handle, self.alpha.handle))
            ctx.set_profile(True)
            # one call: factorisation + forward sweep of alpha beside it + the riders' forward solve inside it + backward sweep
            stage('posv_with_riders', lambda: c(lib.mgpi_posv_riders_dev(h, self.KR.handle, self.alpha.handle, None)))
            out['potrs'] = ctx.get_profile(6)[0]            # the backward sweep: what the solve still costs
            out['riders_update_launches'] = ctx.get_profile(4)[0]
            ctx.set_profile(False)
            out['potrf_and_variance_solve'] = out['posv_with_riders'] - out['potrs']
            stage('predict_mean', lambda: c(lib.mgpi_predict_mean_dev(h, self.XS.handle, self.X.handle, self.kdesc[0], self.kdesc[1],
                                                                      self.alpha.handle, self.mean.handle)))
            stage('variance_from_riders', lambda: c(lib.mgpi_riders_var_dev(h, self.KR.handle, self.XS.handle, self.kdesc[0], self.kdesc[1],
                                                                            self.var.handle)))
            return out
        stage('build_K_lower', lambda: c(lib.mgpi_cov_matrix_dev(h, self.X.handle, None, self.sdesc[0], self.sdesc[1], 1, self.L.handle)))
        # factor + solve is ONE call (mgpi_posv_dev): the forward sweep of the solve runs beside the factorisation, the backward
        # sweep after it.  'potrs' = that backward sweep (HIP-event pair inside the library, profile slot 6): what the solve
        # still costs; 'potrf' = the rest of the call (factorisation with the forward sweep in its shadow).
        c(lib.mgpi_mat_copy(h, self.F.handle, self.alpha.handle))
        ctx.set_profile(True)
        stage('posv', lambda: c(lib.mgpi_posv_dev(h, self.L.handle, self.alpha.handle, None)))
        out['potrs'] = ctx.get_profile(6)[0]
        ctx.set_profile(False)
        out['potrf'] = out['posv'] - out['potrs']
        stage('predict_mean', lambda: c(lib.mgpi_predict_mean_dev(h, self.XS.handle, self.X.handle, self.kdesc[0], self.kdesc[1],
                                                                  self.alpha.handle, self.mean.handle)))
        stage('predict_var', lambda: c(lib.mgpi_predict_var_dev(h, self.XS.handle, self.X.handle, self.kdesc[0], self.kdesc[1],
                                                                self.L.handle, 0, self.var.handle)))
        # for continuity with rounds 1-2: both sweeps as a call of their own on the finished factor
        c(lib.mgpi_mat_copy(h, self.F.handle, self.alpha.handle))
        stage('potrs_standalone_both_sweeps', lambda: c(lib.mgpi_potrs_dev(h, self.L.handle, self.alpha.handle)))
        return out


class DistributedStep(object):
    """one GP across all ranks: block-cyclic build + factor + solve + mean + variance (inputs replicated on the
    hosts, K / L / V distributed in HBM; RCCL communicators and every device operation through the C-ABI)"""

    def __init__(self, rank, world, x, f, xs, ell):
        import mgpi_amd as mgpi
        from mgpi_amd.distributed import DistributedGP, HipTileBackend, LocalTransport, ProcessGrid, RcclTransport
        from mgpi_amd.rendezvous import exchange_unique_id
        self.ctx = mgpi.default_context()
        if world > 1:
            uid = exchange_unique_id(rank, world, RcclTransport.new_unique_id)
            self.transport = RcclTransport(self.ctx, uid, rank, world)
        else:
            self.transport = LocalTransport()
        self.grid = ProcessGrid(self.transport)
        self.be = HipTileBackend(self.ctx)
        k = mgpi.SquaredExponentialKernel(1.0, *ell)
        nug = mgpi.WhiteNoiseKernel(1e-2)
        self.gp = DistributedGP(self.grid, self.be, x, k.terms(), (k + nug).terms(), nb=NB)
        self.f, self.xs = f, xs
        self.n, self.m = len(x), len(xs)

    def run(self):
        gp = self.gp
        if os.environ.get('MGPI_BENCH_PHASES') == '1':      # developer switch: a host sync and a wall-clock stamp per phase (stderr)
            t, out = time.perf_counter(), []
            for name, fn in (('build', gp.build), ('factor', lambda: gp.factor(source_f=self.f)), ('solve', lambda: gp.solve(self.f)),
                             ('mean', lambda: setattr(self, 'mean', gp.mean(self.xs))),
                             ('variance', lambda: setattr(self, 'var', gp.variance(self.xs)))):
                fn()
                self.be.sync()
                now = time.perf_counter()
                out.append('%s %.1f ms' % (name, (now - t) * 1e3))
                t = now
            sys.stderr.write('phases: ' + ', '.join(out) + '\n')
            return
        if os.environ.get('MGPI_BENCH_HOST_STAMPS') == '1':  # developer switch: host time per call, NO extra syncs (stderr)
            t = [time.perf_counter()]
            gp.build(); t.append(time.perf_counter())
            gp.factor(source_f=self.f); t.append(time.perf_counter())
            gp.solve(self.f); t.append(time.perf_counter())
            self.mean = gp.mean(self.xs); t.append(time.perf_counter())
            self.var = gp.variance(self.xs); t.append(time.perf_counter())
            sys.stderr.write('host: ' + ', '.join('%s %.1f' % (n, (b - a) * 1e3) for n, a, b in zip(('build', 'factor', 'solve', 'mean', 'variance'), t, t[1:])) + ' ms\n')
            return
        gp.build()
        gp.factor(source_f=self.f)                  # the forward substitution rides along, block step by block step
        gp.solve(self.f)                            # ... so this is the backward sweep
        self.mean = gp.mean(self.xs)
        self.var = gp.variance(self.xs)

    def check(self, rows=1024):
        """as GpuStep.check, with the replicated alpha the distributed solve returned (every rank computes the same numbers)"""
        gp = self.gp
        idx = np.unique(np.linspace(0, self.n - 1, min(rows, self.n)).astype(np.int64))
        ka = self.be.predict_mean(gp.x[idx], gp.x, gp.solve_terms, gp.alpha)
        resid = float(np.max(np.abs(ka - self.f[idx])) / np.max(np.abs(self.f)))
        return finish_check(resid, len(idx), np.asarray(self.var), np.asarray(self.mean), 1.0)

    def fence(self):
        """every rank has arrived and its streams are drained (the contract's barrier + device synchronise)"""
        self.be.sync()
        self.grid.barrier(self.be)
        self.be.sync()

    def max_over_ranks(self, value):
        buf = self.be.alloc(1, 1)
        self.be.upload(buf, np.array([float(value)]))
        self.grid.all_reduce(self.be, buf, 'max')
        return float(self.be.download(buf).reshape(-1)[0])

    def close(self):
        self.transport.close()


def cpu_baseline(m=4096, seed=7, budget_s=10.0):
    """numpy oracle ("port" of the reference algorithm: pair materialisation + LU inverse + left-associated products) on
    a bounded sample of the same recipe, timed the way BASELINE.md section 3 plans it: compress(source_x, source_f) and
    predict(target_x, source_x, alpha) separately, best of R, at N=8192 (16x16x32 grid, BASELINE.json configs[1]'s size)
    with the workload's own M.  R = 3 when one compress takes under `budget_s` seconds, else 2 (the leg stays near 30 s).
    The stage split of compress (kernel build ~ N^2 D, LU inverse ~ N^3) gives the labelled EXTRAPOLATION to N=32768,
    where the port cannot run (three 26 GB temporaries per kernel build)."""
    from oracle import mgpi_oracle as oracle
    n_side = (16, 16, 32)
    x, f, xs, ell = make_inputs(n_side, m, seed)
    kernel, nugget = [oracle.term_se(1.0, *ell)], [oracle.term_white(1e-2)]
    runs, alpha = [], None
    while True:
        st = {}
        t0 = time.perf_counter()
        alpha = oracle.compress(kernel, nugget, x, f, timings=st)
        st['compress'] = time.perf_counter() - t0
        runs.append(st)
        if len(runs) >= (3 if runs[0]['compress'] < budget_s else 2):
            break
    best = min(runs, key=lambda r: r['compress'])
    t_compress = best['compress']
    predicts = []
    for _ in range(3):
        t0 = time.perf_counter()
        oracle.predict(kernel, xs, x, alpha)
        predicts.append(time.perf_counter() - t0)
    t_predict = min(predicts)
    dt = t_compress + t_predict
    n = len(x)
    try:    # the threads numpy's BLAS actually runs (its build caps them), not just the host's core count
        from threadpoolctl import threadpool_info
        threads = max([p.get('num_threads', 1) for p in threadpool_info() if p.get('user_api') == 'blas'] or [1])
    except Exception:
        threads = os.cpu_count()
    cpu_model = ''
    try:
        with open('/proc/cpuinfo') as fh:
            cpu_model = next((line.split(':', 1)[1].strip() for line in fh if line.startswith('model name')), '')
    except OSError:
        pass
    flops = n**3 / 3.0 + 2.0 * n * n + 2.0 * m * n              # the GPU step's count for factor + solve + mean
    # labelled extrapolation to the bench's own N (C3): kernel build ~ N^2 D (same D), LU inverse ~ N^3, predict ~ M N D
    big = 32768
    r = big / float(n)
    extrap = best['x2cov'] * r**2 + best['inv'] * r**3 + (t_compress - best['x2cov'] - best['inv']) * r**2 + t_predict * r
    return {'value': flops / dt / 1e12, 'unit': 'TFLOP/s', 'cores': threads, 'host_cores': os.cpu_count(), 'cpu_model': cpu_model,
            'kind': 'port',
            'sample': 'oracle.compress + oracle.predict (numpy pair materialisation + LU inverse) at N=%d D=3 M=%d, same recipe; best '
                      'of %d compress / 3 predict runs: %.1f s; algorithmic flops counted as for the GPU step (Cholesky count '
                      'N^3/3, although the port inverts by LU like the reference)' % (n, m, len(runs), dt),
            'seconds': dt, 'compress_seconds': t_compress, 'predict_seconds': t_predict, 'repeats': len(runs),
            'compress_stage_seconds': {'x2cov': best['x2cov'], 'inv': best['inv']},
            'all_compress_seconds': [r_['compress'] for r_ in runs],
            'extrapolated': {'N': big, 'M': m, 'seconds': extrap,
                             'label': 'EXTRAPOLATION, not a measurement: x2cov * (N/8192)^2 + inv * (N/8192)^3 + predict * (N/8192); '
                                      'the port cannot run at this size (pair temporaries exceed host memory)'}}


def load_traffic(workload):
    """HBM bytes per trailing-update launch from the committed rocprofv3 --pmc summary (profiles/), or None"""
    path = os.path.join(ROOT, 'profiles', 'traffic.json')
    try:
        with open(path) as fh:
            return json.load(fh).get(workload, {}).get('bytes_per_launch')
    except (OSError, ValueError):
        return None


def visible_gpus():
    """number of HIP devices, counted in a CHILD process: the parent of self-launched ranks must never touch the GPU"""
    import subprocess
    code = 'import sys; sys.path.insert(0, %r); from mgpi_amd import _lib; print(_lib.device_count())' % ROOT
    out = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True, timeout=600)
    try:
        return int(out.stdout.strip().splitlines()[-1])
    except (ValueError, IndexError):
        sys.stderr.write(out.stderr[-2000:])
        return 0


def self_launch(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks ourselves, before any GPU call in this process"""
    from mgpi_amd.rendezvous import spawn_ranks
    have = visible_gpus()
    if args.gpus > have:
        sys.stderr.write('bench.py: --gpus %d requested but %d HIP device(s) are visible\n' % (args.gpus, have))
        return 2
    # every rank is watched: the first one that dies ends the job with its stderr, and nothing outlives the deadline
    codes = spawn_ranks([os.path.abspath(__file__)] + sys.argv[1:], args.gpus, timeout=float(os.environ.get('MGPI_LAUNCH_TIMEOUT', '1500')))
    bad = [(r, c) for r, c in enumerate(codes) if c != 0]
    if bad:
        sys.stderr.write('bench.py: ranks failed (rank, exit code): %s\n' % bad)
        return 1
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=1)
    ap.add_argument('--workload', default='c3', choices=sorted(WORKLOADS))
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    from mgpi_amd.rendezvous import rank_env
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args))
    rank, local_rank, world = rank_env()
    if world != args.gpus:
        sys.stderr.write('bench.py: --gpus %d but the launcher started %d rank(s)\n' % (args.gpus, world))
        sys.exit(2)
    os.environ.setdefault('MGPI_DEVICE', str(local_rank))

    # developer switch: run the multi-GPU code path on a 1 x 1 grid (one GPU) to exercise its plumbing
    force_dist = world == 1 and os.environ.get('MGPI_BENCH_FORCE_DIST') == '1'
    distributed = world > 1 or force_dist

    if not distributed:
        d, m = WORKLOADS[args.workload][1], WORKLOADS[args.workload][2]
        x, f, xs, ell, sigma, nugget_sigma = make_workload(args.workload)
        name, par = args.workload.upper(), 'single GPU'
        if os.environ.get('MGPI_BENCH_SIDE'):      # developer switch: the same recipe on another grid (A/B against the 1 x 1 driver)
            n_side = tuple(int(v) for v in os.environ['MGPI_BENCH_SIDE'].split('x'))
            d = len(n_side)
            x, f, xs, ell = make_inputs(n_side, m, 3)
            sigma, nugget_sigma, name = 1.0, 1e-2, 'custom grid %s' % os.environ['MGPI_BENCH_SIDE']
        step = GpuStep(x, f, xs, ell, sigma, nugget_sigma)
    else:
        if not force_dist and world not in MULTI:
            sys.stderr.write('bench.py: supported GPU counts: 1, %s\n' % sorted(MULTI))
            sys.exit(2)
        n_side, d, m, seed = MULTI[2 if force_dist else world]
        if force_dist and os.environ.get('MGPI_BENCH_SIDE'):
            n_side = tuple(int(v) for v in os.environ['MGPI_BENCH_SIDE'].split('x'))
            d = len(n_side)
        x, f, xs, ell = make_inputs(n_side, m, seed)           # the same seed on every rank: replicated inputs
        step = DistributedStep(rank, world, x, f, xs, ell)
        name = 'C4' if world == 8 else 'C3 scaled by %d^(1/3)' % world
        par = '2-D block-cyclic %dx%d, nb=%d, RCCL panel broadcasts over xGMI' % (step.grid.P, step.grid.Q, NB)
    n = len(x)
    ctx = step.ctx

    def fence():
        if distributed:
            step.fence()
        else:
            ctx.sync()

    c5 = (not distributed) and args.workload.startswith('c5')
    slot = 4 if c5 else 0      # the dominant kernel: NN update gemm of the variance solve (C5) / lower syrk of the factorisation
    for _ in range(args.warmup):
        step.run()
    # HIP-event pairs around the DOMINANT kernel's launches only (and the one predict-mean launch, whose pair count rides in
    # slot 5): every recorded event is a marker between two kernels of its stream, 5 - 9 us each on a latency chain; the
    # other stage timers run in an untimed pass below.  No host sync.
    ctx.set_profile(True, slots=(slot, 5))
    fence()
    t0 = time.perf_counter()
    per_step = os.environ.get('MGPI_BENCH_PER_STEP') == '1'     # developer switch: a fence and a wall-clock stamp after every step (stderr)
    for _ in range(args.steps):
        ts = time.perf_counter()
        step.run()
        if per_step:
            if os.environ.get('MGPI_BENCH_PER_STEP_FENCE', '1') != '0':
                fence()
            sys.stderr.write('step %.1f ms\n' % ((time.perf_counter() - ts) * 1e3))
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        elapsed = step.max_over_ranks(elapsed)

    # untimed extras: (a) the same kernel with the look-ahead off, i.e. without a panel factorisation running beside it,
    # (b) one pass with a HIP-event pair around every stage
    iso, stages = None, None
    if not distributed:
        dom_ms, dom_launches = ctx.get_profile(slot)
        timed_flops = ctx.get_profile_flops(slot)
        timed_pairs = ctx.get_profile_flops(5)
        pred_ms, _ = ctx.get_profile(5)         # the fused predict-mean launches of the timed passes
        ctx.set_profile(False)
        check = step.check()       # the outputs of the last timed pass, before the untimed extras overwrite them
        ctx.set_profile(True)      # (c) one pass like the timed ones with every stage timer on
        step.run()
        ctx.sync()
        timed = [ctx.get_profile(i) for i in range(6)]
        ctx.set_lookahead(False)
        ctx.set_profile(True)
        step.run()
        ctx.sync()
        iso_ms, iso_launches = ctx.get_profile(slot)
        iso_flops = ctx.get_profile_flops(slot)
        ctx.set_lookahead(True)
        ctx.set_profile(False)
        iso = iso_flops / (iso_ms * 1e-3) / 1e12 if iso_ms > 0 else None
        stages = step.run_staged()
        (syrk_ms, syrk_launches), (panel_ms, _), (potrf_ms, _), (colupd_ms, colupd_launches) = timed[:4]
        stage_steps = 1            # the stage timers cover ONE (untimed) pass
    else:
        dom_ms, dom_launches = ctx.get_profile(0)
        syrk_ms, panel_ms, potrf_ms, colupd_ms, stage_steps = dom_ms, 0.0, 0.0, 0.0, args.steps
        ctx.set_profile(False)
        check = step.check()
    ctx.set_profile(False)

    if rank == 0:
        flops = step_flops(n, m)                                  # one GP of size n, whatever the GPU count
        value = flops * args.steps / elapsed / 1e12
        if not distributed:
            tr_flops = timed_flops / args.steps                    # reported by the library next to each launch
        else:
            tr_flops = step.gp.trailing_flops / (args.steps + args.warmup)      # this rank's share, per step
        tflops_dom = tr_flops * args.steps / (dom_ms * 1e-3) / 1e12
        if c5:
            kernel_name = 'gemm_f64_pipe_kernel<NN> (update launches of V = L^-1 K*^T, the variance solve; main and look-ahead stream)'
            data_desc = 'stellarcollapse-shaped table (ye x logrho x logenergy -> logtemp, per-axis downsample %d)' % WORKLOADS[args.workload][0]
        else:
            kernel_name = 'gemm_f64_pipe_kernel<NT%s> (Cholesky trailing update%s)' % ((',lower', ', syrk') if not distributed else ('', ', rank 0 local blocks'))
            data_desc = 'jittered grid'
        out = {
            'metric': 'condition+predict achieved fp64 TFLOP/s (wall-clock in ms_per_step) at N training points',
            'value': value, 'unit': 'TFLOP/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True, 'scaling': 'weak',
            'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': '%s: N=%d D=%d %s, SquaredExponential + WhiteNoise nugget, blocked '
                                   'Cholesky nb=%d, M=%d queries (mean+variance)' % (name, n, d, data_desc, NB, m),
                       'N': n, 'D': d, 'M': m, 'nb': NB, 'parallelism': par},
            'roofline': {'bound': 'mfma', 'achieved': tflops_dom, 'peak': PEAK_FP64_MFMA_TFLOPS, 'unit': 'TFLOP/s',
                         'frac': tflops_dom / PEAK_FP64_MFMA_TFLOPS,
                         'traffic': load_traffic(args.workload) if not distributed else None,
                         'kernel': kernel_name,
                         'launches': int(dom_launches), 'avg_launch_ms': dom_ms / max(1, dom_launches),
                         'algorithmic_flops_per_step': tr_flops,
                         'note': 'timed region runs with look-ahead: the next block\'s update and solve/factorisation (same gemm '
                                 'family, other stream) share the GPU with this kernel; isolated_* = the same launches with the '
                                 'look-ahead off',
                         'isolated_achieved': iso, 'isolated_frac': (iso / PEAK_FP64_MFMA_TFLOPS) if iso else None},
            'stages_ms_per_step': {'potrf': potrf_ms / stage_steps, 'potrf_trailing_update': syrk_ms / stage_steps,
                                   'potrf_next_panel_column_update': colupd_ms / stage_steps,
                                   'potrf_panels_not_overlapped': panel_ms / stage_steps,
                                   'note': 'one untimed pass with every stage timer on; the timed region records events around the '
                                           'dominant kernel only (roofline.launches / avg_launch_ms)'},
        }
        if stages is not None:
            out['stages_ms_one_pass'] = stages
            # stage-level rates of that one pass: every launch of the stage, both streams, against its algorithmic flops
            if 'potrf' in stages:
                out['stage_tflops_one_pass'] = {'potrf': n**3 / 3.0 / (stages['potrf'] * 1e-3) / 1e12,
                                                'predict_var': float(n) * n * m / (stages['predict_var'] * 1e-3) / 1e12}
            else:       # riders: the variance solve's N^2 M flops run inside the factorisation's launches
                out['stage_tflops_one_pass'] = {'potrf_and_variance_solve': (n**3 / 3.0 + float(n) * n * m) /
                                                (stages['potrf_and_variance_solve'] * 1e-3) / 1e12}
            out['predict_mean'] = {'pairs_per_s': timed_pairs / (pred_ms * 1e-3) if pred_ms > 0 else None,
                                   'ms_per_step': pred_ms / args.steps}
            # the backward sweep of the single-right-hand-side solve against the HBM roofline: the factor's lower triangle once
            if stages.get('potrs'):
                gbs = 8.0 * n * n / 2.0 / (stages['potrs'] * 1e-3) / 1e9
                out['stage_hbm_one_pass'] = {'potrs_backward_sweep': {'GB/s': gbs, 'frac_of_8_TB/s': gbs / 8000.0,
                                                                     'algorithmic_bytes': 8.0 * n * n / 2.0}}
        out['check'] = check
        if world == 1:
            # what THIS box's GPU sustains on the fp64 matrix pipe right now (the boxes of one pool differ by a few per cent);
            # the roofline above stays on the 78.6 TFLOP/s spec peak
            try:
                out['device_probe'] = {'mfma_f64_tflops': ctx.probe_mfma_f64(),
                                       'what': 'back-to-back v_mfma_f64_16x16x4, random operands, 2 x 4 waves per CU, ~0.1 s, after the timed region'}
            except Exception as err:      # a diagnostic, never a reason to lose the line
                out['device_probe'] = {'error': str(err)}
        if world == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(m=min(m, 4096))
        print(json.dumps(out), flush=True)
    if distributed:
        step.fence()
        step.close()
    if not check['ok']:
        sys.stderr.write('bench.py: the timed outputs FAILED their check: %s\n' % json.dumps(check))
        sys.exit(3)


if __name__ == '__main__':
    main()
