// Blocked right-looking Cholesky  K = L L^T  (lower, row-major, in place in HBM) and the triangular solves
// built on it.  Takes the role of np.linalg.inv / np.linalg.slogdet in the reference
// (mgpi/interpolators.py:92,179,339,343): alpha = L^-T L^-1 f, logdet = 2 sum log L_ii.
//
// Two-level blocking: outer block nb (ctx->nb, default 512) for the trailing update on fp64 MFMA (gemm.hip),
// inner block 64 for the panel: one launch per 64-column step (panel_step_kernel: deferred right-looking update, 64 x 64
// Cholesky and 64-row substitution, all in MFMA accumulator layout, panel_device.h).  Look-ahead of depth one: panel k+1
// is factored on a second stream while the trailing update of panel k runs.
#include "mgpi_internal.h"
#include <math.h>
#include <vector>

#ifdef PANEL_TRACE   // diagnostic build only
#define PANEL_STAMPS 1
#include <hip/hip_runtime.h>
__device__ long long panel_stamps[8];
extern "C" int mgpi_debug_read_stamps(long long* out) {
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(panel_stamps), sizeof(long long) * 8) == hipSuccess ? 0 : -1;
}
#endif
#include "panel_device.h"

// Early returns (MGPI_TRY / MGPI_HIP_TRY) inside the look-ahead sections must not leave work running on the second stream
// - or the ctx pointing at it - when the caller gets its error status back: armed on entry, disarmed on the success path.
struct mgpi_stream_join_guard {
    mgpi_ctx* c;
    hipStream_t main;
    bool armed;
    ~mgpi_stream_join_guard() {
        if (!armed) return;
        c->stream = main;
        c->lean = 0;
        c->gemm_zero_c = 0;
        if (c->stream2) (void)hipStreamSynchronize(c->stream2);
        if (c->stream3) (void)hipStreamSynchronize(c->stream3);
        (void)hipStreamSynchronize(main);
    }
};

// single-rhs sweeps (defined further down); `diag` says where the 64 x 64 DIAGONAL tiles of a block are read from - during a
// factorisation they still wait in the scratch strip (see panel_step_kernel), afterwards they are in the matrix
struct trsv_diag_src {
    const double* base;     // element (r, c) of the diagonal tile that starts at row/column c0 of the block: base[(c0 + r) * rs + cm * c0 + c]
    int64_t rs;
    int cm;
};
static void launch_trsv_diag(hipStream_t stream, const double* L, int64_t lda, double* v, int w, int trans, const trsv_diag_src* diag = nullptr);
static void launch_trsv_update(hipStream_t stream, const double* L, int64_t lda, double* v, int64_t r0, int w, int64_t first,
                               int64_t count, int trans, const double* warm = nullptr, int warm_w = 0);
// A (m rows x w columns, row stride lda) <- A L^-T for the factored w x w diagonal block L (row stride ldl); strip: where the
// 64 x 64 diagonal tiles of L are read from during a factorisation (row r of the block at strip + 64 r), or nullptr (in L)
static int launch_trsm_right_steps(hipStream_t stream, double* A, int64_t lda, int64_t m, const double* L, int64_t ldl, int64_t w,
                                   const double* strip);

// ---- ONE launch per 64-column step of a panel (round 2) -----------------------------------------------------------------
// Step j of a panel (diagonal block at row / column c0) used to be three launches on the chain the trailing update waits
// for: potf2 (or potf2+trsm fused), the row trsm, and the right-looking gemm that applies the 64 new columns to the rest
// of the panel.  Here the gemm is DEFERRED by one step and split by who needs it, and everything runs in one grid:
//   * workgroup b < nfused ("column role"): row block b at/below the diagonal block.  Applies the previous step's columns
//     X' (rows >= c0, columns [c0 - 64, c0)) to ITS rows of column block [c0, c0 + 64) and - redundantly, every workgroup -
//     to the diagonal block, factors that 64 x 64 block, solves its rows against it, stores them.  The products land in
//     MFMA accumulators, which is also the layout the factorisation and the substitution run in (panel_device.h).
//     Workgroup 0 owns the diagonal block's rows: it stores the factor (to the scratch strip S, NOT in place: late
//     workgroups of this launch still read the unfactored block; diag_blocks_back_kernel copies the strip over the
//     diagonal blocks when the whole factorisation is done) and reports a failed pivot.
//   * workgroups >= nfused ("update role"): one 64 x 64 tile each of the REST of the panel (column blocks right of this
//     step's), to which they apply X' - the part of the deferred gemm that is not on this step's critical path.
// The chain per step is then one launch of (two contraction-64 products + 64 x 64 Cholesky + 64-row substitution).
// launch_bounds(256, 2): with room for only one workgroup per CU the compiler parks the MFMA accumulators in AGPRs, where
// v_mfma_f64 issues at less than half the rate (DESIGN.md section c).
// Measured and dropped: folding the PREVIOUS panel's trailing update (contraction 512) into the column role, to take the
// skinny column-update gemm off the chain - with operands streamed straight from L2 the 64 x 64 x 512 products per
// workgroup made a step 2x slower (N = 8192: potrf 13.8 ms against 7.5).
// Also measured and dropped (round 2, removed in round 3): left-looking steps for large panels, the factorisation's pivot
// recurrence on the matrix pipe (v_mfma_f64_4x4x4), the substitution on the vector pipe, the three-launch round-1 kernels.
// `mode` splits the work for LARGE panels, where hundreds of workgroups repeating the 64 x 64 factorisation would take CU
// time from the trailing update that runs beside the panel:
//   mode 0  everything in one launch (every column-role workgroup factors the diagonal block itself)
//   mode 1  one workgroup: update + factor the diagonal block, store it (strip S) and its LDS images (Lt, 1/diag) to F;
//           the row blocks BELOW the diagonal block and the update-role tiles then run in panel_units_kernel, which reads
//           the images from F instead of factoring ([r3]; the one-unit-per-workgroup form of that launch, "mode 2", is gone)
#define PANEL_F_DOUBLES (IB * IB + IB)
#ifdef PANEL_TRACE   // diagnostic build only (tools/contention_probe.py): per-workgroup phase stamps of the step kernels
__device__ unsigned long long g_panel_trace[1 << 18];
__device__ unsigned g_panel_trace_n;
extern "C" int mgpi_debug_read_trace(unsigned long long* host, unsigned* n) {
    if (hipMemcpyFromSymbol(n, HIP_SYMBOL(g_panel_trace_n), sizeof(unsigned)) != hipSuccess) return -1;
    if (hipMemcpyFromSymbol(host, HIP_SYMBOL(g_panel_trace), sizeof(unsigned long long) * (1 << 18)) != hipSuccess) return -1;
    unsigned zero = 0;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_panel_trace_n), &zero, sizeof(unsigned)) != hipSuccess) return -1;
    return 0;
}
#define PT_STAMP(k) do { if (threadIdx.x == 0) pt[k] = __builtin_amdgcn_s_memrealtime(); } while (0)
#define PT_FLUSH(role) do { if (threadIdx.x == 0) { const unsigned slot = atomicAdd(&g_panel_trace_n, 1u); if (slot < (1u << 18) / 8) { \
    unsigned long long* o = g_panel_trace + 8 * slot; o[0] = (unsigned long long)c0; o[1] = ((unsigned long long)(role) << 32) | blockIdx.x; \
    o[2] = pt[0]; o[3] = pt[1]; o[4] = pt[2]; o[5] = pt[3]; o[6] = nfused; o[7] = __builtin_amdgcn_s_memrealtime(); } } } while (0)
#else
#define PT_STAMP(k) do { } while (0)
#define PT_FLUSH(role) do { } while (0)
#endif
__global__ __launch_bounds__(256, 2) void panel_step_kernel(double* __restrict__ A, int64_t lda, int64_t c0, int has_prev, int nfused,
                                                         int rest_cols, int64_t pivot_base, const double* __restrict__ diag0,
                                                         int* __restrict__ info, double* __restrict__ S, double ptol, int mode,
                                                         double* __restrict__ F, int64_t strideA, int64_t strideS, int64_t strideD) {
    __shared__ panel_lds s;
    MGPI_CHAIN_PRIO();
#ifdef PANEL_TRACE
    unsigned long long pt[4] = {0, 0, 0, 0};
#endif
    PT_STAMP(0);
    // strided batch (mgpi_loglikelihood_batch): problem y = blockIdx.y has its own matrix, scratch strip, original diagonal
    // and status word; single problems pass a 1-deep grid
    A += (int64_t)blockIdx.y * strideA;
    S += (int64_t)blockIdx.y * strideS;
    diag0 += (int64_t)blockIdx.y * strideD;
    info += blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    // the wave index in a scalar register: branches on it become scalar branches instead of EXEC-masked regions whose merges
    // copy every accumulator (60 v_mov_b64 per round - vector instructions the starved issue stage pays for one by one)
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t cp = c0 - IB, c1 = c0 + IB;
    panel_double4_t acc[4];
    if ((int)blockIdx.x >= nfused) {
        // update role: tile (ti, tj) of the rows >= c1, column blocks c1 + 64 tj (tj < rest_cols); only tiles on/below the diagonal
        int ti, tj;
        panel_rest_decode((int)blockIdx.x - nfused, nfused - 1, ti, tj);
        double* C = A + (c1 + (int64_t)IB * ti) * lda + c1 + (int64_t)IB * tj;
        panel_load(acc, C, lda, w, g, c, false);
        panel_gemm_nt(acc, A + (c1 + (int64_t)IB * ti + 16 * w + c) * lda + cp + 2 * g, A + (c1 + (int64_t)IB * tj + c) * lda + cp + 2 * g, lda, 0, 4);
        PT_STAMP(1);
        panel_store(acc, C, lda, w, g, c);
        PT_FLUSH(2);
        return;
    }
    const int b = (int)blockIdx.x;
    const int64_t r0 = c0 + (int64_t)IB * b;                      // this workgroup's rows
    if (b > 0) {
        panel_load(acc, A + r0 * lda + c0, lda, w, g, c, false);
        if (has_prev) panel_gemm_nt(acc, A + (r0 + 16 * w + c) * lda + cp + 2 * g, A + (c0 + c) * lda + cp + 2 * g, lda, 0, 4);
    }
    {
        if (tid < IB) s.dtol[tid] = ptol * fabs(diag0[tid]);
        // the diagonal block, current as of the previous step
        panel_double4_t accD[4];
        panel_load(accD, A + c0 * lda + c0, lda, w, g, c, true);
        if (has_prev) panel_gemm_nt(accD, A + (c0 + 16 * w + c) * lda + cp + 2 * g, A + (c0 + c) * lda + cp + 2 * g, lda, 0, w + 1);
        int bad = 0;
        PT_STAMP(1);
        potf2_acc_rounds<0>::run(accD, s, w, g, c, bad, b == 0 ? S : nullptr);
        PT_STAMP(2);
        if (b == 0) {
            if (tid == 0 && bad != 0) {
                const int cand = (int)(pivot_base + bad);
                int old = atomicCAS(info, 0, cand);
                while (old != 0 && old > cand) {
                    const int prev = atomicCAS(info, old, cand);
                    if (prev == old) break;
                    old = prev;
                }
            }
            if (mode == 1) {
                __syncthreads();
                for (int i = tid; i < IB * IB; i += 256) F[i] = (&s.Lt[0][0])[i];
                if (tid < IB) F[IB * IB + tid] = s.invd[tid];
            }
            PT_FLUSH(0);
            return;
        }
    }
    __syncthreads();                                               // the factor's LDS images are complete
    panel_trsm_images(s, tid);
    trsm_acc_mx<0>::run(acc, s, w, g, c, A + (r0 + 16 * w + c) * lda + c0);
    PT_STAMP(3);
    PT_FLUSH(1);
}

// [r3] The rows launch of a SPLIT step (what follows a mode-1 launch of panel_step_kernel) with NU units of work per workgroup, as a kernel of its
// own (panel_step_kernel keeps its register budget): rows != 0: row blocks 1 + NU p .. NU + NU p below the diagonal block, their
// substitution chains interleaved (trsm_acc_mxn), the factor's images from F; rows == 0: update-role tiles NU p .. NU p + NU - 1
// (ntiles in all).  A unit index beyond the last one is clamped to it: the same loads, the same arithmetic, the same values
// stored twice by the same workgroup.  The split steps run beside the trailing update of a large matrix: 1/NU of the
// workgroups and image loads, and less time in which these latency-bound workgroups keep CU slots from the big kernel.
// Measured (bench.py C3, same box, NU = 1 / 2 / 4): 258.8 / 258.2 / 258.7 ms per step, the trailing update's in-region time
// 177.9 / 176.1 / 176.0 ms (0.799 / 0.807 / 0.808 of peak) - four units spill (256 VGPRs + 21), two ship; C4 1669.5 -> 1664.2 ms.
template <int NU>
__global__ __launch_bounds__(256, 2) void panel_units_kernel(double* __restrict__ A, int64_t lda, int64_t c0, int has_prev, int nfused,
                                                           int rows, const double* __restrict__ F, int ntiles) {
    __shared__ panel_lds s;
    MGPI_CHAIN_PRIO();
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t cp = c0 - IB, c1 = c0 + IB;
    panel_double4_t acc[NU][4];
    if (!rows) {
        double* C[NU];
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            int t = NU * (int)blockIdx.x + u, ti, tj;
            t = t < ntiles ? t : ntiles - 1;
            panel_rest_decode(t, nfused - 1, ti, tj);
            C[u] = A + (c1 + (int64_t)IB * ti) * lda + c1 + (int64_t)IB * tj;
            panel_load(acc[u], C[u], lda, w, g, c, false);
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) {
            const int64_t ro = (C[u] - A) / lda, co = (C[u] - A) % lda;          // the tile's first row / column
            panel_gemm_nt(acc[u], A + (ro + 16 * w + c) * lda + cp + 2 * g, A + (co + c) * lda + cp + 2 * g, lda, 0, 4);
        }
#pragma unroll
        for (int u = 0; u < NU; ++u) panel_store(acc[u], C[u], lda, w, g, c);
        return;
    }
    double* out[NU];
#pragma unroll
    for (int u = 0; u < NU; ++u) {
        int b = 1 + NU * (int)blockIdx.x + u;
        b = b < nfused ? b : nfused - 1;
        const int64_t r = c0 + (int64_t)IB * b;
        panel_load(acc[u], A + r * lda + c0, lda, w, g, c, false);
        out[u] = A + (r + 16 * w + c) * lda + c0;
    }
    for (int i = tid; i < IB * IB; i += 256) (&s.Lt[0][0])[i] = F[i];
    if (tid < IB) s.invd[tid] = F[IB * IB + tid];
    if (has_prev) {
#pragma unroll
        for (int u = 0; u < NU; ++u) panel_gemm_nt(acc[u], out[u] - c0 + cp + 2 * g, A + (c0 + c) * lda + cp + 2 * g, lda, 0, 4);
    }
    __syncthreads();
    panel_trsm_images(s, tid);
    trsm_acc_mxn<0, NU>::run(acc, s, w, g, c, out);
}

// one step: diagonal block at (c0, c0) of the matrix A (c0 absolute), `rows_below` rows below the block, the panel ends at
// column panel_end; has_prev: the columns [c0 - 64, c0) of this panel were solved by the previous step and still have to be
// applied to everything right of them.  split: two launches (diagonal block, then the rows) instead of one
static int launch_panel_step(mgpi_ctx* ctx, double* A, int64_t lda, int64_t c0, int has_prev, int64_t rows_below, int64_t panel_end,
                             int64_t pivot_base, const double* diag0, double* S, bool split) {
    const int nfused = (int)(1 + rows_below / IB);
    const int rest_cols = has_prev ? (int)((panel_end - c0 - IB) / IB) : 0;
    int64_t nrest = 0;
    for (int tj = 0; tj < rest_cols; ++tj) nrest += nfused - 1 - tj;       // tiles on/below the diagonal, rows >= c0 + 64
    double* F = ctx->d_scal + 64;                                          // 64 x 64 + 64 doubles of the ctx's scalar scratch
    if (!split) {
        hipLaunchKernelGGL(panel_step_kernel, dim3((unsigned)(nfused + nrest)), dim3(256), 0, ctx->stream, A, lda, c0, has_prev, nfused,
                           rest_cols, pivot_base, diag0, ctx->d_info, S, ctx->pivot_tol, 0, F, (int64_t)0, (int64_t)0, (int64_t)0);
    } else {
        hipLaunchKernelGGL(panel_step_kernel, dim3(1), dim3(256), 0, ctx->stream, A, lda, c0, has_prev, nfused, rest_cols, pivot_base,
                           diag0, ctx->d_info, S, ctx->pivot_tol, 1, F, (int64_t)0, (int64_t)0, (int64_t)0);
        // [r3] the rows, then the update-role tiles, two units of work per workgroup (panel_units_kernel)
        const int nrow = nfused - 1;
        if (nrow > 0)
            hipLaunchKernelGGL(panel_units_kernel<2>, dim3((unsigned)((nrow + 1) / 2)), dim3(256), 0, ctx->stream, A, lda, c0, has_prev, nfused, 1, F, 0);
        if (nrest > 0)
            hipLaunchKernelGGL(panel_units_kernel<2>, dim3((unsigned)((nrest + 1) / 2)), dim3(256), 0, ctx->stream, A, lda, c0, has_prev, nfused, 0, F, (int)nrest);
    }
    MGPI_HIP_TRY(hipGetLastError());
    return 0;
}

// A[64 k + i][64 k + j] = S[64 k + i][j]: the factored diagonal blocks back into the matrix
__global__ __launch_bounds__(256) void diag_blocks_back_kernel(double* __restrict__ A, int64_t lda, const double* __restrict__ S,
                                                               int64_t first) {
    const int64_t k = first + blockIdx.x;
    for (int idx = threadIdx.x; idx < IB * IB; idx += 256) {
        const int i = idx >> 6, j = idx & 63;
        A[(k * IB + i) * lda + k * IB + j] = S[(k * IB + i) * IB + j];
    }
}

// multi-rhs 64-row block solve:  B <- L_jj^-1 B  (TRANS = false)  or  L_jj^-T B  (TRANS = true).
// B is (64 x n) row-major; one 256-thread workgroup = 64 columns of B, 4 lanes per column (part q owns rows
// [16q, 16q+16) of the block).
//   L x = b   : forward sweep,  C[k][m] = L[m][k], m > k
//   L^T x = b : backward sweep, C[k][m] = L[k][m], m < k
// blockIdx.y: one problem of a strided batch (sL / sB doubles apart; a 1-deep grid and zero strides otherwise)
template <bool TRANS>
__global__ __launch_bounds__(256) void trsm_cols_kernel(const double* __restrict__ Ljj, int64_t ldl,
                                                        double* __restrict__ B, int64_t ldb, int64_t sL, int64_t sB) {
    __shared__ __attribute__((aligned(16))) double C[IB][IB];
    MGPI_CHAIN_PRIO();
    const int tid = threadIdx.x, q = tid & 3;
    Ljj += (int64_t)blockIdx.y * sL;
    B += (int64_t)blockIdx.y * sB;
    if (TRANS) stage_coeff<false, false>(Ljj, ldl, C, tid);
    else stage_coeff<true, true>(Ljj, ldl, C, tid);
    double* col = B + (int64_t)blockIdx.x * 64 + (tid >> 2) + (int64_t)(16 * q) * ldb;   // n is a multiple of 64
    double x[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = col[j * ldb];
    __syncthreads();
    quad_sweep<IB - 1, !TRANS>::run(x, C, q);
#pragma unroll
    for (int j = 0; j < 16; ++j) col[j * ldb] = x[j];
}

int mgpi_launch_trsm_cols(mgpi_ctx* ctx, const double* Ljj, int64_t ldl, double* B, int64_t ldb, int64_t n, int trans) {
    if (n == 0) return 0;
    MGPI_REQUIRE(n % 64 == 0, "trsm_cols: n=%lld is not a multiple of 64", (long long)n);
    dim3 grid((unsigned)(n / 64)), block(256);
    if (trans) hipLaunchKernelGGL(trsm_cols_kernel<true>, grid, block, 0, ctx->stream, Ljj, ldl, B, ldb, (int64_t)0, (int64_t)0);
    else hipLaunchKernelGGL(trsm_cols_kernel<false>, grid, block, 0, ctx->stream, Ljj, ldl, B, ldb, (int64_t)0, (int64_t)0);
    MGPI_HIP_TRY(hipGetLastError());
    return 0;
}

// ---- inverse diagonal blocks of the factor (variance path only) ------------------------------------------------
// D_b = -(L_bb)^-1 for every nb x nb diagonal block of L, all blocks at once: start from -I and run the forward block
// solve of trsm_blocked on each block, the 8 inner steps batched over the blocks (2 launches per step instead of
// 2 per step per block).  With D the diagonal solve of the multi-rhs forward substitution, V_b <- L_bb^-1 V_b, is ONE
// gemm instead of a chain of 8 x (64-row gemm + 64-row substitution kernel) - the chain is what the variance of a
// medium-sized problem spends its time in (N = 8192: 9.6 ms, of which the gemm work is 5).
// Error budget: the product with an explicit inverse is accurate to ~cond(L_bb) eps relative to |V_b|, and
// cond(L_bb) <= sqrt(cond K); the variance is compared at 1e-8 of the prior variance, which
// leaves five orders of room at cond K = 1e7.  The factorisation and the solves that produce alpha do NOT use it.
__global__ void neg_identity_kernel(double* __restrict__ X, int64_t ldx, int64_t sX, int64_t w, int batch) {
    const int64_t id = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (id < (int64_t)batch * w) {
        const int64_t j = id / w, i = id % w;
        X[j * sX + i * ldx + i] = -1.0;
    }
}

// X_j (w x w, row stride ldx, sX doubles apart, ZEROED by the caller) := -(L_j)^-1 for `batch` lower-triangular blocks
// L_j (row stride lda, sL doubles apart); w a multiple of 64
static int invert_lower_blocks(mgpi_ctx* ctx, const double* L, int64_t lda, int64_t sL, double* X, int64_t ldx, int64_t sX,
                               int64_t w, int batch) {
    if (batch <= 0 || w <= 0) return 0;
    MGPI_REQUIRE(w % IB == 0, "block inversion: width %lld is not a multiple of %d", (long long)w, IB);
    hipLaunchKernelGGL(neg_identity_kernel, dim3((unsigned)((batch * w + 255) / 256)), dim3(256), 0, ctx->stream, X, ldx, sX, w, batch);
    for (int64_t c0 = 0; c0 < w; c0 += IB) {       // right-looking: the batched gemm has a contraction of 64
        hipLaunchKernelGGL(trsm_cols_kernel<false>, dim3((unsigned)(w / 64), (unsigned)batch), dim3(256), 0, ctx->stream,
                           L + c0 * lda + c0, lda, X + c0 * ldx, ldx, sL, sX);
        const int64_t c1 = c0 + IB;
        if (c1 < w)   // X_j[c1:w, :] -= L_j[c1:w, c0:c1] X_j[c0:c1, :]
            MGPI_TRY(mgpi_launch_gemm_nn_batched(ctx, w - c1, w, IB, L + c1 * lda + c0, lda, sL, X + c0 * ldx, ldx, sX,
                                                 X + c1 * ldx, ldx, sX, batch));
    }
    MGPI_HIP_TRY(hipGetLastError());
    return 0;
}

// [r3] The same inverses by recursive halving, for `count` blocks of width nb = 64 * 2^p stored nb (nb + 1) doubles apart
// (mgpi_dinv_block): with L = [A 0; C B] the inverse is [A^-1 0; -B^-1 C A^-1  B^-1], so after the 64 x 64 diagonal tiles are
// inverted (ONE batched substitution launch over all tiles of all blocks) every level h = 64, 128, .. is two batched gemms
// over all (block, pair) sub-problems - the skewed block stride makes their offsets one arithmetic progression, 2h (ld + 1)
// in L and in X alike.  In the stored sign (X = -L^-1, "C -= A B" gemms on zeroed targets):  T = -C X11,  X21 = -X22 T.
// 1 + 2 log2(nb / 64) kernel launches (7 at nb = 512) instead of 2 nb / 64 - 1 (15), none of them a 64-row substitution over
// the whole block width: 0.3 -> 0.1 ms, which matters since the single-right-hand-side backward sweep uses the inverses too.
static int invert_lower_blocks_halving(mgpi_ctx* ctx, const double* L, int64_t lda, double* X, int64_t nb, int64_t count,
                                       double* T) {
    if (count <= 0) return 0;
    const int64_t sX = nb * (nb + 1);
    hipLaunchKernelGGL(neg_identity_kernel, dim3((unsigned)((count * nb + 255) / 256)), dim3(256), 0, ctx->stream, X, nb, sX, nb, (int)count);
    const int64_t tiles = count * (nb / IB);
    MGPI_REQUIRE(tiles < 65536, "block inversion: %lld diagonal tiles", (long long)tiles);
    hipLaunchKernelGGL(trsm_cols_kernel<false>, dim3(1, (unsigned)tiles), dim3(256), 0, ctx->stream, L, lda, X, nb, IB * (lda + 1),
                       IB * (nb + 1));
    for (int64_t h = IB; 2 * h <= nb; h *= 2) {
        const int subs = (int)(count * (nb / (2 * h)));
        MGPI_HIP_TRY(hipMemsetAsync(T, 0, (size_t)subs * h * h * sizeof(double), ctx->stream));
        MGPI_TRY(mgpi_launch_gemm_nn_batched(ctx, h, h, h, L + h * lda, lda, 2 * h * (lda + 1), X, nb, 2 * h * (nb + 1), T, h, h * h, subs));
        MGPI_TRY(mgpi_launch_gemm_nn_batched(ctx, h, h, h, X + h * (nb + 1), nb, 2 * h * (nb + 1), T, h, h * h, X + h * nb, nb,
                                             2 * h * (nb + 1), subs));
    }
    MGPI_HIP_TRY(hipGetLastError());
    return 0;
}

// block b of the cache: nb x nb, row stride nb, nb (nb + 1) doubles apart (see invert_lower_blocks_halving)
static inline double* mgpi_dinv_block(const mgpi_mat* Lm, int64_t b) { return Lm->dinv + b * Lm->dinv_nb * (Lm->dinv_nb + 1); }

static int ensure_dinv(mgpi_ctx* ctx, const mgpi_mat* Lc) {
    mgpi_mat* Lm = const_cast<mgpi_mat*>(Lc);          // a cache on the factor, invisible to the caller
    const int64_t nb = ctx->nb;
    if (Lm->dinv != nullptr && Lm->dinv_nb == nb) return 0;
    const int64_t pn = Lm->prows <= Lm->pcols ? Lm->prows : mgpi_roundup(Lm->rows, IB);
    if (Lm->dinv) {
        mgpi_pool_free(ctx, Lm->dinv, Lm->dinv_bytes);
        Lm->dinv = nullptr;
    }
    const int64_t nfull = pn / nb, wlast = pn - nfull * nb;          // full blocks, width of a narrower last one
    const int64_t sX = nb * (nb + 1);
    const size_t bytes = (size_t)(nfull + 1) * (size_t)sX * sizeof(double);
    MGPI_TRY(mgpi_pool_alloc(ctx, bytes, &Lm->dinv, &Lm->dinv_bytes));
    double* X = Lm->dinv;
    const int64_t lda = Lm->ld;
    MGPI_HIP_TRY(hipMemsetAsync(X, 0, bytes, ctx->stream));
    const bool pow2 = nb >= IB && ((nb / IB) & (nb / IB - 1)) == 0;
    if (pow2 && nfull * (nb / IB) < 65536) {
        double* T = nullptr;
        size_t tbytes = 0;
        MGPI_TRY(mgpi_pool_alloc(ctx, (size_t)nfull * (size_t)(nb / 2) * (size_t)(nb / 2) * sizeof(double) + 64, &T, &tbytes));
        const int rc = invert_lower_blocks_halving(ctx, Lm->d, lda, X, nb, nfull, T);
        mgpi_pool_free(ctx, T, tbytes);                              // reuse is ordered on this stream
        if (rc) return rc;
    } else {
        MGPI_TRY(invert_lower_blocks(ctx, Lm->d, lda, nb * (lda + 1), X, nb, sX, nb, (int)nfull));
    }
    MGPI_TRY(invert_lower_blocks(ctx, Lm->d + nfull * nb * (lda + 1), lda, 0, X + nfull * sX, nb, 0, wlast, wlast > 0 ? 1 : 0));
    Lm->dinv_nb = nb;
    return 0;
}

// Strided batch for the multi-GPU driver: block j is A[r0 + j row_step : + w, c0 + j col_step : + w] (the diagonal
// blocks a rank owns sit at a constant stride in its local block-cyclic matrix); out rows [j w, (j+1) w) := -(block j)^-1.
extern "C" int mgpi_trtri_neg_dev(mgpi_ctx* ctx, const mgpi_mat* A, int64_t r0, int64_t c0, int64_t row_step, int64_t col_step,
                                  int64_t w, int batch, mgpi_mat* out) {
    MGPI_REQUIRE(ctx && A && out, "mgpi_trtri_neg_dev: null pointer");
    MGPI_REQUIRE(batch >= 1 && w >= IB && w % IB == 0, "mgpi_trtri_neg_dev: batch=%d, w=%lld (a positive multiple of %d)", batch,
                 (long long)w, IB);
    MGPI_REQUIRE(r0 >= 0 && c0 >= 0 && row_step >= 0 && col_step >= 0 && r0 + (batch - 1) * row_step + w <= A->prows &&
                     c0 + (batch - 1) * col_step + w <= A->pcols,
                 "mgpi_trtri_neg_dev: blocks reach outside the %lld x %lld matrix", (long long)A->prows, (long long)A->pcols);
    MGPI_REQUIRE(out->prows >= (int64_t)batch * w && out->pcols >= w, "mgpi_trtri_neg_dev: out is %lld x %lld, needs %lld x %lld",
                 (long long)out->prows, (long long)out->pcols, (long long)batch * w, (long long)w);
    MGPI_HIP_TRY(hipSetDevice(ctx->device));
    MGPI_HIP_TRY(hipMemset2DAsync(out->d, (size_t)out->ld * sizeof(double), 0, (size_t)w * sizeof(double), (size_t)batch * w,
                                  ctx->stream));
    mgpi_mat_invalidate(out);
    return invert_lower_blocks(ctx, A->d + r0 * A->ld + c0, A->ld, row_step * A->ld + col_step, out->d, out->ld, w * out->ld, w, batch);
}

// rows [0, w) of T (zeroed, ldt) := L_bb^-1 * B[r0:r0+w, :]  then copied back over those rows of B
__global__ void copy_rows_kernel(const double* __restrict__ src, int64_t lds_, double* __restrict__ dst, int64_t ldd, int64_t rows,
                                 int64_t cols) {
    const int64_t c = ((int64_t)blockIdx.x * blockDim.x + threadIdx.x) * 2, r = blockIdx.y;
    if (c < cols && r < rows) *reinterpret_cast<double2*>(dst + r * ldd + c) = *reinterpret_cast<const double2*>(src + r * lds_ + c);
}

// ---- small helpers ------------------------------------------------------------------------------------
// padded diagonal := 1 so the padded matrix stays positive definite (rows n..pn)
__global__ void pad_identity_kernel(double* A, int64_t lda, int64_t n, int64_t pn) {
    const int64_t i = n + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < pn) A[i * lda + i] = 1.0;
}

__global__ void save_diag_kernel(const double* __restrict__ A, int64_t lda, int64_t pn, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < pn) out[i] = A[i * lda + i];
}

// out[0] = sum_i log(A[i][i]) over i < n (single workgroup, fixed summation order => deterministic)
// accumulate != 0: out[0] += sum (the block-by-block factorisation of the multi-GPU driver)
// range != nullptr: range[0] = min_i A[i][i], range[1] = max_i A[i][i] (what trsv_sweeps judges the factor's conditioning by)
__global__ __launch_bounds__(256) void logdiag_kernel(const double* __restrict__ A, int64_t lda, int64_t n,
                                                      double* __restrict__ out, int accumulate, double* __restrict__ range) {
    __shared__ double part[256], lo[256], hi[256];
    double s = 0.0, mn = INFINITY, mx = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        const double d = A[i * lda + i];
        s += log(d);
        mn = fmin(mn, d);                                     // a NaN diagonal (failed factorisation) is ignored here: the status says so
        mx = fmax(mx, d);
    }
    part[threadIdx.x] = s;
    lo[threadIdx.x] = mn;
    hi[threadIdx.x] = mx;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            part[threadIdx.x] += part[threadIdx.x + w];
            lo[threadIdx.x] = fmin(lo[threadIdx.x], lo[threadIdx.x + w]);
            hi[threadIdx.x] = fmax(hi[threadIdx.x], hi[threadIdx.x + w]);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        out[0] = accumulate ? out[0] + part[0] : part[0];
        if (range != nullptr) {
            range[0] = lo[0];
            range[1] = hi[0];
        }
    }
}

// One wave that returns once *counter has reached `want` (wrap-safe): how the look-ahead stream waits for PART of a kernel
// that runs on the main stream.  hipStreamWaitValue32 does the same without a wave (measured equal: ~1 us from the last
// increment to the next kernel's start, tools/waitvalue_probe.hip) but a factorisation that uses it never finishes under
// rocprofv3 --pmc (tools/pmc_hang_probe.sh), so it is not used.  The producer is always submitted before this kernel, and the
// poll is bounded: after 2^23 sleeps (~23 s; the largest trailing update that fits the HBM takes ~1.5 s) it gives up and flags the factorisation as failed (status -1) instead of hanging.
__global__ void wait_count_kernel(const unsigned* __restrict__ counter, unsigned want, int* __restrict__ info) {
    if (threadIdx.x != 0) return;
    if (__hip_atomic_load(info, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 0) return;      // an earlier wait of this factorisation gave up
    for (int guard = 0; guard < (1 << 23); ++guard) {
        if ((int)(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - want) >= 0) return;
        __builtin_amdgcn_s_sleep(32);
    }
    atomicCAS(info, 0, -1);
}

// Can a kernel of the look-ahead stream run WHILE a kernel of the main stream is still running?  The merged trailing update
// depends on it (its consumer starts on a count the producer publishes half-way), and there are environments where the answer
// is no: rocprofv3 --pmc, AMD_SERIALIZE_KERNEL, HIP_LAUNCH_BLOCKING, a single hardware queue, a debug layer - there a merged
// factorisation would stall in wait_count_kernel until its bound and then fail.  The known switches are read from the
// environment; everything else is caught by asking the device once per context: a two-way handshake between one wave on
// each stream, both bounded (~16 ms), which only succeeds when the two kernels are resident at the same time, whatever the
// order a serialising tool would run them in.
__global__ void overlap_probe_kernel(unsigned* __restrict__ mine, const unsigned* __restrict__ theirs, int* __restrict__ ok) {
    if (threadIdx.x != 0) return;
    __hip_atomic_store(mine, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int guard = 0; guard < (1 << 14); ++guard) {
        if (__hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0u) {
            atomicAdd(ok, 1);
            return;
        }
        __builtin_amdgcn_s_sleep(32);
    }
}

static bool serialising_environment() {
    for (const char* name : {"ROCPROF_COUNTER_COLLECTION", "AMD_SERIALIZE_KERNEL", "HIP_LAUNCH_BLOCKING", "CUDA_LAUNCH_BLOCKING"})
        if (const char* v = getenv(name)) if (atoi(v) != 0) return true;
    if (const char* v = getenv("GPU_MAX_HW_QUEUES")) if (atoi(v) == 1) return true;
    return false;
}

// the handshake between stream a (which owns the scratch and the event) and stream b: 1 = the two kernels were resident together
static int overlap_handshake(mgpi_ctx* ctx, hipStream_t a, hipStream_t b) {
    if (serialising_environment()) return 0;
    unsigned* flags = reinterpret_cast<unsigned*>(ctx->d_scal + 8072);     // two handshake words and the verdict
    int* ok = reinterpret_cast<int*>(flags + 2);
    if (hipMemsetAsync(flags, 0, 4 * sizeof(unsigned), a) != hipSuccess) return 0;
    if (hipEventRecord(ctx->ev_join, a) != hipSuccess || hipStreamWaitEvent(b, ctx->ev_join, 0) != hipSuccess) return 0;
    hipLaunchKernelGGL(overlap_probe_kernel, dim3(1), dim3(64), 0, a, flags, flags + 1, ok);
    hipLaunchKernelGGL(overlap_probe_kernel, dim3(1), dim3(64), 0, b, flags + 1, flags, ok);
    int both = 0;
    if (hipEventRecord(ctx->ev_join, b) != hipSuccess || hipStreamWaitEvent(a, ctx->ev_join, 0) != hipSuccess ||
        hipMemcpyAsync(&both, ok, sizeof(int), hipMemcpyDeviceToHost, a) != hipSuccess || hipStreamSynchronize(a) != hipSuccess) {
        (void)hipGetLastError();
        return 0;
    }
    return both == 2 ? 1 : 0;
}

// 1: the two streams of ctx overlap; 0: they do not (or the question could not be asked).  Asked once per context.
static int streams_overlap(mgpi_ctx* ctx) {
    if (ctx->streams_overlap >= 0) return ctx->streams_overlap;
    ctx->streams_overlap = ctx->stream2 ? overlap_handshake(ctx, ctx->stream, ctx->stream2) : 0;
    return ctx->streams_overlap;
}

// the same question for the streams of TWO contexts of one device (the multi-GPU driver's main and look-ahead contexts)
extern "C" int mgpi_ctx_streams_overlap(mgpi_ctx* a, mgpi_ctx* b, int* yes) {
    MGPI_REQUIRE(a && b && yes, "mgpi_ctx_streams_overlap: null pointer");
    MGPI_REQUIRE(a->device == b->device, "mgpi_ctx_streams_overlap: contexts on different devices");
    MGPI_HIP_TRY(hipSetDevice(a->device));
    *yes = (a == b) ? 0 : overlap_handshake(a, a->stream, b->stream);
    return 0;
}

// progress counter of ctx's merged trailing updates (mgpi_gemm_nt_stair_merged_dev): back to zero, on ctx's stream
extern "C" int mgpi_ctx_count_reset(mgpi_ctx* ctx) {
    MGPI_REQUIRE(ctx != nullptr, "mgpi_ctx_count_reset: null ctx");
    MGPI_HIP_TRY(hipSetDevice(ctx->device));
    MGPI_HIP_TRY(hipMemsetAsync(ctx->d_scal + 8064, 0, sizeof(unsigned), ctx->stream));
    return 0;
}

// what `waiter` enqueues from now on starts once `owner`'s progress counter has reached `target` (one sleeping wave on waiter's
// stream; bounded: if the count never comes the waiter's status word turns negative and its next status read fails)
extern "C" int mgpi_ctx_wait_count(mgpi_ctx* waiter, mgpi_ctx* owner, int64_t target) {
    MGPI_REQUIRE(waiter && owner, "mgpi_ctx_wait_count: null ctx");
    MGPI_REQUIRE(waiter->device == owner->device && target >= 0 && target < ((int64_t)1 << 31), "mgpi_ctx_wait_count: bad argument");
    MGPI_HIP_TRY(hipSetDevice(waiter->device));
    hipLaunchKernelGGL(wait_count_kernel, dim3(1), dim3(64), 0, waiter->stream, reinterpret_cast<const unsigned*>(owner->d_scal + 8064),
                       (unsigned)target, waiter->d_info);
    MGPI_HIP_TRY(hipGetLastError());
    return 0;
}

// ---- the factorisation ----------------------------------------------------------------------------------
// Enqueues the whole factorisation of K on ctx's streams and returns without a host synchronisation.
//   pivot_base  global index of K's first row: a failing pivot k is reported as pivot_base + k (1-based)
//   keep_status 0: the failing-pivot word is cleared first and sum log L_ii goes to d_scal[0] (mgpi_potrf_dev reads both)
//               1: the failing pivot is min-accumulated into the ctx status word and sum log L_ii is ADDED to
//                  d_scal[1] - the block-by-block factorisation of the multi-GPU driver reads them once at the end
//   fwd_rhs     nullptr, or a vector of pn doubles (zero beyond n): the FORWARD sweep L y = f of a solve runs beside the
//               factorisation - as soon as a super-panel's columns are final, a third stream solves its diagonal blocks (64 x 64
//               tiles from the scratch strip) and applies its columns to the rows below, in the shadow of the trailing update
//               that runs at the same time.  On return (stream order) fwd_rhs holds y; the solve then only owes the backward
//               sweep.  Same kernels, same arithmetic and order as trsv_pair's forward half.
//   tall        0, or 1: K is a TALL block column (m x w, m >= w, both multiples of 64; a window of a rank's local block-cyclic
//               matrix): its top w x w block is factored and the rows below are solved against it - one super-panel of the
//               factorisation and nothing else (no trailing matrix to update).  The block column of the multi-GPU driver on grids
//               with one process row, where the whole panel is local: the same fused step kernels as the single-GPU path.
//   diag_ref    nullptr, or the ORIGINAL diagonal entries of the rows being factored (device, one per column factored): what a
//               pivot is measured against.  A block that earlier block steps have already updated no longer shows them - the
//               diagonal entry of a duplicated source row is rounding noise by then, and noise above 100 eps times noise would
//               pass - so the multi-GPU driver hands them in (it keeps the diagonal of K from the build).
//   rider_rows  0, or pm (a multiple of 128): the matrix has pm more rows below its pn x pn square, R (pm x pn) - "riders".  They
//               take part in every panel solve and every column update like the rows of K below the panel (the rows simply run
//               on), and after every trailing update of the square ONE more NT gemm on the main stream applies the super-panel to
//               the rider columns behind the next one - so that they end as R L^-T: the forward solve of a multi-right-hand-side
//               system (V^T = K* L^-T of the posterior variance) INSIDE the factorisation instead of a blocked triangular solve of
//               its own afterwards.  The same N^2 M flops, solved by the panels' own substitutions, no second chain of
//               diagonal-block solves.  bench.py steps, same box (two-pass form / this): C2 13.94 / 12.99 ms, C3 260.0 / 258.6.
//               Measured and dropped (profiles/r03_riders_ab.md): the riders' update on the third stream beside the square's
//               (C2 14.35, C3 262.6); inside the square's launch as a trapezoid tile list (14.03 / 263.8: the next panel's
//               block column then waits for the rider tiles too); the riders as a sweep of their own one super-panel behind
//               (solve on the third stream from the strip's diagonal tiles: 8 substitution launches per panel, a second latency
//               chain the main stream ends up waiting for: 13.37 / 261.0).
static int potrf_enqueue(mgpi_ctx* ctx, mgpi_mat* K, int64_t pivot_base, int keep_status, double* fwd_rhs = nullptr, int tall = 0,
                         const double* diag_ref = nullptr, int64_t rider_rows = 0) {
    MGPI_REQUIRE(ctx && K, "mgpi_potrf_dev: null pointer");
    if (!tall && rider_rows > 0) {
        MGPI_REQUIRE(rider_rows % MGPI_TILE == 0 && K->pcols % MGPI_TILE == 0 && K->prows >= K->pcols + rider_rows && K->cols <= K->pcols,
                     "mgpi_posv_riders_dev: matrix is %lld x %lld (padded %lld x %lld), riders %lld", (long long)K->rows,
                     (long long)K->cols, (long long)K->prows, (long long)K->pcols, (long long)rider_rows);
    } else if (!tall) {
        MGPI_REQUIRE(K->rows == K->cols, "mgpi_potrf_dev: matrix is %lld x %lld, not square", (long long)K->rows, (long long)K->cols);
        MGPI_REQUIRE(K->pcols >= K->prows || K->pcols >= mgpi_roundup(K->rows, IB),
                     "mgpi_potrf_dev: matrix must be tile padded (allocate with cols > %d)", MGPI_MAX_D);
    } else {
        MGPI_REQUIRE(K->prows >= K->pcols && K->prows % IB == 0 && K->pcols % IB == 0 && K->rows == K->prows && K->cols == K->pcols && fwd_rhs == nullptr,
                     "mgpi_potrf_panel_enqueue_dev: block column is %lld x %lld (rows >= columns, both multiples of %d)",
                     (long long)K->prows, (long long)K->pcols, IB);
    }
    MGPI_HIP_TRY(hipSetDevice(ctx->device));
    K->diag_min = K->diag_max = 0.0;                 // unknown until this factorisation's status is read
    if (K->dinv) {                                   // inverse diagonal blocks of a previous factor
        mgpi_pool_free(ctx, K->dinv, K->dinv_bytes);
        K->dinv = nullptr;
    }
    const int64_t n = rider_rows > 0 ? K->cols : K->rows;
    const int64_t pn = tall ? K->prows : (rider_rows > 0 ? K->pcols : (K->prows <= K->pcols ? K->prows : mgpi_roundup(n, IB)));   // rows of the factored extent (multiple of 64)
    const int64_t pm = rider_rows, prow = pn + pm;                                // riders: rows [pn, prow)
    const int64_t ncols = tall ? K->pcols : pn;                                   // columns that get factored
    const int64_t ndiag = tall ? K->pcols : n;                                    // diagonal entries that enter sum log L_ii
    const int64_t lda = K->ld;
    double* A = K->d;
    const int64_t nb = ctx->nb;

    if (!keep_status) MGPI_HIP_TRY(hipMemsetAsync(ctx->d_info, 0, sizeof(int), ctx->stream));
    if (pn > n) {
        hipLaunchKernelGGL(pad_identity_kernel, dim3((unsigned)((pn - n + 63) / 64)), dim3(64), 0, ctx->stream, A, lda, n, pn);
        MGPI_HIP_TRY(hipGetLastError());
    }
    if (ctx->diag_len < pn) {
        if (ctx->d_diag) {
            MGPI_HIP_TRY(hipStreamSynchronize(ctx->stream));
            MGPI_HIP_TRY(hipFree(ctx->d_diag));
            ctx->d_diag = nullptr;
            ctx->diag_len = 0;
        }
        // diag(K) before factoring, then a 64-wide strip for the factored diagonal blocks (panel_step_kernel)
        MGPI_HIP_TRY(hipMalloc((void**)&ctx->d_diag, (size_t)pn * (1 + IB) * sizeof(double)));
        ctx->diag_len = pn;
    }
    if (diag_ref != nullptr) {
        MGPI_HIP_TRY(hipMemcpyAsync(ctx->d_diag, diag_ref, (size_t)ncols * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    } else {
        hipLaunchKernelGGL(save_diag_kernel, dim3((unsigned)((ncols + 255) / 256)), dim3(256), 0, ctx->stream, A, lda, ncols,
                           ctx->d_diag);
        MGPI_HIP_TRY(hipGetLastError());
    }
    mgpi_prof_begin(ctx, 2);

    // panel factorisation of columns [r0, r0+w), rows [r0, pn), enqueued on whatever ctx->stream currently is: one launch per
    // 64-column step (panel_step_kernel), right-looking inside the panel - after 64 columns are solved ALL remaining columns
    // of the panel are updated with them (contraction 64, deferred by one step).  Panels whose first diagonal block has more
    // than MGPI_PANEL_STEP_ROWS rows below it run every step as two launches (diagonal block, then the rows).
    // [r3] ... of matrices above 16384 rows, where a trailing update runs beside the panel; smaller ones never split (C2 -1.7 %).
    static const int64_t step_rows_env = getenv("MGPI_PANEL_STEP_ROWS") ? atoll(getenv("MGPI_PANEL_STEP_ROWS")) : -1;
    const bool small_matrix = !tall && pn <= 16384;     // a block column of the multi-GPU driver belongs to a large matrix
    const int64_t step_rows = step_rows_env >= 0 ? step_rows_env : (small_matrix ? INT64_MAX / 2 : 8192);
    ctx->skinny_tiles = small_matrix ? 1300 : 448;
    int64_t strip_first = pn / IB;                      // first 64-block whose factored diagonal block waits in the scratch strip
    // the steps of the columns [c_begin, c_end) of the panel [r0, r0 + w)
    auto factor_panel_range = [&](int64_t r0, int64_t w, int64_t c_begin, int64_t c_end) -> int {
        const bool split = prow - r0 - IB > step_rows;
        for (int64_t c0 = c_begin; c0 < c_end; c0 += IB) {
            MGPI_TRY(launch_panel_step(ctx, A, lda, c0, c0 > r0 ? 1 : 0, prow - c0 - IB, r0 + w, pivot_base + c0, ctx->d_diag + c0,
                                       ctx->d_diag + ctx->diag_len + c0 * IB, split));
            if (c0 / IB < strip_first) strip_first = c0 / IB;
        }
        return 0;
    };
    auto factor_panel = [&](int64_t r0, int64_t w) -> int { return factor_panel_range(r0, w, r0, r0 + w); };

    // Two-level blocking.  The panels are nb wide (the blocked algorithm of the reference configuration), but while
    // the trailing matrix is large the trailing update is applied once per SUPER-PANEL of MGPI_SUPER_PANELS (default
    // 2) panels, i.e. with a contraction length of 1024 instead of 512: the C tiles then cross HBM half as often and
    // the tile prologue/epilogue is amortised over twice the MFMA work.  Measured on the lower-syrk launch alone
    // (tools/syrk_only.py n reps K): 66.9 / 69.5 / 70.6 TFLOP/s at 32256 rows for K = 512 / 1024 / 2048, 57.7 / 63.2 /
    // 68.9 at 16384.  Inside a super-panel the factorisation is the same right-looking nb-panel algorithm restricted
    // to the super-panel's columns.  In the whole factorisation the gain is small, because the work moved out of the
    // big update (wider column updates, the skinny updates inside the super-panel) runs on the look-ahead stream and
    // lengthens the chain the next big update waits for: potrf at N = 32768 is 195.2 ms with 1 panel per update,
    // 193.1 with 2, 195.5 with 3, 195.4-199 with 4 (bench.py, 3 steps each).  Below MGPI_SUPER_MIN trailing rows the
    // width drops back to one panel.
    //
    // Look-ahead (depth 1, in units of super-panels).  The trailing update of super-panel k is split by columns:
    // the block columns of super-panel k+1, and the rest.  The column update and then the factorisation of
    // super-panel k+1 run on the second (high-priority) stream, the rest on the main stream AT THE SAME TIME - the two
    // write disjoint columns and both only read super-panel k.  So the main stream sees one big lower-triangular
    // update after another.  Events: ev_upd = "the rest of step k-1 is done" (its columns include the ones the second
    // stream updates next), ev_panel = "super-panel k+1 is factored".
    // Large panels are factored as sub-panels of MGPI_PANEL_INNER columns with a gemm between them: inside a (sub-)panel the
    // update is right-looking in 64-column steps of contraction 64 (short chain, but every step rewrites the rest of the
    // panel through 64 x 64 x 64 tiles straight from L2), between sub-panels it is one LDS-staged gemm of contraction
    // MGPI_PANEL_INNER.  Used while more than MGPI_PANEL_INNER_ROWS rows lie below the panel, where the chain is not critical.
    // N = 32768: potrf 188.1 ms with 512 (no sub-panels), 186.3 with 256, 187.2 with 128.
    static const int64_t inner_w = getenv("MGPI_PANEL_INNER") ? atoll(getenv("MGPI_PANEL_INNER")) : 256;
    static const int64_t inner_rows = getenv("MGPI_PANEL_INNER_ROWS") ? atoll(getenv("MGPI_PANEL_INNER_ROWS")) : 8192;
    auto factor_super = [&](int64_t c0, int64_t wc, bool on_side) -> int {
        const int64_t nb = (inner_w >= IB && inner_w % IB == 0 && inner_w < ctx->nb && prow - c0 - wc > inner_rows) ? inner_w : ctx->nb;
        for (int64_t p0 = c0; p0 < c0 + wc; p0 += nb) {
            const int64_t w = (c0 + wc - p0 < nb) ? (c0 + wc - p0) : nb;
            ctx->lean = on_side ? 1 : 0;
            int rc = factor_panel(p0, w);
            ctx->lean = 0;
            if (rc) return rc;
            const int64_t p1 = p0 + w, rem = c0 + wc - p1, t = prow - p1;
            if (rem > 0 && t > 0)    // the rest of the super-panel's columns: A[p1:pn, p1:c0+wc] -= P P[0:rem]^T
                MGPI_TRY(mgpi_launch_gemm(ctx, 0, 0, 0, t, rem, w, A + p1 * lda + p0, lda, A + p1 * lda + p0, lda,
                                          A + p1 * lda + p1, lda));
        }
        return 0;
    };
    static const int64_t super_min = getenv("MGPI_SUPER_MIN") ? atoll(getenv("MGPI_SUPER_MIN")) : 14336;
    static const int64_t super_panels = getenv("MGPI_SUPER_PANELS") ? atoll(getenv("MGPI_SUPER_PANELS")) : 2;
    auto super_width = [&](int64_t c0) -> int64_t {      // width of the super-panel that starts at column c0
        const int64_t left = pn - c0;
        // the very first one has nothing to hide behind: keep it one panel wide
        int64_t wc = (c0 > 0 && left - super_panels * nb >= super_min) ? super_panels * nb : nb;
        return wc < left ? wc : left;
    };

    hipStream_t main_stream = ctx->stream;
    mgpi_stream_join_guard join_on_error{ctx, main_stream, true};
    const bool lookahead = ctx->stream2 != nullptr && !ctx->no_lookahead;
    // forward sweep of the columns [c_begin, c_end), just factored on ctx->stream (whichever that is right now)
    bool fwd_used = false;
    auto forward_columns = [&](int64_t c_begin, int64_t c_end) -> int {
        if (fwd_rhs == nullptr) return 0;
        hipStream_t fs = (ctx->stream3 != nullptr && lookahead) ? ctx->stream3 : ctx->stream;
        if (fs != ctx->stream) {
            MGPI_HIP_TRY(hipEventRecord(ctx->ev_fwd, ctx->stream));
            MGPI_HIP_TRY(hipStreamWaitEvent(fs, ctx->ev_fwd, 0));
            fwd_used = true;
        }
        const double* S = ctx->d_diag + ctx->diag_len;             // factored 64 x 64 diagonal tiles, row r of the matrix at S + 64 r
        for (int64_t b0 = c_begin; b0 < c_end; b0 += nb) {
            const int w = (int)((c_end - b0 < nb) ? (c_end - b0) : nb);
            const trsv_diag_src src{S + b0 * IB, IB, 0};
            launch_trsv_diag(fs, A + b0 * lda + b0, lda, fwd_rhs + b0, w, 0, &src);
            launch_trsv_update(fs, A, lda, fwd_rhs, b0, w, b0 + w, pn - (b0 + w), 0);
        }
        MGPI_HIP_TRY(hipGetLastError());
        return 0;
    };
    // riders (rows [pn, prow)), super-panel k in columns [r0, r0 + w): R[:, c_lo:c_hi] -= R[:, r0:r0+w] A[c_lo:c_hi, r0:r0+w]^T, a
    // plain NT gemm on the shipped kernel, on whatever ctx->stream is
    auto rider_update = [&](int64_t r0, int64_t w, int64_t c_lo, int64_t c_hi) -> int {
        if (pm <= 0 || c_hi <= c_lo) return 0;
        mgpi_prof_begin(ctx, 4);
        const int rc = mgpi_launch_gemm(ctx, 0, 0, 0, pm, c_hi - c_lo, w, A + pn * lda + r0, lda, A + c_lo * lda + r0, lda, A + pn * lda + c_lo, lda);
        mgpi_prof_end(ctx, 4);
        mgpi_prof_flops(ctx, 4, 2.0 * (double)pm * (double)(c_hi - c_lo) * (double)w);
        return rc;
    };
    // Merged trailing update (round 2, second half).  The block column of super-panel k+1 used to be updated by a gemm of its
    // own on the second stream, beside the main stream's update of the rest: a skinny launch (24.7 ms of stream time per
    // factorisation at N = 32768 for 1.1e12 flops) that took CUs from the big one.  Now ONE launch per step updates the whole
    // trailing matrix, the tiles of that block column first (merged_decode), and counts them on a device word as they are
    // stored; the second stream waits for the count (wait_count_kernel: ~1 us after the last tile, tools/waitvalue_probe.hip)
    // and factors super-panel k+1 while the rest of the same launch runs.  Used while at least MGPI_MERGED_MIN rows trail
    // (below that the chain is critical and the narrow-tile column update reaches the panel sooner); MGPI_POTRF_MERGED=0
    // keeps the separate column update everywhere (developer switch, tested).
    // Under rocprofv3 --pmc (ROCPROF_COUNTER_COLLECTION=1) a profiled kernel runs ALONE on the GPU, so one that feeds a kernel
    // running beside it can never do so: the separate column update is used there (same arithmetic) - and in every other
    // environment in which the two streams do not overlap (streams_overlap above: known switches + a device handshake).
    // An explicit MGPI_POTRF_MERGED=1 still forces the merged form (how the bounded wait's failure path is tested).
    static const int merged_env = getenv("MGPI_POTRF_MERGED") ? atoi(getenv("MGPI_POTRF_MERGED")) : -1;
    // [r3] 12288 -> 8192 after the round's other changes (tools/sweep_r03_merged.sh, MGPI_MERGED_MIN = 4096 / 6144 / 8192 / 12288):
    // C3 257.7 / 257.7 / 257.7 / 258.6 ms per step, potrf N = 20480 51.9 / 52.2 / 52.3 / 52.6 ms, N <= 10240 within noise
    static const int64_t merged_min = getenv("MGPI_MERGED_MIN") ? atoll(getenv("MGPI_MERGED_MIN")) : 8192;
    // (the device is only asked when this matrix is large enough for a merged step to occur at all)
    const bool merged_on = merged_env >= 0 ? merged_env != 0 : (lookahead && pn > merged_min && streams_overlap(ctx) == 1);
    unsigned* d_sync = reinterpret_cast<unsigned*>(ctx->d_scal + 8064);
    unsigned sync_target = 0;          // tiles counted so far in this factorisation: the word only grows until the next reset
    bool merged_prev = false;          // did the previous step take the merged form (the second stream then owes no event wait)
    if (lookahead && merged_on) MGPI_HIP_TRY(hipMemsetAsync(d_sync, 0, sizeof(unsigned), main_stream));
    // The TAIL.  Once few rows trail, a panel step (~31 us) is shorter than what surrounds it: the next panel's column update
    // alone takes 70 - 125 us however small it is (one tile is 32 dependent stages of contraction 16), so the last eight
    // panels of any factorisation cost ~3.9 ms where their 64 steps need 2.5 (rocprofv3 timeline at N = 8192).  Below
    // MGPI_TAIL_ROWS trailing rows the rest of the matrix is therefore factored as ONE panel, the way the batched
    // log-likelihood does it: no column updates, no trailing gemms - the update-role workgroups of every step apply its 64
    // columns to everything right of them (64 x 64 x 64 tiles from L2, hidden beside the step's own chain).
    // Measured (potrf, ms; MGPI_TAIL_ROWS = 0 / 2048 / 4096): N=2048 1.28 / 1.13 / 1.12, N=4096 2.74 / 2.55 / 2.56, N=8192 7.53 / 7.36 / 7.41,
    // N=13824 21.0 / 20.8 / 20.7 - a step costs 35 - 40 us in that form, so the gain is what the column updates cost.
    static const int64_t tail_rows = getenv("MGPI_TAIL_ROWS") ? atoll(getenv("MGPI_TAIL_ROWS")) : 2048;
    int64_t wc = super_width(0);
    if (pn <= tail_rows) wc = pn;                                // small matrices: one panel from the start
    if (tall) wc = ncols;                                        // a block column: its columns are one super-panel, and that is all
    mgpi_prof_begin(ctx, 1);
    // [r3] a panel wider than nb (the whole of a small matrix, the tail of a large one) with a forward sweep riding along:
    // nb columns at a time, so that the sweep of each block starts when ITS steps are done, not after the panel's last one
    auto factor_wide_panel = [&](int64_t r0, int64_t w) -> int {
        if (fwd_rhs == nullptr) return factor_panel(r0, w);
        for (int64_t c = r0; c < r0 + w; c += nb) {
            const int64_t ce = (c + nb < r0 + w) ? c + nb : r0 + w;
            MGPI_TRY(factor_panel_range(r0, w, c, ce));
            MGPI_TRY(forward_columns(c, ce));
        }
        return 0;
    };
    if (!tall && wc == pn && pn <= tail_rows) {
        MGPI_TRY(factor_wide_panel(0, pn));
        mgpi_prof_end(ctx, 1);
    } else {
        MGPI_TRY(factor_super(0, wc, false));
        mgpi_prof_end(ctx, 1);
        MGPI_TRY(forward_columns(0, wc));
    }
    if (lookahead) MGPI_HIP_TRY(hipEventRecord(ctx->ev_upd, main_stream));
    bool panel_on_side = false;                                  // is the current super-panel the product of stream2?
    // [r3] In the un-merged look-ahead steps the main stream's update of the rest waits until the next panel's block column
    // has been updated on the second stream: started together, the big update took the CUs the skinny one - which is on the
    // chain - needed (rocprofv3 timeline at N = 8192, profiles/r03_potrf_n8192_timeline.md: 250 us per column update beside
    // the big one, 82 us ahead of it).  potrf N = 8192 6.87 -> 6.6 ms, N = 12288 15.4 -> 14.9, loglikelihood N = 7933 7.05 -> 6.8.
    // Not with riders (their rows make the column update long enough to delay the big one; C2 12.15 -> 12.28 ms).
    const bool col_first = pm == 0;
    for (int64_t r0 = 0; r0 < pn && !tall;) {
        const int64_t w = wc;                                 // super-panel k (already factored)
        const int64_t t = pn - r0 - w;                        // trailing size
        if (t <= 0) break;
        const int64_t r1 = r0 + w;
        if (t <= tail_rows && t % 128 == 0) {
            // the whole trailing matrix in one update, then one panel of t columns, everything on the main stream
            if (lookahead && panel_on_side) MGPI_HIP_TRY(hipStreamWaitEvent(main_stream, ctx->ev_panel, 0));
            panel_on_side = false;
            const double* Pt = A + r1 * lda + r0;
            mgpi_prof_begin(ctx, 0);
            MGPI_TRY(mgpi_launch_gemm(ctx, 0, 0, 1, t, t, w, Pt, lda, Pt, lda, A + r1 * lda + r1, lda));
            mgpi_prof_end(ctx, 0);
            mgpi_prof_flops(ctx, 0, (double)t * ((double)t + 1.0) * (double)w);
            MGPI_TRY(rider_update(r0, w, r1, pn));
            mgpi_prof_begin(ctx, 1);
            MGPI_TRY(factor_wide_panel(r1, t));
            mgpi_prof_end(ctx, 1);
            break;
        }
        const int64_t w1 = super_width(r1);                  // super-panel k+1
        const double* P = A + r1 * lda + r0;                  // rows r1.., the w columns of super-panel k
        const int64_t t2 = t - w1;
        if (lookahead && merged_on && t2 > 0 && t2 >= merged_min && t % 128 == 0 && w1 % 128 == 0 && pn % 128 == 0) {
            if (panel_on_side) MGPI_HIP_TRY(hipStreamWaitEvent(main_stream, ctx->ev_panel, 0));
            // The producer goes into its queue BEFORE the consumers that wait for its count: a tool that serialises dispatches
            // in submission order (rocprofv3 --pmc does) would otherwise hold this launch back behind kernels that cannot
            // start without it.
            const int cw = (int)(w1 / 128);
            mgpi_prof_begin(ctx, 0);
            MGPI_TRY(mgpi_launch_syrk_merged(ctx, t, w, P, lda, A + r1 * lda + r1, lda, cw, d_sync));
            mgpi_prof_end(ctx, 0);
            mgpi_prof_flops(ctx, 0, (double)t * ((double)t + 1.0) * (double)w);
            MGPI_TRY(rider_update(r0, w, r1 + w1, pn));      // riders, columns behind the next super-panel (main stream)
            // the first wait of a factorisation also orders the second stream behind the reset of the counter
            if (!merged_prev) MGPI_HIP_TRY(hipStreamWaitEvent(ctx->stream2, ctx->ev_upd, 0));
            MGPI_HIP_TRY(hipEventRecord(ctx->ev_upd, main_stream));
            sync_target += (unsigned)merged_column_tiles((int)(t / 128), cw);
            hipLaunchKernelGGL(wait_count_kernel, dim3(1), dim3(64), 0, ctx->stream2, d_sync, sync_target, ctx->d_info);
            MGPI_HIP_TRY(hipGetLastError());
            ctx->stream = ctx->stream2;
            // riders, the next super-panel's columns: behind the count - the launch that bumps it started after the previous
            // step's rider update on the main stream, which wrote these columns last
            int rc = rider_update(r0, w, r1, r1 + w1);
            if (rc == 0) rc = factor_super(r1, w1, true);
            if (rc == 0) rc = forward_columns(r1, r1 + w1);
            ctx->stream = main_stream;
            MGPI_TRY(rc);
            MGPI_HIP_TRY(hipEventRecord(ctx->ev_panel, ctx->stream2));
            panel_on_side = true;
            merged_prev = true;
        } else if (lookahead) {
            merged_prev = false;
            // main stream: wait for super-panel k (before ev_panel is recorded again below)
            if (panel_on_side) MGPI_HIP_TRY(hipStreamWaitEvent(main_stream, ctx->ev_panel, 0));
            // second stream, enqueued first so that it gets the first free CUs:
            //   A[r1:pn, r1:r1+w1] -= P P[0:w1]^T, then factor super-panel k+1
            MGPI_HIP_TRY(hipStreamWaitEvent(ctx->stream2, ctx->ev_upd, 0));
            ctx->stream = ctx->stream2;
            mgpi_prof_begin(ctx, 3);
            int rc = mgpi_launch_gemm(ctx, 0, 0, 0, t + pm, w1, w, P, lda, P, lda, A + r1 * lda + r1, lda);    // rider rows follow the square's
            mgpi_prof_end(ctx, 3);
            if (col_first) MGPI_HIP_TRY(hipEventRecord(ctx->ev_join, ctx->stream2));
            if (rc == 0) rc = factor_super(r1, w1, true);
            if (rc == 0) rc = forward_columns(r1, r1 + w1);
            ctx->stream = main_stream;
            MGPI_TRY(rc);
            MGPI_HIP_TRY(hipEventRecord(ctx->ev_panel, ctx->stream2));
            panel_on_side = true;
            // main stream: the rest of the trailing update (lower tiles): A[r2:pn, r2:pn] -= P[w1:] P[w1:]^T
            if (t2 > 0) {
                if (col_first) MGPI_HIP_TRY(hipStreamWaitEvent(main_stream, ctx->ev_join, 0));
                mgpi_prof_begin(ctx, 0);
                const double* P2 = P + w1 * lda;
                MGPI_TRY(mgpi_launch_gemm(ctx, 0, 0, 1, t2, t2, w, P2, lda, P2, lda, A + (r1 + w1) * lda + (r1 + w1), lda));
                mgpi_prof_end(ctx, 0);
                mgpi_prof_flops(ctx, 0, (double)t2 * ((double)t2 + 1.0) * (double)w);
                MGPI_TRY(rider_update(r0, w, r1 + w1, pn));
            }
            MGPI_HIP_TRY(hipEventRecord(ctx->ev_upd, main_stream));
        } else {
            mgpi_prof_begin(ctx, 3);
            MGPI_TRY(mgpi_launch_gemm(ctx, 0, 0, 0, t + pm, w1, w, P, lda, P, lda, A + r1 * lda + r1, lda));
            mgpi_prof_end(ctx, 3);
            mgpi_prof_begin(ctx, 1);
            MGPI_TRY(factor_super(r1, w1, false));
            mgpi_prof_end(ctx, 1);
            MGPI_TRY(forward_columns(r1, r1 + w1));
            if (t2 > 0) {
                mgpi_prof_begin(ctx, 0);
                const double* P2 = P + w1 * lda;
                MGPI_TRY(mgpi_launch_gemm(ctx, 0, 0, 1, t2, t2, w, P2, lda, P2, lda, A + (r1 + w1) * lda + (r1 + w1), lda));
                mgpi_prof_end(ctx, 0);
                mgpi_prof_flops(ctx, 0, (double)t2 * ((double)t2 + 1.0) * (double)w);
                MGPI_TRY(rider_update(r0, w, r1 + w1, pn));
            }
        }
        r0 = r1;
        wc = w1;
    }
    if (lookahead && panel_on_side) MGPI_HIP_TRY(hipStreamWaitEvent(main_stream, ctx->ev_panel, 0));
    if (fwd_used) {                  // the forward sweep reads the strip and the panels: it joins before anything below moves them
        MGPI_HIP_TRY(hipEventRecord(ctx->ev_fwd, ctx->stream3));
        MGPI_HIP_TRY(hipStreamWaitEvent(main_stream, ctx->ev_fwd, 0));
    }

    if (strip_first < ncols / IB)  // the factored diagonal blocks wait in the scratch strip
        hipLaunchKernelGGL(diag_blocks_back_kernel, dim3((unsigned)(ncols / IB - strip_first)), dim3(256), 0, ctx->stream, A, lda,
                           ctx->d_diag + ctx->diag_len, strip_first);
    mgpi_prof_end(ctx, 2);

    hipLaunchKernelGGL(logdiag_kernel, dim3(1), dim3(256), 0, ctx->stream, A, lda, ndiag, ctx->d_scal + (keep_status ? 1 : 0),
                       keep_status, keep_status ? (double*)nullptr : ctx->d_scal + 4);
    MGPI_HIP_TRY(hipGetLastError());
    K->factored = tall ? 0 : 1;          // mgpi_potrf_dev takes it back when the status says otherwise
    join_on_error.armed = false;
    return 0;
}

// the status of the factorisation just enqueued: one read-back of the failing-pivot word and sum log L_ii
static int potrf_status(mgpi_ctx* ctx, mgpi_mat* K, double* logdet) {
    int info = 0;
    double sc[6] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0};          // [0] sum log L_ii, [4] min L_ii, [5] max L_ii
    if (ctx->h_pinned) {
        char* land = (char*)ctx->h_pinned;
        MGPI_HIP_TRY(hipMemcpyAsync(land, ctx->d_info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        MGPI_HIP_TRY(hipMemcpyAsync(land + 8, ctx->d_scal, sizeof(sc), hipMemcpyDeviceToHost, ctx->stream));
        MGPI_HIP_TRY(hipStreamSynchronize(ctx->stream));
        memcpy(&info, land, sizeof(int));
        memcpy(sc, land + 8, sizeof(sc));
    } else {
        MGPI_HIP_TRY(hipMemcpyAsync(&info, ctx->d_info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
        MGPI_HIP_TRY(hipMemcpyAsync(sc, ctx->d_scal, sizeof(sc), hipMemcpyDeviceToHost, ctx->stream));
        MGPI_HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    const double ld = sc[0];
    K->diag_min = (sc[4] > 0.0 && sc[4] <= sc[5] && sc[5] < INFINITY) ? sc[4] : 0.0;
    K->diag_max = K->diag_min > 0.0 ? sc[5] : 0.0;
    if (info < 0) {
        K->factored = 0;
        mgpi_set_error("mgpi_potrf_dev: the trailing update's progress counter was not reached (internal error)");
        return MGPI_ERR_HIP;
    }
    if (info != 0) {
        K->factored = 0;
        mgpi_set_error("covariance is not positive definite! (pivot %d is not above %.3g times its original diagonal entry)", info,
                       ctx->pivot_tol);
        return info;
    }
    if (logdet) *logdet = 2.0 * ld;
    return 0;
}

extern "C" int mgpi_potrf_dev(mgpi_ctx* ctx, mgpi_mat* K, double* logdet) {
    MGPI_TRY(potrf_enqueue(ctx, K, 0, 0));
    return potrf_status(ctx, K, logdet);
}

static int trsv_sweeps(mgpi_ctx* ctx, const mgpi_mat* Lm, double* v, int which);

// Factor and solve in one call (LAPACK's dposv):  K <- L,  B <- K^-1 B.  What compress / condition / fit do with
// np.linalg.inv(K) @ f in the reference (mgpi/interpolators.py:92-100, :179-189).  For ONE right-hand side the forward sweep
// L y = f runs beside the factorisation on a third stream (potrf_enqueue, fwd_rhs), so the solve costs the backward sweep
// only; the arithmetic is that of mgpi_potrf_dev followed by mgpi_potrs_dev, operation for operation.
extern "C" int mgpi_posv_dev(mgpi_ctx* ctx, mgpi_mat* K, mgpi_mat* B, double* logdet) {
    MGPI_REQUIRE(ctx && K && B, "mgpi_posv_dev: null pointer");
    MGPI_REQUIRE(B->rows == K->rows, "mgpi_posv_dev: B has %lld rows, K is %lld", (long long)B->rows, (long long)K->rows);
    MGPI_REQUIRE(B->prows >= K->prows || B->prows >= mgpi_roundup(K->rows, IB), "mgpi_posv_dev: B row padding smaller than K's");
    if (!(B->cols == 1 && B->ld == 1)) {
        MGPI_TRY(mgpi_potrf_dev(ctx, K, logdet));
        return mgpi_potrs_dev(ctx, K, B);
    }
    MGPI_TRY(potrf_enqueue(ctx, K, 0, 0, B->d));
    MGPI_TRY(potrf_status(ctx, K, logdet));
    mgpi_prof_begin(ctx, 6);
    const int rc = trsv_sweeps(ctx, K, B->d, 2);
    mgpi_prof_end(ctx, 6);
    return rc;
}

// Factor, solve AND forward-solve the riders in one call.  KR is a wide matrix of n columns (padded pn) and pn + m rows: rows
// [0, n) hold K (lower tiles), rows [pn, pn + m) the riders R = K(x*, x) (mgpi_cov_riders_dev builds both).  On return the top
// square is L, the rider rows hold R L^-T = V^T with V = L^-1 K(x, x*) - the matrix whose column norms the posterior variance
// subtracts (mgpi/interpolators.py:199: cov = K** - K* K^-1 K*^T, diagonal only) - and B (may be NULL) holds K^-1 B.  The riders
// are solved by the panels' own substitutions and updated inside the trailing-update launches (potrf_enqueue, rider_rows).
extern "C" int mgpi_posv_riders_dev(mgpi_ctx* ctx, mgpi_mat* KR, mgpi_mat* B, double* logdet) {
    MGPI_REQUIRE(ctx && KR, "mgpi_posv_riders_dev: null pointer");
    const int64_t pn = KR->pcols, pm = KR->prows - KR->pcols;
    MGPI_REQUIRE(pm >= MGPI_TILE && pm % MGPI_TILE == 0 && pn % MGPI_TILE == 0,
                 "mgpi_posv_riders_dev: need a (pn + pm) x pn matrix with pm >= %d rider rows (padded %lld x %lld)", MGPI_TILE,
                 (long long)KR->prows, (long long)KR->pcols);
    double* rhs = nullptr;
    if (B != nullptr) {
        MGPI_REQUIRE(B->rows == KR->cols && B->cols == 1 && B->ld == 1 && B->prows >= pn, "mgpi_posv_riders_dev: B must be an (n x 1) vector");
        rhs = B->d;
    }
    MGPI_TRY(potrf_enqueue(ctx, KR, 0, 0, rhs, 0, nullptr, pm));
    MGPI_TRY(potrf_status(ctx, KR, logdet));
    if (rhs == nullptr) return 0;
    // the backward sweep runs on the top square only: a non-owning square handle over it
    mgpi_mat L = *KR;
    L.rows = L.cols = KR->cols;
    L.prows = L.pcols = pn;
    L.owns = 0;
    L.dinv = nullptr;
    mgpi_prof_begin(ctx, 6);
    const int rc = trsv_sweeps(ctx, &L, rhs, 2);
    mgpi_prof_end(ctx, 6);
    if (L.dinv) mgpi_pool_free(ctx, L.dinv, L.dinv_bytes);       // the inverted diagonal blocks the sweep built on the temporary handle
    return rc;
}

// The multi-GPU driver factors one nb x nb diagonal block per block step; reading the status back after each of them
// would put a host round trip on the critical path of every step (round 1: 217 ms against 189 at N = 32768).  These three
// keep the status on the device: reset once, enqueue the block factorisations, read once.
static int diag_ref_ok(const mgpi_mat* K, const mgpi_mat* diag0, const double** out) {
    *out = nullptr;
    if (diag0 == nullptr) return 0;
    MGPI_REQUIRE(diag0->ld == 1 && diag0->prows >= K->pcols, "pivot reference: need a vector with one entry per column (%lld), got %lld",
                 (long long)K->pcols, (long long)diag0->prows);
    *out = diag0->d;
    return 0;
}

extern "C" int mgpi_potrf_enqueue_dev(mgpi_ctx* ctx, mgpi_mat* K, int64_t pivot_base, const mgpi_mat* diag0) {
    MGPI_REQUIRE(ctx && K, "mgpi_potrf_enqueue_dev: null pointer");
    MGPI_REQUIRE(pivot_base >= 0 && pivot_base < (int64_t)2147483647 - 70000, "mgpi_potrf_enqueue_dev: pivot_base out of range");
    const double* ref = nullptr;
    MGPI_TRY(diag_ref_ok(K, diag0, &ref));
    return potrf_enqueue(ctx, K, pivot_base, 1, nullptr, 0, ref);
}

// the tall form (see potrf_enqueue): panel = an (m x w) block column whose top w x w block is the diagonal block
extern "C" int mgpi_potrf_panel_enqueue_dev(mgpi_ctx* ctx, mgpi_mat* panel, int64_t pivot_base, const mgpi_mat* diag0) {
    MGPI_REQUIRE(ctx && panel, "mgpi_potrf_panel_enqueue_dev: null pointer");
    MGPI_REQUIRE(pivot_base >= 0 && pivot_base < (int64_t)2147483647 - 70000, "mgpi_potrf_panel_enqueue_dev: pivot_base out of range");
    const double* ref = nullptr;
    MGPI_TRY(diag_ref_ok(panel, diag0, &ref));
    return potrf_enqueue(ctx, panel, pivot_base, 1, nullptr, 1, ref);
}

extern "C" int mgpi_ctx_status_reset(mgpi_ctx* ctx) {
    MGPI_REQUIRE(ctx != nullptr, "mgpi_ctx_status_reset: null ctx");
    MGPI_HIP_TRY(hipSetDevice(ctx->device));
    MGPI_HIP_TRY(hipMemsetAsync(ctx->d_info, 0, sizeof(int), ctx->stream));
    MGPI_HIP_TRY(hipMemsetAsync(ctx->d_scal + 1, 0, sizeof(double), ctx->stream));
    return 0;
}

// *pivot: smallest 1-based global index of a failing pivot since the reset (0 = none); *logdiag: sum of log L_ii over
// every block factored since the reset.  Synchronises ctx's stream.
extern "C" int mgpi_ctx_status_read(mgpi_ctx* ctx, int64_t* pivot, double* logdiag) {
    MGPI_REQUIRE(ctx != nullptr, "mgpi_ctx_status_read: null ctx");
    MGPI_HIP_TRY(hipSetDevice(ctx->device));
    int info = 0;
    double ld = 0.0;
    MGPI_HIP_TRY(hipMemcpyAsync(&info, ctx->d_info, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    MGPI_HIP_TRY(hipMemcpyAsync(&ld, ctx->d_scal + 1, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    MGPI_HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (info < 0) {
        mgpi_set_error("mgpi_ctx_status_read: a trailing update's progress counter was not reached (internal error)");
        return MGPI_ERR_HIP;
    }
    if (pivot) *pivot = info;
    if (logdiag) *logdiag = ld;
    return 0;
}

// ---- batched log marginal likelihood at fixed x -------------------------------------------
// The reference's hyper-parameter loops call loglikelihood once per optimiser step / per walker
// (mgpi/interpolators.py:369-387, :474-478, :513-514, :587-601): same x, same f, B different kernel parameters, N in the
// hundreds to low thousands - a regime where one factorisation is a chain of ~N/64 latency-bound launches and the GPU is
// otherwise empty.  Here B problems share every launch (grid.y / grid.z = problem):
//   1. B lower kernel matrices in one launch, kernel descriptions read from device memory;
//   2. f is appended to every matrix as one more row block, so that the panel steps' row substitution leaves
//      y = L^-1 f in it - no separate triangular solve:  f^T K^-1 f = y^T y;
//   3. the whole matrix is ONE panel: N/64 launches of panel_step_kernel (deferred right-looking update on MFMA, 64 x 64
//      Cholesky, substitution), every launch covering all B problems;
//   4. one launch reduces sum log L_ii and y^T y per problem.
// A problem whose matrix is not positive definite reports its pivot in info[b] and -inf in out[b]; the others are unaffected.
#define MGPI_LL_BATCH_MAX_PN 4096

// per problem: diag0 = diagonal of the fresh matrix; rows pn .. pn+63 := [f^T; 0 ...]
__global__ __launch_bounds__(256) void ll_prepare_kernel(double* __restrict__ A, int64_t lda, int64_t strideA, int64_t n, int64_t pn,
                                                         const double* __restrict__ f, double* __restrict__ diag0,
                                                         int* __restrict__ info) {
    double* Ab = A + (int64_t)blockIdx.z * strideA;
    const int64_t j = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int r = blockIdx.y;                                         // 0..63: row pn + r; 64: the diagonal
    if (j == 0 && r == 0) info[blockIdx.z] = 0;
    if (j >= pn) return;
    if (r < IB) Ab[(pn + r) * lda + j] = (r == 0 && j < n) ? f[j] : 0.0;
    else diag0[(int64_t)blockIdx.z * pn + j] = Ab[j * lda + j];
}

// out[b] = -1/2 y^T y - sum log L_ii - n/2 log(2 pi), or -inf when info[b] != 0; L_ii from the scratch strip
__global__ __launch_bounds__(256) void ll_finish_kernel(const double* __restrict__ A, int64_t lda, int64_t strideA, int64_t n, int64_t pn,
                                                        const double* __restrict__ S, const int* __restrict__ info,
                                                        double* __restrict__ out) {
    __shared__ double part[2][256];
    const double* y = A + (int64_t)blockIdx.x * strideA + pn * lda;
    const double* Sb = S + (int64_t)blockIdx.x * pn * IB;
    double q = 0.0, l = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) {
        q += y[i] * y[i];
        l += log(Sb[i * IB + (i % IB)]);
    }
    part[0][threadIdx.x] = q;
    part[1][threadIdx.x] = l;
    __syncthreads();
    for (int w = 128; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) {
            part[0][threadIdx.x] += part[0][threadIdx.x + w];
            part[1][threadIdx.x] += part[1][threadIdx.x + w];
        }
        __syncthreads();
    }
    if (threadIdx.x == 0)
        out[blockIdx.x] = (info[blockIdx.x] != 0) ? -INFINITY : -0.5 * part[0][0] - part[1][0] - 0.5 * (double)n * 1.8378770664093453;
}


// *out = sum_i v[i]^2, one workgroup (the quadratic form of the sequential log-likelihood path)
__global__ __launch_bounds__(256) void sumsq_kernel(const double* __restrict__ v, int64_t n, double* __restrict__ out) {
    __shared__ double part[256];
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) acc = fma(v[i], v[i], acc);
    part[threadIdx.x] = acc;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) part[threadIdx.x] += part[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = part[0];
}

extern "C" int mgpi_loglikelihood_batch(mgpi_ctx* ctx, const mgpi_mat* X, const mgpi_mat* f, const mgpi_kernel_desc* terms, int nterms,
                                        int batch, double* out, int32_t* info) {
    MGPI_REQUIRE(ctx && X && f && terms && out, "mgpi_loglikelihood_batch: null pointer");
    MGPI_REQUIRE(batch >= 1 && batch <= 4096, "mgpi_loglikelihood_batch: batch %d outside [1, 4096]", batch);
    MGPI_REQUIRE(f->rows == X->rows && f->ld == 1, "mgpi_loglikelihood_batch: f must be an (n x 1) vector matching X");
    const int64_t n = X->rows;
    const int d = (int)X->cols;
    const int64_t pn = mgpi_roundup(n, IB);
    MGPI_HIP_TRY(hipSetDevice(ctx->device));
    std::vector<mgpi_terms_dev> tb((size_t)batch);
    for (int b = 0; b < batch; ++b) MGPI_TRY(mgpi_make_terms(terms + (size_t)b * nterms, nterms, d, &tb[b]));
    if (pn > MGPI_LL_BATCH_MAX_PN) {
        // large problems fill the chip on their own: one after the other through the blocked factorisation, x still resident
        mgpi_mat *K = nullptr, *a = nullptr;
        int st = mgpi_mat_alloc_wide(ctx, n, n, &K);
        if (st == 0) st = mgpi_mat_alloc(ctx, n, 1, &a);
        for (int b = 0; b < batch && st == 0; ++b) {
            double logdet = 0.0;
            st = mgpi_cov_matrix_dev(ctx, X, nullptr, terms + (size_t)b * nterms, nterms, 1, K);
            // f^T K^-1 f = |L^-1 f|^2: the forward sweep only, riding beside the factorisation (potrf_enqueue, fwd_rhs)
            if (st == 0) st = mgpi_mat_copy(ctx, f, a);
            if (st == 0) st = potrf_enqueue(ctx, K, 0, 0, a->d);
            if (st == 0) st = potrf_status(ctx, K, &logdet);
            if (st > 0) {                                   // not positive definite: this member only
                if (info) info[b] = st;
                out[b] = -INFINITY;
                st = 0;
                continue;
            }
            if (st == 0) {                                  // the sum of squares on the device, one scalar back
                double q = 0.0;
                double* dq = ctx->d_scal + 3;
                hipLaunchKernelGGL(sumsq_kernel, dim3(1), dim3(256), 0, ctx->stream, a->d, n, dq);
                double* land = ctx->h_pinned ? (double*)ctx->h_pinned : &q;
                hipError_t e = hipGetLastError();                 // a HIP error must not skip the two frees below
                if (e == hipSuccess) e = hipMemcpyAsync(land, dq, sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
                if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);
                if (e != hipSuccess) {
                    mgpi_set_error("mgpi_loglikelihood_batch: %s (sequential path, member %d)", hipGetErrorString(e), b);
                    st = MGPI_ERR_HIP;
                    break;
                }
                q = *land;
                out[b] = -0.5 * q - 0.5 * logdet - 0.5 * (double)n * 1.8378770664093453;
                if (info) info[b] = 0;
            }
        }
        mgpi_mat_free(K);
        mgpi_mat_free(a);
        return st;
    }
    // The batched form holds (pn + 64) x lda doubles per member: large walker counts are processed in chunks that fit the free
    // device memory (at batch = 4096, pn = 4096 one block would be 558 GB) instead of failing on the allocation.
    {
        const size_t lda0 = (size_t)mgpi_roundup(pn, MGPI_TILE);
        const size_t per = ((size_t)(pn + IB) * lda0 + (size_t)pn * IB + (size_t)pn) * sizeof(double) + sizeof(mgpi_terms_dev) + 64;
        size_t free_b = 0, total_b = 0;
        MGPI_HIP_TRY(hipMemGetInfo(&free_b, &total_b));
        size_t budget = free_b / 2 + (ctx->pool ? ctx->pool->held : 0);
        static const size_t cap = getenv("MGPI_LL_CHUNK_BYTES") ? (size_t)atoll(getenv("MGPI_LL_CHUNK_BYTES")) : 0;   // tests: force chunking
        if (cap > 0 && cap < budget) budget = cap;
        int chunk = (int)(budget / per);
        if (chunk < 1) chunk = 1;
        if (chunk < batch) {
            for (int b0 = 0; b0 < batch; b0 += chunk) {
                const int cnt = batch - b0 < chunk ? batch - b0 : chunk;
                MGPI_TRY(mgpi_loglikelihood_batch(ctx, X, f, terms + (size_t)b0 * nterms, nterms, cnt, out + b0, info ? info + b0 : nullptr));
            }
            return 0;
        }
    }
    const int64_t lda = mgpi_roundup(pn, MGPI_TILE);
    const int64_t strideA = (pn + IB) * lda;
    const size_t bytesA = (size_t)batch * strideA * sizeof(double);
    const size_t bytesS = (size_t)batch * pn * IB * sizeof(double);
    const size_t bytesD = (size_t)batch * pn * sizeof(double);
    const size_t bytesT = mgpi_roundup((int64_t)(batch * sizeof(mgpi_terms_dev)), 256);
    const size_t bytesO = mgpi_roundup((int64_t)(batch * sizeof(double)), 256), bytesI = mgpi_roundup((int64_t)(batch * sizeof(int)), 256);
    double* blk = nullptr;
    size_t got = 0;
    MGPI_TRY(mgpi_pool_alloc(ctx, bytesA + bytesS + bytesD + bytesT + bytesO + bytesI, &blk, &got));
    struct release { mgpi_ctx* c; double* p; size_t b; ~release() { mgpi_pool_free(c, p, b); } } guard{ctx, blk, got};
    char* wptr = (char*)blk;
    double* A = (double*)wptr;             wptr += bytesA;
    double* S = (double*)wptr;             wptr += bytesS;
    double* D0 = (double*)wptr;            wptr += bytesD;
    mgpi_terms_dev* T = (mgpi_terms_dev*)wptr;  wptr += bytesT;
    double* O = (double*)wptr;             wptr += bytesO;
    int* I = (int*)wptr;
    if (batch == 1) {
        // one problem: the description travels as a kernel argument (no upload) and the specialised build applies
        MGPI_TRY(mgpi_launch_cov(ctx, tb[0], X->d, X->ld, n, pn, X->d, X->ld, n, pn, 1, 1, 1, A, lda));
    } else {
        MGPI_HIP_TRY(hipMemcpyAsync(T, tb.data(), (size_t)batch * sizeof(mgpi_terms_dev), hipMemcpyHostToDevice, ctx->stream));
        MGPI_TRY(mgpi_launch_cov_batch(ctx, T, d, X->d, X->ld, n, pn, A, lda, strideA, batch));
    }
    hipLaunchKernelGGL(ll_prepare_kernel, dim3((unsigned)((pn + 255) / 256), IB + 1, (unsigned)batch), dim3(256), 0, ctx->stream, A, lda,
                       strideA, n, pn, f->d, D0, I);
    for (int64_t c0 = 0; c0 < pn; c0 += IB) {
        const int has_prev = c0 > 0;
        const int nfused = (int)(1 + (pn + IB - c0 - IB) / IB);            // the f block is one more row block below
        const int rest_cols = has_prev ? (int)((pn - c0 - IB) / IB) : 0;
        int64_t nrest = 0;
        for (int tj = 0; tj < rest_cols; ++tj) nrest += nfused - 1 - tj;
        hipLaunchKernelGGL(panel_step_kernel, dim3((unsigned)(nfused + nrest), (unsigned)batch), dim3(256), 0, ctx->stream, A, lda, c0,
                           has_prev, nfused, rest_cols, c0, D0 + c0, I, S + c0 * IB, ctx->pivot_tol, 0, (double*)nullptr, strideA,
                           pn * IB, pn);
    }
    hipLaunchKernelGGL(ll_finish_kernel, dim3((unsigned)batch), dim3(256), 0, ctx->stream, A, lda, strideA, n, pn, S, I, O);
    MGPI_HIP_TRY(hipGetLastError());
    // values and status words sit next to each other: one read-back
    const size_t back_bytes = bytesO + (size_t)batch * sizeof(int);
    std::vector<char> pageable;
    char* back = (char*)ctx->h_pinned;
    if (back == nullptr || back_bytes > MGPI_PINNED_BYTES) {
        pageable.resize(back_bytes);
        back = pageable.data();
    }
    MGPI_HIP_TRY(hipMemcpyAsync(back, O, back_bytes, hipMemcpyDeviceToHost, ctx->stream));
    MGPI_HIP_TRY(hipStreamSynchronize(ctx->stream));
    memcpy(out, back, (size_t)batch * sizeof(double));
    if (info) memcpy(info, back + bytesO, (size_t)batch * sizeof(int));
    return 0;
}

// ---- multi-rhs triangular solves:  L X = B (trans = 0),  L^T X = B (trans = 1),  B is (n x m) wide ------
// Xout (forward solve with the inverse diagonal blocks only): the SOLVED blocks are written there instead of over B - rows
// of B are dead once their block is solved, so nothing is copied back, and the product with the inverse block starts from
// zero accumulators instead of a cleared scratch tile (C5: a memset and a copy of 512 x 37 k doubles per block, 14 % of the
// variance time, gone).  B and Xout must not overlap.
static int trsm_blocked(mgpi_ctx* ctx, const mgpi_mat* Lm, double* B, int64_t ldb, int64_t ncols, int trans, int use_inv = 0,
                        double* Xout = nullptr) {
    const int64_t pn = Lm->prows <= Lm->pcols ? Lm->prows : mgpi_roundup(Lm->rows, IB);
    const int64_t lda = Lm->ld, nb = ctx->nb;
    const double* L = Lm->d;
    const int64_t nc = mgpi_roundup(ncols, 64);
    if (!trans) {
        // the variance path may replace the chain of a diagonal block solve by one gemm with the block's inverse
        double* T = nullptr;
        size_t tbytes = 0;
        const bool inv = use_inv && pn > nb;
        if (inv) MGPI_TRY(ensure_dinv(ctx, Lm));
        if (inv && Xout == nullptr) MGPI_TRY(mgpi_pool_alloc(ctx, (size_t)nb * (size_t)nc * sizeof(double), &T, &tbytes));
        struct scratch_guard {      // T goes back to the pool on every way out (reuse is ordered on the main stream, which
            mgpi_ctx* c;            // has joined the second one by then - or the call failed and nothing runs on T any more)
            double*& p;
            size_t& b;
            ~scratch_guard() { if (p) mgpi_pool_free(c, p, b); }
        } release_T{ctx, T, tbytes};
        // where later updates read the solved rows of a block from
        const double* X = (inv && Xout != nullptr) ? Xout : B;
        // rows [r0, r0+w) of B <- L_bb^-1 (rows of B): 64-wide inner blocks, left-looking inside the outer block
        auto solve_diag_block = [&](int64_t r0, int64_t w) -> int {
            if (inv) {
                // X_b = 0 - (-(L_bb)^-1) B_b; full-size tiles whatever the stream
                const int lean = ctx->lean;
                ctx->lean = 0;
                ctx->gemm_zero_c = 1;
                double* dst = Xout ? Xout + r0 * ldb : T;
                int rc = mgpi_launch_gemm(ctx, 0, 1, 0, w, nc, w, mgpi_dinv_block(Lm, r0 / nb), nb, B + r0 * ldb, ldb, dst, Xout ? ldb : nc);
                ctx->gemm_zero_c = 0;
                ctx->lean = lean;
                if (rc) return rc;
                if (!Xout)
                    hipLaunchKernelGGL(copy_rows_kernel, dim3((unsigned)((nc / 2 + 255) / 256), (unsigned)w), dim3(256), 0, ctx->stream,
                                       T, nc, B + r0 * ldb, ldb, w, nc);
                return 0;
            }
            for (int64_t c0 = r0; c0 < r0 + w; c0 += IB) {
                const int64_t kk = c0 - r0;
                if (kk > 0)   // B[c0:c0+64,:] -= L[c0:c0+64, r0:c0] * B[r0:c0,:]
                    MGPI_TRY(mgpi_launch_gemm(ctx, 0, 1, 0, IB, nc, kk, L + c0 * lda + r0, lda, B + r0 * ldb, ldb,
                                              B + c0 * ldb, ldb));
                MGPI_TRY(mgpi_launch_trsm_cols(ctx, L + c0 * lda + c0, lda, B + c0 * ldb, ldb, ncols, 0));
            }
            return 0;
        };
        // look-ahead as in the factorisation: the update of the next (super-)block's rows and then their solve run on
        // the second stream (the solve with lean kernels) while the update of all the rows below runs on the main
        // stream at the same time - disjoint rows of B, both only read the solved rows of block k.
        // Two-level blocking, as in the factorisation: with the inverse diagonal blocks the solve of a block is cheap,
        // so while many rows remain the big update is applied once per SUPER-BLOCK of MGPI_TRSM_SUPER (default 4)
        // blocks, i.e. with a contraction of 2048 (NN gemm alone at 32768 x 4096: 62 / 69 / 71 TFLOP/s for K = 512 / 1024 /
        // 2048).  C3 variance: 72.5 ms with 1 block per update, 71.1 with 2, 69.8 with 4, 72.1 with 8, 76.0 with 16.
        static const int64_t super_blocks = getenv("MGPI_TRSM_SUPER") ? atoll(getenv("MGPI_TRSM_SUPER")) : 4;
        // [r2b] the threshold below which the width drops back to one block: 8192 -> 2048 rows (C2 variance 6.02 -> 5.61 ms, C5
        // 2.708 -> 2.671 s, C3 unchanged 70.6 -> 70.4 ms)
        static const int64_t super_min = getenv("MGPI_TRSM_SUPER_MIN") ? atoll(getenv("MGPI_TRSM_SUPER_MIN")) : 2048;
        auto super_width = [&](int64_t r0) -> int64_t {
            const int64_t left = pn - r0;
            int64_t wc = (inv && r0 > 0 && left - super_blocks * nb >= super_min) ? super_blocks * nb : nb;
            return wc < left ? wc : left;
        };
        // rows [r0, r0 + wc) of B <- (L restricted to that diagonal super-block)^-1 B: block by block, right-looking
        auto solve_super = [&](int64_t r0, int64_t wc, bool on_side) -> int {
            for (int64_t p0 = r0; p0 < r0 + wc; p0 += nb) {
                const int64_t w = (r0 + wc - p0 < nb) ? (r0 + wc - p0) : nb;
                ctx->lean = on_side ? 1 : 0;
                int rc = solve_diag_block(p0, w);
                ctx->lean = 0;
                if (rc) return rc;
                const int64_t p1 = p0 + w, rem = r0 + wc - p1;
                if (rem > 0) {  // B[p1:r0+wc,:] -= L[p1:r0+wc, p0:p1] * B[p0:p1,:]
                    mgpi_prof_begin(ctx, 4);     // the update launches of BOTH streams are timed: with deep super-blocks a good part
                    int rcu = mgpi_launch_gemm(ctx, 0, 1, 0, rem, nc, w, L + p1 * lda + p0, lda, X + p0 * ldb, ldb, B + p1 * ldb, ldb);
                    mgpi_prof_end(ctx, 4);       // of the solve's flops runs inside the super-block, on the look-ahead stream
                    mgpi_prof_flops(ctx, 4, 2.0 * (double)rem * (double)nc * (double)w);
                    MGPI_TRY(rcu);
                }
            }
            return 0;
        };
        hipStream_t main_stream = ctx->stream;
        mgpi_stream_join_guard join_on_error{ctx, main_stream, true};
        // [r3] With the inverted diagonal blocks the solve of a (super-)block is a few gemms, no chain worth hiding: run side by
        // side, the next block's solve and the big update slow each other down by more than the overlap saves (C3 variance
        // 70.7 ms with the look-ahead, 69.0 without; N = 8192: 5.83 / 5.58 - tools/variance_lookahead_probe.py).  The look-ahead
        // stays for the substitution form (MGPI_VAR_DINV=0); MGPI_TRSM_LOOKAHEAD=1 / 0 forces it on / off.
        static const int trsm_la_env = getenv("MGPI_TRSM_LOOKAHEAD") ? atoi(getenv("MGPI_TRSM_LOOKAHEAD")) : -1;
        const bool lookahead = ctx->stream2 != nullptr && !ctx->no_lookahead && (trsm_la_env >= 0 ? trsm_la_env != 0 : !inv);
        int64_t wc = super_width(0);
        MGPI_TRY(solve_super(0, wc, false));
        if (lookahead) MGPI_HIP_TRY(hipEventRecord(ctx->ev_upd, main_stream));
        bool block_on_side = false;
        for (int64_t r0 = 0; r0 < pn;) {
            const int64_t w = wc;
            const int64_t t = pn - r0 - w;
            if (t <= 0) break;
            const int64_t r1 = r0 + w;
            const int64_t w1 = super_width(r1);
            const int64_t t2 = t - w1;
            if (lookahead) {
                if (block_on_side) MGPI_HIP_TRY(hipStreamWaitEvent(main_stream, ctx->ev_panel, 0));   // rows of block k solved
                // second stream: B[r1:r1+w1,:] -= L[r1:r1+w1, r0:r0+w] * B[r0:r0+w,:], then solve that (super-)block
                MGPI_HIP_TRY(hipStreamWaitEvent(ctx->stream2, ctx->ev_upd, 0));                        // update k-1 of these rows done
                ctx->stream = ctx->stream2;
                mgpi_prof_begin(ctx, 4);
                int rc = mgpi_launch_gemm(ctx, 0, 1, 0, w1, nc, w, L + r1 * lda + r0, lda, X + r0 * ldb, ldb, B + r1 * ldb, ldb);
                mgpi_prof_end(ctx, 4);
                mgpi_prof_flops(ctx, 4, 2.0 * (double)w1 * (double)nc * (double)w);
                if (rc == 0) rc = solve_super(r1, w1, true);
                ctx->stream = main_stream;
                MGPI_TRY(rc);
                MGPI_HIP_TRY(hipEventRecord(ctx->ev_panel, ctx->stream2));
                block_on_side = true;
                // main stream: the remaining rows  B[r1+w1:,:] -= L[r1+w1:, r0:r0+w] * B[r0:r0+w,:]
                if (t2 > 0) {
                    mgpi_prof_begin(ctx, 4);
                    MGPI_TRY(mgpi_launch_gemm(ctx, 0, 1, 0, t2, nc, w, L + (r1 + w1) * lda + r0, lda, X + r0 * ldb, ldb,
                                              B + (r1 + w1) * ldb, ldb));
                    mgpi_prof_end(ctx, 4);
                    mgpi_prof_flops(ctx, 4, 2.0 * (double)t2 * (double)nc * (double)w);
                }
                MGPI_HIP_TRY(hipEventRecord(ctx->ev_upd, main_stream));
            } else {
                // one stream: ALL the rows below super-block k in one launch, then the next super-block's solve
                mgpi_prof_begin(ctx, 4);
                MGPI_TRY(mgpi_launch_gemm(ctx, 0, 1, 0, t, nc, w, L + r1 * lda + r0, lda, X + r0 * ldb, ldb, B + r1 * ldb, ldb));
                mgpi_prof_end(ctx, 4);
                mgpi_prof_flops(ctx, 4, 2.0 * (double)t * (double)nc * (double)w);
                MGPI_TRY(solve_super(r1, w1, false));
            }
            r0 = r1;
            wc = w1;
        }
        if (lookahead && block_on_side) MGPI_HIP_TRY(hipStreamWaitEvent(main_stream, ctx->ev_panel, 0));
        join_on_error.armed = false;
    } else {
        const int64_t nblk = (pn + nb - 1) / nb;
        for (int64_t b = nblk - 1; b >= 0; --b) {
            const int64_t r0 = b * nb;
            const int64_t w = (pn - r0 < nb) ? (pn - r0) : nb;
            for (int64_t c0 = r0 + w - IB; c0 >= r0; c0 -= IB) {
                const int64_t below = r0 + w - (c0 + IB);
                if (below > 0)   // B[c0:c0+64,:] -= L[c0+64:r0+w, c0:c0+64]^T * B[c0+64:r0+w,:]
                    MGPI_TRY(mgpi_launch_gemm(ctx, 1, 1, 0, IB, nc, below, L + (c0 + IB) * lda + c0, lda,
                                              B + (c0 + IB) * ldb, ldb, B + c0 * ldb, ldb));
                MGPI_TRY(mgpi_launch_trsm_cols(ctx, L + c0 * lda + c0, lda, B + c0 * ldb, ldb, ncols, 1));
            }
            if (r0 > 0)          // B[0:r0,:] -= L[r0:r0+w, 0:r0]^T * B[r0:r0+w,:]
                MGPI_TRY(mgpi_launch_gemm(ctx, 1, 1, 0, r0, nc, w, L + r0 * lda, lda, B + r0 * ldb, ldb, B, ldb));
        }
    }
    return 0;
}

// Xout != nullptr: the solution is delivered there IF the inverse diagonal blocks are in use (returns 1 in *in_xout), else in
// place in B as always
int mgpi_trsm_raw(mgpi_ctx* ctx, const mgpi_mat* L, double* B, int64_t ldb, int64_t ncols, int trans, int use_inv, double* Xout,
                  int* in_xout) {
    static const int allow = getenv("MGPI_VAR_DINV") ? atoi(getenv("MGPI_VAR_DINV")) : 1;     // developer switches
    static const int split = getenv("MGPI_VAR_SPLIT") ? atoi(getenv("MGPI_VAR_SPLIT")) : 1;
    const int64_t pn = L->prows <= L->pcols ? L->prows : mgpi_roundup(L->rows, IB);
    const bool to_x = use_inv && allow && split && Xout != nullptr && !trans && pn > ctx->nb;
    if (in_xout) *in_xout = to_x ? 1 : 0;
    return trsm_blocked(ctx, L, B, ldb, ncols, trans, use_inv && allow, to_x ? Xout : nullptr);
}

extern "C" int mgpi_trsm_dev(mgpi_ctx* ctx, const mgpi_mat* L, mgpi_mat* B, int trans) {
    MGPI_REQUIRE(ctx && L && B, "mgpi_trsm_dev: null pointer");
    MGPI_REQUIRE(L->factored, "mgpi_trsm_dev: L has not been factored by mgpi_potrf_dev");
    if (!L->factored) return MGPI_ERR_STATE;
    MGPI_REQUIRE(B->rows == L->rows, "mgpi_trsm_dev: B has %lld rows, L is %lld", (long long)B->rows, (long long)L->rows);
    MGPI_REQUIRE(B->pcols % 64 == 0 && B->prows >= L->prows, "mgpi_trsm_dev: B must be a wide tile-padded matrix");
    MGPI_HIP_TRY(hipSetDevice(ctx->device));
    return trsm_blocked(ctx, L, B->d, B->ld, B->pcols, trans);
}

// ---- single right-hand side: HBM-bound blocked trsv ------------------------------------------------------
// forward, diagonal nb-block: y_b = L_bb^-1 f_b.  One workgroup; 64-wide sub-blocks; the solved part of y in LDS.
// Per sub-block: (a) gemv with the already solved entries - lane = row, the 4 waves split the columns, each
// lane streams its own row 16 bytes at a time (the 4 waves together consume whole 128-byte lines);
// (b) 64-step substitution by wave 0 with the 64x64 diagonal tile in LDS and lane broadcasts by v_readlane.
__global__ __launch_bounds__(256) void trsv_diag_fwd_kernel(const double* __restrict__ L, int64_t lda,
                                                            double* __restrict__ y, int w, const double* __restrict__ Dg, int64_t drs, int dcm) {
    __shared__ double ys[4096];
    __shared__ double Ls[IB][IB + 1];
    __shared__ double part[4][IB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < w; i += 256) ys[i] = y[i];
    for (int c0 = 0; c0 < w; c0 += IB) {
#pragma unroll
        for (int u = 0; u < IB * IB / 256; ++u) {   // 16 independent loads in flight
            const int idx = tid + 256 * u;
            Ls[idx >> 6][idx & 63] = Dg[(int64_t)(c0 + (idx >> 6)) * drs + dcm * c0 + (idx & 63)];
        }
        __syncthreads();
        const double* row = L + (int64_t)(c0 + lane) * lda;
        double s = 0.0;
        for (int m0 = 2 * wave; m0 < c0; m0 += 64) {   // c0 is a multiple of 64: exactly 8 loads per trip, all in flight
            double2 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const double2*>(row + m0 + 8 * u);
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u].x * ys[m0 + 8 * u] + v[u].y * ys[m0 + 8 * u + 1];
        }
        part[wave][lane] = s;
        __syncthreads();
        if (wave == 0) {
            double b = ys[c0 + lane] - (part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane]);
            const double rd = 1.0 / Ls[lane][lane];
            // rolled loop: k lives in an SGPR and indexes v_readlane directly, so the 64-step chain is
            // ~5 dependent VALU ops per step with no LDS round trip
#pragma unroll 4
            for (int k = 0; k < IB; ++k) {
                const double yk = readlane_f64(b, k) * readlane_f64(rd, k);
                b = (lane == k) ? yk : ((lane > k) ? fma(-Ls[lane][k], yk, b) : b);
            }
            ys[c0 + lane] = b;
        }
        __syncthreads();
    }
    for (int i = tid; i < w; i += 256) y[i] = ys[i];
}

// backward, diagonal nb-block: x_b = L_bb^-T y_b
__global__ __launch_bounds__(256) void trsv_diag_bwd_kernel(const double* __restrict__ L, int64_t lda,
                                                            double* __restrict__ y, int w) {
    __shared__ double ys[4096];
    __shared__ double Ls[IB][IB + 1];
    __shared__ double part[4][IB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < w; i += 256) ys[i] = y[i];
    for (int c0 = w - IB; c0 >= 0; c0 -= IB) {
#pragma unroll
        for (int u = 0; u < IB * IB / 256; ++u) {   // 16 independent loads in flight
            const int idx = tid + 256 * u;
            Ls[idx >> 6][idx & 63] = L[(int64_t)(c0 + (idx >> 6)) * lda + c0 + (idx & 63)];
        }
        __syncthreads();
        // x_j (j = c0+lane) needs sum_{i >= c0+64} L[i][j] x_i : column access, coalesced across lanes
        double s = 0.0;
        for (int i0 = c0 + IB + wave; i0 < w; i0 += 64) {   // (w - c0 - 64) is a multiple of 64: 16 loads per trip
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = L[(int64_t)(i0 + 4 * u) * lda + c0 + lane];
#pragma unroll
            for (int u = 0; u < 16; ++u) s += v[u] * ys[i0 + 4 * u];
        }
        part[wave][lane] = s;
        __syncthreads();
        if (wave == 0) {
            double b = ys[c0 + lane] - (part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane]);
            const double rd = 1.0 / Ls[lane][lane];
#pragma unroll 4
            for (int k = IB - 1; k >= 0; --k) {
                const double xk = readlane_f64(b, k) * readlane_f64(rd, k);
                b = (lane == k) ? xk : ((lane < k) ? fma(-Ls[k][lane], xk, b) : b);
            }
            ys[c0 + lane] = b;
        }
        __syncthreads();
    }
    for (int i = tid; i < w; i += 256) y[i] = ys[i];
}

// ---- pipelined forms of the two kernels above (w <= 1024): 320 threads = one SOLVER wave + four STREAMING waves.
// The chain of a diagonal block is 8 sub-blocks x (64-step substitution); in the kernels above every sub-block also
// waits for three global-memory round trips (diagonal tile, row segments, reduction) that do not depend on the
// solution - 12 us per sub-block, of which the substitution is under 3.  Here the streaming waves fetch, while the
// solver substitutes sub-block s, everything sub-block s+1 needs that is already computable: the partial sums over
// the part solved before s, the next diagonal tile and the 64 x 64 coupling tile (into registers).  After the barrier
// only a register-resident 64 x 64 gemv with the fresh y_s remains.  Same arithmetic, different summation order.
// Measured (N = 32768, rocprofv3): forward 94 -> 72 us per diagonal block, backward 79 -> 52 us; potrs 13.4 -> 10.3 ms.
// Round 3, measured and dropped: the substitution from REGISTERS (the streaming waves write the next tile as finished
// coefficients - scaled by the pivot's reciprocal, zero where the equation does not involve the pivot - and the solver
// loads its 64 coefficients between the barriers, so that the chain is v_readlane x 2 + v_fma per pivot).  The chain got
// shorter, the kernels did not: forward 44.8 us per diagonal block against 43.9, backward 39.8 against 32.9 (N = 32768,
// rocprofv3; with two or four trips of row-segment loads in flight alike) - a sub-block is bound by the streaming waves'
// dependent global-memory round trips and the two barriers, not by the solver's 64 pivots.
// A variant with lanes along the row and shuffle reductions for the forward partial sums (coalesced, but 16
// accumulators x 6 shuffles after the barrier) was measured and dropped: 174 us per diagonal block.
#define TRSV_PIPE_MAXW 1024
__global__ __launch_bounds__(320) void trsv_diag_fwd_pipe_kernel(const double* __restrict__ L, int64_t lda,
                                                                 double* __restrict__ y, int w, const double* __restrict__ Dg, int64_t drs, int dcm) {
    __shared__ double ys[TRSV_PIPE_MAXW];
    __shared__ double Tt[IB][IB + 1];      // current diagonal tile, TRANSPOSED: Tt[k][r] = T[r][k]; the pad keeps the
                                           // transposing writes (lane -> first index) off a single bank
    __shared__ double part[4][IB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool solver = wave == 0;
    const int q = wave - 1;                // streaming wave 0..3
    const int st = tid - 64;               // streaming thread 0..255
    for (int i = tid; i < w; i += 320) ys[i] = y[i];
    if (!solver) {
#pragma unroll
        for (int u = 0; u < IB * IB / 256; ++u) {
            const int idx = st + 256 * u;
            Tt[idx & 63][idx >> 6] = Dg[(int64_t)(idx >> 6) * drs + (idx & 63)];
        }
    }
    __syncthreads();
    double pend = 0.0;                     // solver: sum_{k < c0} L[c0 + lane][k] y[k]
    for (int c0 = 0; c0 < w; c0 += IB) {
        const int c1 = c0 + IB;
        const bool more = c1 < w;
        double tn[IB * IB / 256], xr[16], snext = 0.0;
        if (solver) {
            // b_r -= (T[r][k] / T[k][k]) * b_k : the scaled coefficient does not depend on the running b, so the chain
            // per step is readlane -> fma (the dependent fp64 latency is 32 cycles per op); y = b / diag at the end
            double b = ys[c0 + lane] - pend;
            const double rd = 1.0 / Tt[lane][lane];
#pragma unroll 8
            for (int k = 0; k < IB; ++k) {
                const double coef = Tt[k][lane] * readlane_f64(rd, k);
                const double bk = readlane_f64(b, k);
                b = (lane > k) ? fma(-coef, bk, b) : b;
            }
            ys[c0 + lane] = b * rd;
        } else if (more) {
#pragma unroll
            for (int u = 0; u < IB * IB / 256; ++u) {          // next diagonal tile
                const int idx = st + 256 * u;
                tn[u] = Dg[(int64_t)(c1 + (idx >> 6)) * drs + dcm * c1 + (idx & 63)];
            }
            const double* row = L + (int64_t)(c1 + lane) * lda;
#pragma unroll
            for (int u = 0; u < 16; u += 2) {                  // coupling tile: row c1 + lane, columns c0 + 16 q + u
                const double2 v = *reinterpret_cast<const double2*>(row + c0 + 16 * q + u);
                xr[u] = v.x;
                xr[u + 1] = v.y;
            }
            for (int m0 = 2 * q; m0 < c0; m0 += 64) {          // columns solved before this sub-block
                double2 v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = *reinterpret_cast<const double2*>(row + m0 + 8 * u);
#pragma unroll
                for (int u = 0; u < 8; ++u) snext += v[u].x * ys[m0 + 8 * u] + v[u].y * ys[m0 + 8 * u + 1];
            }
        }
        __syncthreads();                                       // y of sub-block c0 is in ys; Tt is free
        if (!solver && more) {
#pragma unroll
            for (int u = 0; u < 16; ++u) snext += xr[u] * ys[c0 + 16 * q + u];
            part[q][lane] = snext;
#pragma unroll
            for (int u = 0; u < IB * IB / 256; ++u) {
                const int idx = st + 256 * u;
                Tt[idx & 63][idx >> 6] = tn[u];
            }
        }
        __syncthreads();
        if (solver && more) pend = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
    }
    for (int i = tid; i < w; i += 320) y[i] = ys[i];
}

__global__ __launch_bounds__(320) void trsv_diag_bwd_pipe_kernel(const double* __restrict__ L, int64_t lda,
                                                                 double* __restrict__ y, int w) {
    __shared__ double ys[TRSV_PIPE_MAXW];
    __shared__ double Tn[IB][IB];          // current diagonal tile, row-major: the substitution reads row k by lane
    __shared__ double part[4][IB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool solver = wave == 0;
    const int q = wave - 1;
    const int st = tid - 64;
    for (int i = tid; i < w; i += 320) ys[i] = y[i];
    if (!solver) {
        const int c0 = w - IB;
#pragma unroll
        for (int u = 0; u < IB * IB / 256; ++u) {
            const int idx = st + 256 * u;
            Tn[idx >> 6][idx & 63] = L[(int64_t)(c0 + (idx >> 6)) * lda + c0 + (idx & 63)];
        }
    }
    __syncthreads();
    double pend = 0.0;                     // solver: sum_{i >= c0 + 64} L[i][c0 + lane] x[i]
    for (int c0 = w - IB; c0 >= 0; c0 -= IB) {
        const int c1 = c0 - IB;
        const bool more = c1 >= 0;
        double tn[IB * IB / 256], xr[16], snext = 0.0;
        if (solver) {
            double b = ys[c0 + lane] - pend;
            const double rd = 1.0 / Tn[lane][lane];
#pragma unroll 8
            for (int k = IB - 1; k >= 0; --k) {
                const double coef = Tn[k][lane] * readlane_f64(rd, k);       // off the chain (see the forward kernel)
                const double bk = readlane_f64(b, k);
                b = (lane < k) ? fma(-coef, bk, b) : b;
            }
            ys[c0 + lane] = b * rd;
        } else if (more) {
#pragma unroll
            for (int u = 0; u < IB * IB / 256; ++u) {          // next diagonal tile
                const int idx = st + 256 * u;
                tn[u] = L[(int64_t)(c1 + (idx >> 6)) * lda + c1 + (idx & 63)];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u)                       // coupling tile: rows c0 + 16 q + u, column c1 + lane
                xr[u] = L[(int64_t)(c0 + 16 * q + u) * lda + c1 + lane];
            for (int i0 = c0 + IB + q; i0 < w; i0 += 64) {     // rows solved before this sub-block
                double v[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) v[u] = L[(int64_t)(i0 + 4 * u) * lda + c1 + lane];
#pragma unroll
                for (int u = 0; u < 16; ++u) snext += v[u] * ys[i0 + 4 * u];
            }
        }
        __syncthreads();
        if (!solver && more) {
#pragma unroll
            for (int u = 0; u < 16; ++u) snext += xr[u] * ys[c0 + 16 * q + u];
            part[q][lane] = snext;
#pragma unroll
            for (int u = 0; u < IB * IB / 256; ++u) {
                const int idx = st + 256 * u;
                Tn[idx >> 6][idx & 63] = tn[u];
            }
        }
        __syncthreads();
        if (solver && more) pend = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
    }
    for (int i = tid; i < w; i += 320) y[i] = ys[i];
}

static void launch_trsv_diag(hipStream_t stream, const double* L, int64_t lda, double* v, int w, int trans, const trsv_diag_src* diag) {
    const trsv_diag_src in_place{L, lda, 1};
    const trsv_diag_src d = (diag != nullptr && !trans) ? *diag : in_place;
    if (w <= TRSV_PIPE_MAXW) {
        if (trans) hipLaunchKernelGGL(trsv_diag_bwd_pipe_kernel, dim3(1), dim3(320), 0, stream, L, lda, v, w);
        else hipLaunchKernelGGL(trsv_diag_fwd_pipe_kernel, dim3(1), dim3(320), 0, stream, L, lda, v, w, d.base, d.rs, d.cm);
    } else {
        if (trans) hipLaunchKernelGGL(trsv_diag_bwd_kernel, dim3(1), dim3(256), 0, stream, L, lda, v, w);
        else hipLaunchKernelGGL(trsv_diag_fwd_kernel, dim3(1), dim3(256), 0, stream, L, lda, v, w, d.base, d.rs, d.cm);
    }
}

// L2 warm-up for the NEXT diagonal block: the substitution kernel that follows is one workgroup, which the dispatcher
// places on XCD 0, and what it waits for per sub-block is one HBM round trip.  The workgroups of the wide gemv that
// land on XCD 0 (blockIdx % 8 == 0) touch one double per 128-byte line of that block's lower triangle, so that the
// lines sit in XCD 0's L2 when the substitution starts.  The loaded values feed a store that never executes.
__device__ __forceinline__ void trsv_warm_l2(const double* __restrict__ nxt, int64_t lda, int wn, double* sink) {
    if (nxt == nullptr || (blockIdx.x & 7) != 0) return;
    const int64_t g = blockIdx.x >> 3, ng = (gridDim.x + 7) >> 3;
    const int64_t lines_per_row = (wn + 15) / 16;
    double acc = 0.0;
    for (int64_t id = g * 256 + threadIdx.x; id < (int64_t)wn * lines_per_row; id += ng * 256) {
        const int64_t r = id / lines_per_row, c = (id % lines_per_row) * 16;
        if (c <= r) acc += nxt[r * lda + c];
    }
    if (acc == 1.2345678e-300) sink[0] = acc;
}

// forward update: f[i] -= sum_{k<w} Lp[i][k] y[k]  for the rows below the block; one wave per row
__global__ __launch_bounds__(256) void trsv_update_fwd_kernel(const double* __restrict__ Lp, int64_t lda,
                                                              const double* __restrict__ y, int w,
                                                              double* __restrict__ f, int64_t rows,
                                                              const double* __restrict__ warm, int warm_w) {
    __shared__ double ys[4096];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    trsv_warm_l2(warm, lda, warm_w, f);
    for (int i = tid; i < w; i += 256) ys[i] = y[i];
    __syncthreads();
    for (int64_t r = (int64_t)blockIdx.x * 4 + wave; r < rows; r += (int64_t)gridDim.x * 4) {
        const double* row = Lp + r * lda;
        double s = 0.0;
        for (int k = 2 * lane; k < w; k += 128) {
            const double2 v = *reinterpret_cast<const double2*>(row + k);
            s += v.x * ys[k] + v.y * ys[k + 1];
        }
        for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
        if (lane == 0) f[r] -= s;
    }
}

// backward update: y[j] -= sum_{i<w} Lb[i][j] x[i]  for the columns left of the block; 64 columns per workgroup
__global__ __launch_bounds__(256) void trsv_update_bwd_kernel(const double* __restrict__ Lb, int64_t lda,
                                                              const double* __restrict__ x, int w,
                                                              double* __restrict__ y, int64_t cols,
                                                              const double* __restrict__ warm, int warm_w) {
    __shared__ double xs[4096];
    __shared__ double part[4][IB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    trsv_warm_l2(warm, lda, warm_w, y);
    for (int i = tid; i < w; i += 256) xs[i] = x[i];
    __syncthreads();
    const int64_t j = (int64_t)blockIdx.x * IB + lane;
    double s = 0.0;
    if (j < cols)
        for (int i0 = wave; i0 < w; i0 += 64) {   // w is a multiple of 64: 16 independent loads per trip
            double v[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) v[u] = Lb[(int64_t)(i0 + 4 * u) * lda + j];
#pragma unroll
            for (int u = 0; u < 16; ++u) s += v[u] * xs[i0 + 4 * u];
        }
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && j < cols) y[j] -= part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
}

// ---- [r3] the backward sweep with the wide gemv OFF the chain: one launch per diagonal block -------------------------------
// Block b can be solved as soon as x_{b+1} has been applied to ITS OWN w columns (a w x wn piece of the gemv, 2 MB at nb = 512);
// the rest of that gemv (the columns left of block b) is only needed by later launches.  One launch therefore holds
//   workgroup 0            the substitution on block b (the pipelined solver above), which first waits - bounded - until
//   workgroups 1 .. w/64   have applied x_{b+1} to the columns of block b (64 columns each, counted on a device word), while
//   the other workgroups   apply x_{b+1} to everything left of block b, beside the substitution.
// Everything in the launch depends on x_{b+1} alone (the previous launch); nobody but workgroup 0 waits, and it waits only for
// workgroups that wait for nothing, so the launch drains whatever order the dispatcher takes.  The chain per block is
// launch + piece (~3 us) + substitution instead of substitution + launch + a gemv over up to 130 MB (18 us at N = 32768).
// If the count does not arrive within the bound the solution is poisoned with NaN (the callers' residual checks fail loudly).
__global__ __launch_bounds__(320) void trsv_bwd_fused_kernel(const double* __restrict__ L, int64_t lda, double* __restrict__ v,
                                                             int64_t r0, int w, int wn, unsigned* __restrict__ counter,
                                                             unsigned target, const double* __restrict__ warm, int warm_w) {
    __shared__ double ys[TRSV_PIPE_MAXW];  // solver: the block's right-hand side / solution; update: x_{b+1}
    __shared__ double Tn[IB][IB];
    __shared__ double part[4][IB];
    __shared__ int gave_up;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (blockIdx.x != 0) {
        // ---- gemv workgroups: 64 columns each, the piece (columns of block b) first in block order
        const int np = w / IB;
        const int64_t c = (int64_t)blockIdx.x - 1;
        const bool piece = c < np;
        const int64_t j = (piece ? r0 + c * IB : (c - np) * IB) + lane;          // rest: columns [0, r0)
        if (!piece) trsv_warm_l2(warm, lda, warm_w, v);
        const double* __restrict__ x = v + r0 + w;
        for (int i = tid; i < wn; i += 320) ys[i] = x[i];
        __syncthreads();
        double s = 0.0;
        if (wave < 4) {
            const double* __restrict__ Lb = L + (r0 + w) * lda + j;
            for (int i0 = wave; i0 < wn; i0 += 64) {                           // wn is a multiple of 64: 16 independent loads per trip
                double t[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) t[u] = Lb[(int64_t)(i0 + 4 * u) * lda];
#pragma unroll
                for (int u = 0; u < 16; ++u) s += t[u] * ys[i0 + 4 * u];
            }
            part[wave][lane] = s;
        }
        __syncthreads();
        if (wave == 0) {
            v[j] -= part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
            if (piece) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");             // the wave's stores before the count
                if (lane == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        return;
    }
    // ---- workgroup 0: the substitution (trsv_diag_bwd_pipe_kernel with the wait in front of the first read of y)
    const double* __restrict__ Ld = L + r0 * lda + r0;
    double* __restrict__ y = v + r0;
    const bool solver = wave == 0;
    const int q = wave - 1;
    const int st = tid - 64;
    if (!solver) {
        const int c0 = w - IB;
#pragma unroll
        for (int u = 0; u < IB * IB / 256; ++u) {
            const int idx = st + 256 * u;
            Tn[idx >> 6][idx & 63] = Ld[(int64_t)(c0 + (idx >> 6)) * lda + c0 + (idx & 63)];
        }
    }
    if (tid == 0) {
        int failed = 0;
        if (wn > 0) {
            failed = 1;
            for (int guard = 0; guard < (1 << 22); ++guard) {
                if ((int)(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0) { failed = 0; break; }
                __builtin_amdgcn_s_sleep(2);
            }
        }
        gave_up = failed;
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                         // the pieces' stores, not this CU's stale lines
    const bool bad = gave_up != 0;
    for (int i = tid; i < w; i += 320) ys[i] = bad ? __builtin_nan("") : y[i];
    __syncthreads();
    double pend = 0.0;                     // solver: sum_{i >= c0 + 64} L[i][c0 + lane] x[i]
    for (int c0 = w - IB; c0 >= 0; c0 -= IB) {
        const int c1 = c0 - IB;
        const bool more = c1 >= 0;
        double tn[IB * IB / 256], xr[16], snext = 0.0;
        if (solver) {
            double b = ys[c0 + lane] - pend;
            const double rd = 1.0 / Tn[lane][lane];
#pragma unroll 8
            for (int k = IB - 1; k >= 0; --k) {
                const double coef = Tn[k][lane] * readlane_f64(rd, k);
                const double bk = readlane_f64(b, k);
                b = (lane < k) ? fma(-coef, bk, b) : b;
            }
            ys[c0 + lane] = b * rd;
        } else if (more) {
#pragma unroll
            for (int u = 0; u < IB * IB / 256; ++u) {          // next diagonal tile
                const int idx = st + 256 * u;
                tn[u] = Ld[(int64_t)(c1 + (idx >> 6)) * lda + c1 + (idx & 63)];
            }
#pragma unroll
            for (int u = 0; u < 16; ++u)                       // coupling tile: rows c0 + 16 q + u, column c1 + lane
                xr[u] = Ld[(int64_t)(c0 + 16 * q + u) * lda + c1 + lane];
            for (int i0 = c0 + IB + q; i0 < w; i0 += 64) {     // rows solved before this sub-block
                double t[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) t[u] = Ld[(int64_t)(i0 + 4 * u) * lda + c1 + lane];
#pragma unroll
                for (int u = 0; u < 16; ++u) snext += t[u] * ys[i0 + 4 * u];
            }
        }
        __syncthreads();
        if (!solver && more) {
#pragma unroll
            for (int u = 0; u < 16; ++u) snext += xr[u] * ys[c0 + 16 * q + u];
            part[q][lane] = snext;
#pragma unroll
            for (int u = 0; u < IB * IB / 256; ++u) {
                const int idx = st + 256 * u;
                Tn[idx >> 6][idx & 63] = tn[u];
            }
        }
        __syncthreads();
        if (solver && more) pend = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
    }
    for (int i = tid; i < w; i += 320) y[i] = ys[i];
}

// ---- [r3] the backward sweep with INVERTED diagonal blocks: no substitution chain at all ---------------------------------------
// With D_b = L_bb^-1 (ensure_dinv: all blocks in 7 batched launches, cached on the factor and shared with the variance path)
// the solve of block b is a 512 x 512 transposed matrix-vector product, x_b = D_b^T r_b - eight workgroups instead of one
// workgroup's 8 x 64 dependent steps.  One launch per block, three roles by workgroup index:
//   solve  (w/64)   wait - bounded - for the pieces' count, then x_b[j] = sum_{i >= j} D_b[i][j] r_b[i] for 64 columns each
//   piece  (w/64)   r_b = y_b - L[b+1 rows][b cols]^T x_{b+1} for 64 columns each -> scratch R, counted on a device word
//   rest            y[j] -= L[b+1 rows][j]^T x_{b+1} for the columns left of block b (HBM-bound, beside the other two)
// Only the solve workgroups wait, and only for workgroups that wait for nothing.  The next launch starts when this one has
// drained, so a block costs max(rest, piece + solve) + one launch: the sweep is the 4 N^2 bytes of L through HBM once.
// If the count does not arrive within the bound the solution is poisoned with NaN (the callers' residual checks fail loudly).
__global__ __launch_bounds__(256) void trsv_bwd_dinv_kernel(const double* __restrict__ L, int64_t lda, double* __restrict__ v,
                                                            int64_t r0, int w, int wn, const double* __restrict__ Dn, int64_t ldd,
                                                            double* __restrict__ R, unsigned* __restrict__ counter, unsigned target,
                                                            const double* __restrict__ warm, int warm_w) {
    __shared__ double xs[TRSV_PIPE_MAXW];
    __shared__ double part[4][IB];
    __shared__ int gave_up;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int ns = w / IB;
    if ((int)blockIdx.x < ns) {
        // ---- solve: columns [64 c, 64 c + 64) of the block
        const int c = blockIdx.x;
        if (tid == 0) {
            int failed = 1;
            for (int guard = 0; guard < (1 << 22); ++guard) {
                if ((int)(__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) - target) >= 0) { failed = 0; break; }
                __builtin_amdgcn_s_sleep(1);
            }
            gave_up = failed;
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                     // the pieces' stores, not this CU's stale lines
        const bool bad = gave_up != 0;
        for (int i = 64 * c + tid; i < w; i += 256) xs[i] = R[i];
        __syncthreads();
        const int j = 64 * c + lane;
        const double* __restrict__ Db = Dn + j;
        double s = 0.0;
        for (int i0 = 64 * c + wave; i0 < w; i0 += 64) {                       // D is lower triangular: rows from the column's own tile down
            double t[16];
#pragma unroll
            for (int u = 0; u < 16; ++u) t[u] = Db[(int64_t)(i0 + 4 * u) * ldd];
#pragma unroll
            for (int u = 0; u < 16; ++u) s += t[u] * xs[i0 + 4 * u];
        }
        part[wave][lane] = s;
        __syncthreads();
        if (wave == 0) v[r0 + j] = bad ? __builtin_nan("") : -(part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane]);   // Dn = -D
        return;
    }
    // ---- piece / rest: 64 columns each, apply x_{b+1} (nothing for the last block of the matrix: wn == 0)
    const int64_t g = (int64_t)blockIdx.x - ns;
    const bool piece = g < ns;
    const int64_t j = (piece ? r0 + g * IB : (g - ns) * IB) + lane;
    if (!piece) trsv_warm_l2(warm, lda, warm_w, v);
    const double* __restrict__ x = v + r0 + w;
    for (int i = tid; i < wn; i += 256) xs[i] = x[i];
    __syncthreads();
    double s = 0.0;
    const double* __restrict__ Lb = L + (r0 + w) * lda + j;
    for (int i0 = wave; i0 < wn; i0 += 64) {                                   // wn is a multiple of 64: 16 independent loads per trip
        double t[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) t[u] = Lb[(int64_t)(i0 + 4 * u) * lda];
#pragma unroll
        for (int u = 0; u < 16; ++u) s += t[u] * xs[i0 + 4 * u];
    }
    part[wave][lane] = s;
    __syncthreads();
    if (wave == 0) {
        const double r = v[j] - (part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane]);
        if (piece) {
            R[j - r0] = r;
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");                 // the wave's stores before the count
            if (lane == 0) __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        } else {
            v[j] = r;
        }
    }
}

static void launch_trsv_update(hipStream_t stream, const double* L, int64_t lda, double* v, int64_t r0, int w, int64_t first,
                               int64_t count, int trans, const double* warm, int warm_w) {
    if (count <= 0) return;
    if (!trans) {    // rows [first, first + count) below the block:  v[i] -= L[i][r0:r0+w] y
        int64_t blocks = (count + 3) / 4;
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(trsv_update_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, L + first * lda + r0, lda, v + r0,
                           w, v + first, count, warm, warm_w);
    } else {         // columns [first, first + count) left of the block:  v[j] -= L[r0:r0+w][j]^T x
        hipLaunchKernelGGL(trsv_update_bwd_kernel, dim3((unsigned)((count + IB - 1) / IB)), dim3(256), 0, stream,
                           L + r0 * lda + first, lda, v + r0, w, v + first, count, warm, warm_w);
    }
}

// L y = f, then L^T x = y, one right-hand side.  FORWARD, per outer block: a one-workgroup substitution kernel on the diagonal
// block (the chain) and a wide gemv on everything that block feeds (HBM-bound), all on one stream.  Splitting the gemv
// so that only the next block's entries stay on the chain and the rest runs on the second stream was measured and
// dropped in round 2: the event record/wait pairs it puts on the chain cost more than the ~15 us gemv they take off it
// (11.4 vs 10.3 ms at N = 32768, 2.55 vs 2.15 ms at N = 8192).  BACKWARD [r3]: one launch per block that holds the block's
// solve AND the gemv of the previous block's solution (trsv_bwd_dinv_kernel / trsv_bwd_fused_kernel above) - what the
// two-stream split wanted, without events.
// which: 1 = forward only (L y = f: what the quadratic form f^T K^-1 f = y^T y of the log-likelihood needs),
//        2 = backward only (L^T x = y), 3 = both
static int trsv_sweeps(mgpi_ctx* ctx, const mgpi_mat* Lm, double* v, int which) {
    const int64_t pn = Lm->prows <= Lm->pcols ? Lm->prows : mgpi_roundup(Lm->rows, IB);
    const int64_t lda = Lm->ld;
    const int64_t nb = ctx->nb;
    const double* L = Lm->d;
    const int64_t nblk = (pn + nb - 1) / nb;
    for (int64_t r0 = 0; (which & 1) && r0 < pn; r0 += nb) {
        const int w = (int)((pn - r0 < nb) ? (pn - r0) : nb);
        launch_trsv_diag(ctx->stream, L + r0 * lda + r0, lda, v + r0, w, 0);
        const int64_t r1 = r0 + w;
        const int w1 = (int)((pn - r1 < nb) ? (pn - r1) : nb);
        launch_trsv_update(ctx->stream, L, lda, v, r0, w, r1, pn - r1, 0, w1 > 0 ? L + r1 * lda + r1 : nullptr, w1);
    }
    // [r3] backward: one fused launch per block (trsv_bwd_fused_kernel / trsv_bwd_dinv_kernel) for nb up to 1024
    const bool fused = nb <= TRSV_PIPE_MAXW && nb % IB == 0;
    // MGPI_TRSV_DINV=1 (default): inverted diagonal blocks (trsv_bwd_dinv_kernel); 0: the substitution kernels
    // Only for factors whose diagonal spans at most MGPI_TRSV_DINV_RATIO (default 1e4: cond K up to ~4e10 for the SE + nugget
    // family, profiles/r03_dinv_accuracy.md; the parity contract ends at 1e7): a product with an explicit inverse has a backward error of ~cond(L_bb) eps
    // instead of eps, and at cond K ~ 5e15 (tests/test_gpu_configs.py::test_pivot_tolerance_is_configurable) that showed:
    // 6e-13 against < 1e-13.  A factor whose status was never read (range unknown) takes the substitution.
    static const int dinv_env = getenv("MGPI_TRSV_DINV") ? atoi(getenv("MGPI_TRSV_DINV")) : 1;
    static const double dinv_ratio = getenv("MGPI_TRSV_DINV_RATIO") ? atof(getenv("MGPI_TRSV_DINV_RATIO")) : 1e4;
    const bool by_inverse = fused && dinv_env != 0 && (which & 2) && Lm->diag_min > 0.0 && Lm->diag_max <= dinv_ratio * Lm->diag_min;
    if (by_inverse) MGPI_TRY(ensure_dinv(ctx, Lm));
    unsigned* piece_count = reinterpret_cast<unsigned*>(ctx->d_scal + 8080);
    double* piece_r = ctx->d_scal + 5000;                        // r_b of the block being solved (<= 1024 doubles)
    unsigned pieces_done = 0;
    if (fused && (which & 2)) MGPI_HIP_TRY(hipMemsetAsync(piece_count, 0, sizeof(unsigned), ctx->stream));
    for (int64_t b = (which & 2) ? nblk - 1 : -1; b >= 0; --b) {
        const int64_t r0 = b * nb;
        const int w = (int)((pn - r0 < nb) ? (pn - r0) : nb);
        const int wp = (int)((r0 < nb) ? r0 : nb);
        const double* warm = wp > 0 ? L + (r0 - wp) * lda + (r0 - wp) : nullptr;
        if (by_inverse) {
            const int wn = (b == nblk - 1) ? 0 : (int)((pn - (r0 + w) < nb) ? (pn - (r0 + w)) : nb);   // the block solved just before
            const int64_t groups = (wn > 0 ? r0 : 0) / IB + 2 * (w / IB);                               // rest + piece + solve
            pieces_done += (unsigned)(w / IB);
            hipLaunchKernelGGL(trsv_bwd_dinv_kernel, dim3((unsigned)groups), dim3(256), 0, ctx->stream, L, lda, v, r0, w, wn,
                               mgpi_dinv_block(Lm, b), nb, piece_r, piece_count, pieces_done, warm, wp);
            continue;
        }
        if (fused) {
            const int wn = (b == nblk - 1) ? 0 : (int)((pn - (r0 + w) < nb) ? (pn - (r0 + w)) : nb);   // the block solved just before
            const int64_t groups = wn > 0 ? (r0 + w) / IB : 0;
            if (wn > 0) pieces_done += (unsigned)(w / IB);
            hipLaunchKernelGGL(trsv_bwd_fused_kernel, dim3((unsigned)(1 + groups)), dim3(320), 0, ctx->stream, L, lda, v, r0, w, wn,
                               piece_count, pieces_done, warm, wp);
            continue;
        }
        launch_trsv_diag(ctx->stream, L + r0 * lda + r0, lda, v + r0, w, 1);
        launch_trsv_update(ctx->stream, L, lda, v, r0, w, 0, r0, 1, warm, wp);
    }
    MGPI_HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int mgpi_potrs_dev(mgpi_ctx* ctx, const mgpi_mat* L, mgpi_mat* B) {
    MGPI_REQUIRE(ctx && L && B, "mgpi_potrs_dev: null pointer");
    if (!L->factored) {
        mgpi_set_error("mgpi_potrs_dev: L has not been factored by mgpi_potrf_dev");
        return MGPI_ERR_STATE;
    }
    MGPI_REQUIRE(B->rows == L->rows, "mgpi_potrs_dev: B has %lld rows, L is %lld", (long long)B->rows, (long long)L->rows);
    MGPI_REQUIRE(B->prows >= L->prows, "mgpi_potrs_dev: B row padding smaller than L's");
    MGPI_HIP_TRY(hipSetDevice(ctx->device));
    if (B->cols == 1 && B->ld == 1) return trsv_sweeps(ctx, L, B->d, 3);
    MGPI_REQUIRE(B->pcols % 64 == 0, "mgpi_potrs_dev: a multi-column B must be tile padded (cols > %d)", MGPI_MAX_D);
    MGPI_TRY(trsm_blocked(ctx, L, B->d, B->ld, B->pcols, 0));
    return trsm_blocked(ctx, L, B->d, B->ld, B->pcols, 1);
}

extern "C" int mgpi_potrs(mgpi_ctx* ctx, const mgpi_mat* L, const double* b, int64_t n, int nrhs, double* x) {
    MGPI_REQUIRE(ctx && L && b && x, "mgpi_potrs: null pointer");
    MGPI_REQUIRE(n == L->rows && nrhs >= 1, "mgpi_potrs: n=%lld does not match L (%lld) or nrhs < 1", (long long)n, (long long)L->rows);
    mgpi_mat* B = nullptr;
    // one rhs: a true vector (ld == 1); several: a wide padded matrix
    int st = (nrhs == 1) ? mgpi_mat_alloc(ctx, n, 1, &B) : mgpi_mat_alloc_wide(ctx, n, nrhs, &B);
    if (st == 0) st = mgpi_mat_upload(B, b, nrhs);
    if (st == 0) st = mgpi_potrs_dev(ctx, L, B);
    if (st == 0) st = mgpi_mat_download(B, x, nrhs);
    mgpi_mat_free(B);
    return st;
}

// ---- block-level entry points for the multi-GPU (2-D block-cyclic) driver ---------------------------------
// A <- A L^-T for a factored diagonal block L (w x w) and a block of rows A (m x w): the panel trsm of the
// right-looking factorisation, applied by the ranks that own the blocks below a remote diagonal block.
// One 64-column step of  A <- A L^-T  in accumulator layout (panel_device.h), the multi-GPU counterpart of
// panel_step_kernel: the factor is GIVEN (the diagonal block arrived by broadcast), so there is nothing to factor - a
// column-role workgroup per 64 rows applies the previous step's columns to its rows of the current column block
// (deferred right-looking update), builds the LDS images of the 64 x 64 diagonal sub-block and substitutes; update-role
// workgroups apply the previous columns to the remaining column blocks.  One launch per step instead of two.
__global__ __launch_bounds__(256, 2) void trsm_right_step_kernel(double* __restrict__ A, int64_t lda, int nrowblk,
                                                                 const double* __restrict__ Lk, int64_t ldl, int64_t c0, int has_prev,
                                                                 int rest_cols, const double* __restrict__ strip) {
    __shared__ panel_lds s;
    MGPI_CHAIN_PRIO();
    const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, c = lane & 15;
    // the wave index in a scalar register: branches on it become scalar branches instead of EXEC-masked regions whose merges
    // copy every accumulator (60 v_mov_b64 per round - vector instructions the starved issue stage pays for one by one)
    const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int64_t cp = c0 - IB, c1 = c0 + IB;
    panel_double4_t acc[4];
    if ((int)blockIdx.x >= nrowblk) {
        const int t = (int)blockIdx.x - nrowblk, ti = t % nrowblk, tj = t / nrowblk;
        double* C = A + ((int64_t)IB * ti) * lda + c1 + (int64_t)IB * tj;
        panel_load(acc, C, lda, w, g, c, false);
        panel_gemm_nt(acc, A + ((int64_t)IB * ti + 16 * w + c) * lda + cp + 2 * g, Lk + (c1 + (int64_t)IB * tj + c) * ldl + cp + 2 * g, ldl, 0, 4);
        panel_store(acc, C, lda, w, g, c);
        return;
    }
    const int64_t r0 = (int64_t)IB * blockIdx.x;
    panel_load(acc, A + r0 * lda + c0, lda, w, g, c, false);
    if (has_prev) panel_gemm_nt(acc, A + (r0 + 16 * w + c) * lda + cp + 2 * g, Lk + (c0 + c) * ldl + cp + 2 * g, ldl, 0, 4);
    // images of the factored 64 x 64 diagonal sub-block: Lt[k][m] = L[m][k] (m >= k), 0 above; 1 / diagonal
    // (during a factorisation the 64 x 64 diagonal tiles still wait in the scratch strip, row r of the block at strip + 64 r)
    const double* Dt = strip ? strip + c0 * IB : Lk + c0 * ldl + c0;
    const int64_t dts = strip ? IB : ldl;
    for (int i = tid; i < IB * IB; i += 256) {
        const int m = i >> 6, k = i & 63;                          // coalesced read of L[m][k]
        s.Lt[k][m] = (m >= k) ? Dt[m * dts + k] : 0.0;
    }
    if (tid < IB) s.invd[tid] = 1.0 / Dt[tid * dts + tid];
    __syncthreads();
    panel_trsm_images(s, tid);
    trsm_acc_mx<0>::run(acc, s, w, g, c, A + (r0 + 16 * w + c) * lda + c0);
}

static int launch_trsm_right_steps(hipStream_t stream, double* A, int64_t lda, int64_t m, const double* L, int64_t ldl, int64_t w,
                                   const double* strip) {
    MGPI_REQUIRE(w % IB == 0 && m % IB == 0 && m / IB < 32768, "trsm_right: %lld rows x %lld columns (multiples of %d, at most 2 M rows)",
                 (long long)m, (long long)w, IB);
    const int nrowblk = (int)(m / IB);
    if (nrowblk == 0) return 0;
    for (int64_t c0 = 0; c0 < w; c0 += IB) {
        const int has_prev = c0 > 0;
        const int rest_cols = has_prev ? (int)((w - c0 - IB) / IB) : 0;
        hipLaunchKernelGGL(trsm_right_step_kernel, dim3((unsigned)(nrowblk * (1 + rest_cols))), dim3(256), 0, stream, A, lda, nrowblk, L, ldl,
                           c0, has_prev, rest_cols, strip);
    }
    MGPI_HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int mgpi_trsm_right_dev(mgpi_ctx* ctx, const mgpi_mat* L, mgpi_mat* A) {
    MGPI_REQUIRE(ctx && L && A, "mgpi_trsm_right_dev: null pointer");
    MGPI_REQUIRE(L->rows == L->cols && A->cols == L->rows, "mgpi_trsm_right_dev: A has %lld columns, L is %lld x %lld",
                 (long long)A->cols, (long long)L->rows, (long long)L->cols);
    MGPI_REQUIRE(L->rows % IB == 0 && A->prows % IB == 0, "mgpi_trsm_right_dev: dimensions must be multiples of %d", IB);
    MGPI_HIP_TRY(hipSetDevice(ctx->device));
    return launch_trsm_right_steps(ctx->stream, A->d, A->ld, A->prows, L->d, L->ld, L->rows, nullptr);
}

// one triangular half-solve with a factored diagonal block on an (n x 1) vector: L y = b (trans = 0), L^T x = b (1)
extern "C" int mgpi_trsv_dev(mgpi_ctx* ctx, const mgpi_mat* L, mgpi_mat* v, int trans) {
    MGPI_REQUIRE(ctx && L && v, "mgpi_trsv_dev: null pointer");
    MGPI_REQUIRE(L->rows == L->cols && v->ld == 1 && v->prows >= L->prows, "mgpi_trsv_dev: shapes do not conform");
    MGPI_REQUIRE(L->prows % IB == 0 && L->prows <= 4096, "mgpi_trsv_dev: block of %lld rows (need a multiple of %d, <= 4096)",
                 (long long)L->prows, IB);
    MGPI_HIP_TRY(hipSetDevice(ctx->device));
    launch_trsv_diag(ctx->stream, L->d, L->ld, v->d, (int)L->prows, trans);
    MGPI_HIP_TRY(hipGetLastError());
    return 0;
}

// y -= A x (trans = 0, A is m x k, x (k), y (m))   or   y -= A^T x (trans = 1, x (m), y (k)); vectors have ld == 1
extern "C" int mgpi_gemv_dev(mgpi_ctx* ctx, const mgpi_mat* A, const mgpi_mat* x, mgpi_mat* y, int trans) {
    MGPI_REQUIRE(ctx && A && x && y, "mgpi_gemv_dev: null pointer");
    MGPI_REQUIRE(x->ld == 1 && y->ld == 1, "mgpi_gemv_dev: x and y must be vectors");
    const int64_t m = A->prows, k = A->pcols;
    MGPI_REQUIRE((trans ? x->prows >= m && y->prows >= k : x->prows >= k && y->prows >= m), "mgpi_gemv_dev: vector too short");
    MGPI_REQUIRE(k % 128 == 0 && m % 64 == 0, "mgpi_gemv_dev: A must be tile padded");
    MGPI_HIP_TRY(hipSetDevice(ctx->device));
    if (!trans) {
        for (int64_t c0 = 0; c0 < k; c0 += 4096) {   // x chunk lives in LDS
            const int w = (int)((k - c0 < 4096) ? (k - c0) : 4096);
            int64_t blocks = (m + 3) / 4;
            if (blocks > 2048) blocks = 2048;
            hipLaunchKernelGGL(trsv_update_fwd_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, A->d + c0, A->ld,
                               x->d + c0, w, y->d, m, (const double*)nullptr, 0);
        }
    } else {
        for (int64_t r0 = 0; r0 < m; r0 += 4096) {
            const int w = (int)((m - r0 < 4096) ? (m - r0) : 4096);
            hipLaunchKernelGGL(trsv_update_bwd_kernel, dim3((unsigned)((k + IB - 1) / IB)), dim3(256), 0, ctx->stream,
                               A->d + r0 * A->ld, A->ld, x->d + r0, w, y->d, k, (const double*)nullptr, 0);
        }
    }
    MGPI_HIP_TRY(hipGetLastError());
    return 0;
}
