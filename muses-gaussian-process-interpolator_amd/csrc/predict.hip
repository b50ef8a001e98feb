// Posterior contractions: the `@` chains of mgpi/interpolators.py:100,125,189,199.
//   predict mean  : mean = K(x*,x) alpha, K* never materialised (fused kernel evaluation + dot product)
//   predict var   : var_i = k(x*_i,x*_i) - || L^-1 k(x, x*_i) ||^2   (diag of :199), V = L^-1 K*^T by blocked trsm
//   condition cov : cov = K** - V^T V                                  (:148,:157,:199)
#include "cov_device.h"
#include <stdlib.h>

int mgpi_trsm_raw(mgpi_ctx* ctx, const mgpi_mat* L, double* B, int64_t ldb, int64_t ncols, int trans, int use_inv, double* Xout,
                  int* in_xout);

#define PM_CHUNK 512   // sources staged per workgroup pass

// partial[chunk][i] = sum_{j in chunk} k(xs_i, x_j) alpha_j ; one thread per query, sources broadcast from LDS
template <int D, int MIX>
__global__ __launch_bounds__(256) void predict_mean_kernel(mgpi_terms_dev t, const double* __restrict__ xs, int64_t ldxs,
                                                           int64_t m, const double* __restrict__ x, int64_t ldx,
                                                           int64_t n, const double* __restrict__ alpha,
                                                           int64_t chunk_len, double* __restrict__ partial) {
    constexpr int DD = (D > 0) ? D : MGPI_MAX_D;
    constexpr int CH = (D > 0) ? PM_CHUNK : PM_CHUNK / 2;   // keep the generic-D image under 64 KB of LDS
    __shared__ double sx[CH][DD];
    __shared__ double sa[CH];
    const int nd = (D > 0) ? D : t.d;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double q[DD];
#pragma unroll
    for (int k = 0; k < DD; ++k) q[k] = (k < nd && i < m) ? xs[i * ldxs + k] : 0.0;
    const int64_t j0 = (int64_t)blockIdx.y * chunk_len;
    int64_t j1 = j0 + chunk_len;
    if (j1 > n) j1 = n;
    double acc = 0.0;
    for (int64_t base = j0; base < j1; base += CH) {
        const int cnt = (int)((j1 - base < CH) ? (j1 - base) : CH);
        __syncthreads();
        for (int idx = threadIdx.x; idx < cnt * nd; idx += 256) {
            const int r = idx / nd, k = idx - r * nd;
            sx[r][k] = x[(base + r) * ldx + k];
        }
        for (int idx = threadIdx.x; idx < cnt; idx += 256) sa[idx] = alpha[base + idx];
        __syncthreads();
        for (int r = 0; r < cnt; ++r) acc += mgpi_cov_entry<D, MIX>(t, q, sx[r]) * sa[r];
    }
    if (i < m) partial[(int64_t)blockIdx.y * m + i] = acc;
}

// MEASURED VARIANT (MGPI_PREDICT_MFMA=1): the contraction K* alpha on the matrix cores, as north_star words it ("the
// K(x*,x) alpha predict step as a tiled fp64 MFMA GEMV").  One wave = 16 queries; per micro-step lane (c, g) evaluates
// k(query c, source 4 s + g) - the A operand of v_mfma_f64_16x16x4_f64 - and supplies alpha[4 s + g] in column 0 of the
// B operand (zeros elsewhere: a GEMV uses one of the instruction's 16 output columns).  The kernel evaluation stays on
// the vector ALU either way (~38 instructions per pair); what moves to the matrix pipe is ONE fma per pair.  Timing and
// the verdict are in DESIGN.md; the default stays the vector kernel above.
typedef double pm_double4_t __attribute__((ext_vector_type(4)));
template <int D, int MIX>
__global__ __launch_bounds__(256) void predict_mean_mfma_kernel(mgpi_terms_dev t, const double* __restrict__ xs, int64_t ldxs,
                                                                int64_t m, const double* __restrict__ x, int64_t ldx,
                                                                int64_t n, const double* __restrict__ alpha,
                                                                int64_t chunk_len, double* __restrict__ partial) {
    constexpr int CH = PM_CHUNK;
    __shared__ double sx[CH][D];
    __shared__ double sa[CH];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, c = lane & 15, g = lane >> 4;
    const int64_t i = (int64_t)blockIdx.x * 64 + 16 * wave + c;         // this lane's query
    double q[D];
#pragma unroll
    for (int k = 0; k < D; ++k) q[k] = (i < m) ? xs[i * ldxs + k] : 0.0;
    const int64_t j0 = (int64_t)blockIdx.y * chunk_len;
    int64_t j1 = j0 + chunk_len;
    if (j1 > n) j1 = n;
    pm_double4_t acc = {0.0, 0.0, 0.0, 0.0};
    for (int64_t base = j0; base < j1; base += CH) {
        const int cnt = (int)((j1 - base < CH) ? (j1 - base) : CH);
        __syncthreads();
        for (int idx = threadIdx.x; idx < CH * D; idx += 256) {
            const int r = idx / D, k = idx - r * D;
            sx[r][k] = (r < cnt) ? x[(base + r) * ldx + k] : 0.0;
        }
        for (int idx = threadIdx.x; idx < CH; idx += 256) sa[idx] = (idx < cnt) ? alpha[base + idx] : 0.0;   // padded sources weigh 0
        __syncthreads();
        for (int r = 0; r < CH; r += 4) {
            const double kv = mgpi_cov_entry<D, MIX>(t, q, sx[r + g]);
            const double bv = (c == 0) ? sa[r + g] : 0.0;
            acc = __builtin_amdgcn_mfma_f64_16x16x4f64(kv, bv, acc, 0, 0, 0);
        }
    }
    // column 0 of the 16 x 16 result: lanes with c == 0 hold rows g + 4 r
    if (c == 0) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int64_t qi = (int64_t)blockIdx.x * 64 + 16 * wave + g + 4 * r;
            if (qi < m) partial[(int64_t)blockIdx.y * m + qi] = acc[r];
        }
    }
}

// out[i] = sum_c partial[c][i] in chunk order (deterministic)
__global__ void reduce_chunks_kernel(const double* __restrict__ partial, int64_t m, int nchunks, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    double s = 0.0;
    for (int c = 0; c < nchunks; ++c) s += partial[(int64_t)c * m + i];
    out[i] = s;
}

extern "C" int mgpi_predict_mean_dev(mgpi_ctx* ctx, const mgpi_mat* Xs, const mgpi_mat* X, const mgpi_kernel_desc* terms,
                                     int nterms, const mgpi_mat* alpha, mgpi_mat* mean) {
    MGPI_REQUIRE(ctx && Xs && X && alpha && mean, "mgpi_predict_mean_dev: null pointer");
    MGPI_REQUIRE(Xs->cols == X->cols, "conflicting number of dimensions for x1 and x2 (%lld vs %lld)", (long long)Xs->cols,
                 (long long)X->cols);
    MGPI_REQUIRE(alpha->rows == X->rows && alpha->ld == 1, "mgpi_predict_mean_dev: alpha must be an (n x 1) vector");
    MGPI_REQUIRE(mean->rows == Xs->rows && mean->ld == 1, "mgpi_predict_mean_dev: mean must be an (m x 1) vector");
    mgpi_terms_dev t;
    MGPI_TRY(mgpi_make_terms(terms, nterms, (int)X->cols, &t));
    MGPI_HIP_TRY(hipSetDevice(ctx->device));
    const int64_t m = Xs->rows, n = X->rows;
    const int64_t qblocks = (m + 255) / 256;
    // enough (query block, source chunk) workgroups to fill 256 CUs a few times over
    int64_t nchunks = (2048 + qblocks - 1) / qblocks;
    const int64_t maxchunks = (n + PM_CHUNK - 1) / PM_CHUNK;
    if (nchunks > maxchunks) nchunks = maxchunks;
    if (nchunks < 1) nchunks = 1;
    if (nchunks > 65535) nchunks = 65535;
    int64_t chunk_len = mgpi_roundup((n + nchunks - 1) / nchunks, PM_CHUNK);
    nchunks = (n + chunk_len - 1) / chunk_len;
    MGPI_TRY(mgpi_ensure_work(ctx, (size_t)nchunks * m * sizeof(double)));
    dim3 grid((unsigned)qblocks, (unsigned)nchunks), block(256);
    static const bool mfma_variant = getenv("MGPI_PREDICT_MFMA") && atoi(getenv("MGPI_PREDICT_MFMA")) != 0;
    if (mfma_variant && t.d == 3 && t.fast_mix >= 1) {          // the measured variant covers the benchmark shape (D = 3)
        dim3 g4((unsigned)((m + 63) / 64), (unsigned)nchunks);
        mgpi_prof_begin(ctx, 5);
        if (t.fast_mix == 1)
            hipLaunchKernelGGL((predict_mean_mfma_kernel<3, 1>), g4, block, 0, ctx->stream, t, Xs->d, Xs->ld, m, X->d, X->ld, n,
                               alpha->d, chunk_len, ctx->work);
        else
            hipLaunchKernelGGL((predict_mean_mfma_kernel<3, 2>), g4, block, 0, ctx->stream, t, Xs->d, Xs->ld, m, X->d, X->ld, n,
                               alpha->d, chunk_len, ctx->work);
        mgpi_prof_end(ctx, 5);
        mgpi_prof_flops(ctx, 5, (double)m * (double)n);
        MGPI_HIP_TRY(hipGetLastError());
        hipLaunchKernelGGL(reduce_chunks_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, ctx->work, m,
                           (int)nchunks, mean->d);
        MGPI_HIP_TRY(hipGetLastError());
        return 0;
    }
#define MGPI_PM_LAUNCH(DV, MV)                                                                                      \
    hipLaunchKernelGGL((predict_mean_kernel<DV, MV>), grid, block, 0, ctx->stream, t, Xs->d, Xs->ld, m, X->d, X->ld, n, \
                       alpha->d, chunk_len, ctx->work)
#define MGPI_PM_CASE(DV)                                                                                           \
    case DV:                                                                                                       \
        if (t.fast_mix == 1) MGPI_PM_LAUNCH(DV, 1);                                                                \
        else if (t.fast_mix == 2) MGPI_PM_LAUNCH(DV, 2);                                                           \
        else MGPI_PM_LAUNCH(DV, 0);                                                                                \
        break;
    mgpi_prof_begin(ctx, 5);
    switch (t.d) {
        MGPI_PM_CASE(1)
        MGPI_PM_CASE(2)
        MGPI_PM_CASE(3)
        MGPI_PM_CASE(4)
        MGPI_PM_CASE(5)
        MGPI_PM_CASE(6)
        MGPI_PM_CASE(7)
        MGPI_PM_CASE(8)
        default:            // more than 8 dimensions: run-time loop over the dimensions, still without the term loop
            if (t.fast_mix == 1) MGPI_PM_LAUNCH(0, 1);
            else if (t.fast_mix == 2) MGPI_PM_LAUNCH(0, 2);
            else MGPI_PM_LAUNCH(0, 0);
    }
#undef MGPI_PM_LAUNCH
#undef MGPI_PM_CASE
    mgpi_prof_end(ctx, 5);
    mgpi_prof_flops(ctx, 5, (double)m * (double)n);       // kernel evaluations (pairs), not flops
    MGPI_HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(reduce_chunks_kernel, dim3((unsigned)((m + 255) / 256)), dim3(256), 0, ctx->stream, ctx->work, m,
                       (int)nchunks, mean->d);
    MGPI_HIP_TRY(hipGetLastError());
    return 0;
}

// ---- variance -------------------------------------------------------------------------------------------
#define CN_ROWS 256   // rows of V summed per workgroup

// partial[rc][c] = sum over rows of chunk rc of V[r][c]^2 ; 64 columns x 4 row lanes per workgroup
__global__ __launch_bounds__(256) void colsq_kernel(const double* __restrict__ V, int64_t ldv, int64_t rows, int64_t cols,
                                                    double* __restrict__ partial) {
    __shared__ double part[4][64];
    const int lane = threadIdx.x & 63, sub = threadIdx.x >> 6;
    const int64_t c = (int64_t)blockIdx.x * 64 + lane;
    const int64_t r0 = (int64_t)blockIdx.y * CN_ROWS;
    int64_t r1 = r0 + CN_ROWS;
    if (r1 > rows) r1 = rows;
    double s = 0.0;
    if (c < cols)
        for (int64_t r = r0 + sub; r < r1; r += 4) {
            const double v = V[r * ldv + c];
            s += v * v;
        }
    part[sub][lane] = s;
    __syncthreads();
    if (sub == 0 && c < cols) partial[(int64_t)blockIdx.y * cols + c] = part[0][lane] + part[1][lane] + part[2][lane] + part[3][lane];
}

__global__ void sum_chunks_kernel(const double* __restrict__ partial, int64_t n, int nchunks, double* __restrict__ out) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double s = 0.0;
    for (int c = 0; c < nchunks; ++c) s += partial[(int64_t)c * n + i];
    out[i] = s;
}

// out[c] = sum over the (padded) rows of V of V[r][c]^2, c < V->cols: the column norms of a local slab of
// V = L^-1 K*^T in the multi-GPU variance sweep (the single-GPU path fuses this with the prior variance)
extern "C" int mgpi_colsq_dev(mgpi_ctx* ctx, const mgpi_mat* V, mgpi_mat* out) {
    MGPI_REQUIRE(ctx && V && out, "mgpi_colsq_dev: null pointer");
    MGPI_REQUIRE(out->ld == 1 && out->rows == V->cols, "mgpi_colsq_dev: out must be a (cols x 1) vector");
    MGPI_HIP_TRY(hipSetDevice(ctx->device));
    const int64_t rows = V->prows, cols = V->cols;
    const int64_t rchunks = (rows + CN_ROWS - 1) / CN_ROWS;
    MGPI_REQUIRE(rchunks <= 65535, "mgpi_colsq_dev: too many rows");
    MGPI_TRY(mgpi_ensure_work(ctx, (size_t)rchunks * cols * sizeof(double)));
    hipLaunchKernelGGL(colsq_kernel, dim3((unsigned)((cols + 63) / 64), (unsigned)rchunks), dim3(256), 0, ctx->stream, V->d, V->ld, rows,
                       cols, ctx->work);
    hipLaunchKernelGGL(sum_chunks_kernel, dim3((unsigned)((cols + 255) / 256)), dim3(256), 0, ctx->stream, ctx->work, cols, (int)rchunks,
                       out->d);
    MGPI_HIP_TRY(hipGetLastError());
    return 0;
}

// var[i] = k(xs_i, xs_i) - sum_c partial[c][i]
template <int D>
__global__ void var_finish_kernel(mgpi_terms_dev t, const double* __restrict__ xs, int64_t ldxs, int64_t m,
                                  const double* __restrict__ partial, int nchunks, int64_t pstride,
                                  double* __restrict__ var) {
    constexpr int DD = (D > 0) ? D : MGPI_MAX_D;
    const int nd = (D > 0) ? D : t.d;
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    double q[DD];
#pragma unroll
    for (int k = 0; k < DD; ++k) q[k] = (k < nd) ? xs[i * ldxs + k] : 0.0;
    double s = 0.0;
    for (int c = 0; c < nchunks; ++c) s += partial[(int64_t)c * pstride + i];
    var[i] = mgpi_cov_entry<D>(t, q, q) - s;
}

// sq[i] = sum_j V[i][j]^2, one wave per row (fixed order: lane partial sums over a stride of 64, then the shuffle tree)
__global__ __launch_bounds__(256) void rowsq_kernel(const double* __restrict__ V, int64_t ldv, int64_t rows, int64_t cols, double* __restrict__ sq) {
    const int lane = threadIdx.x & 63;
    const int64_t r = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= rows) return;
    const double* row = V + r * ldv;
    double s = 0.0;
    for (int64_t j = 2 * lane; j < cols; j += 128) {
        const double2 v = *reinterpret_cast<const double2*>(row + j);
        s = fma(v.x, v.x, s);
        s = fma(v.y, v.y, s);
    }
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    if (lane == 0) sq[r] = s;
}

// posterior variance from the rider rows mgpi_posv_riders_dev left in KR (rows [pn, pn + m) = V^T):
// var_i = k(x*_i, x*_i) - sum_j V^T[i][j]^2   - the diagonal of mgpi/interpolators.py:199
extern "C" int mgpi_riders_var_dev(mgpi_ctx* ctx, const mgpi_mat* KR, const mgpi_mat* Xs, const mgpi_kernel_desc* terms, int nterms,
                                   mgpi_mat* var) {
    MGPI_REQUIRE(ctx && KR && Xs && var && terms, "mgpi_riders_var_dev: null pointer");
    const int64_t pn = KR->pcols, m = Xs->rows;
    MGPI_REQUIRE(KR->factored && KR->rows == pn + m && var->rows == m && var->ld == 1,
                 "mgpi_riders_var_dev: KR must come from mgpi_posv_riders_dev with %lld rider rows", (long long)m);
    mgpi_terms_dev t;
    MGPI_TRY(mgpi_make_terms(terms, nterms, (int)Xs->cols, &t));
    MGPI_HIP_TRY(hipSetDevice(ctx->device));
    MGPI_TRY(mgpi_ensure_work(ctx, (size_t)m * sizeof(double)));
    double* sq = ctx->work;
    hipLaunchKernelGGL(rowsq_kernel, dim3((unsigned)((m + 3) / 4)), dim3(256), 0, ctx->stream, KR->d + pn * KR->ld, KR->ld, m, pn, sq);
    dim3 g2((unsigned)((m + 255) / 256));
#define MGPI_RV_CASE(DV)                                                                                                      \
    case DV:                                                                                                                  \
        hipLaunchKernelGGL(var_finish_kernel<DV>, g2, dim3(256), 0, ctx->stream, t, Xs->d, Xs->ld, m, sq, 1, m, var->d);    \
        break;
    switch (t.d) {
        MGPI_RV_CASE(1)
        MGPI_RV_CASE(2)
        MGPI_RV_CASE(3)
        MGPI_RV_CASE(4)
        default:
            hipLaunchKernelGGL(var_finish_kernel<0>, g2, dim3(256), 0, ctx->stream, t, Xs->d, Xs->ld, m, sq, 1, m, var->d);
    }
#undef MGPI_RV_CASE
    MGPI_HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int mgpi_predict_var_dev(mgpi_ctx* ctx, const mgpi_mat* Xs, const mgpi_mat* X, const mgpi_kernel_desc* terms,
                                    int nterms, const mgpi_mat* L, int64_t m_slab, mgpi_mat* var) {
    MGPI_REQUIRE(ctx && Xs && X && L && var, "mgpi_predict_var_dev: null pointer");
    MGPI_REQUIRE(Xs->cols == X->cols, "conflicting number of dimensions for x1 and x2 (%lld vs %lld)", (long long)Xs->cols,
                 (long long)X->cols);
    MGPI_REQUIRE(L->rows == X->rows && L->rows == L->cols, "mgpi_predict_var_dev: L (%lld) does not match X (%lld rows)",
                 (long long)L->rows, (long long)X->rows);
    MGPI_REQUIRE(var->rows == Xs->rows && var->ld == 1, "mgpi_predict_var_dev: var must be an (m x 1) vector");
    if (!L->factored) {
        mgpi_set_error("mgpi_predict_var_dev: L has not been factored by mgpi_potrf_dev");
        return MGPI_ERR_STATE;
    }
    mgpi_terms_dev t;
    MGPI_TRY(mgpi_make_terms(terms, nterms, (int)X->cols, &t));
    MGPI_HIP_TRY(hipSetDevice(ctx->device));
    const int64_t m = Xs->rows, n = X->rows, pn = L->prows;
    if (m_slab <= 0) {
        // default: two V slabs (work rows, solved rows) of ~4 GB each (288 GB of HBM: wide slabs mean fewer, larger update gemms - at N = 13160 a slab of
        // 8192 queries gives 64 x 103 tiles per update, 13 rounds of the chip with a 4 % tail), between 8192 and 65536 queries
        m_slab = (int64_t)(((size_t)4 << 30) / ((size_t)pn * sizeof(double)));
        if (m_slab < 8192) m_slab = 8192;
        if (m_slab > 65536) m_slab = 65536;
    }
    // keep the V slab under ~24 GB
    while (m_slab > 128 && (size_t)pn * (size_t)m_slab * sizeof(double) > ((size_t)24 << 30)) m_slab /= 2;
    m_slab = mgpi_roundup(m_slab, MGPI_TILE);
    if (m_slab > mgpi_roundup(m, MGPI_TILE)) m_slab = mgpi_roundup(m, MGPI_TILE);
    const int64_t rchunks = (pn + CN_ROWS - 1) / CN_ROWS;
    const size_t vbytes = (size_t)pn * m_slab * sizeof(double);
    const size_t pbytes = (size_t)rchunks * m_slab * sizeof(double);
    MGPI_TRY(mgpi_ensure_work(ctx, 2 * vbytes + pbytes));
    double* V = ctx->work;                                  // K(x, xs_slab), updated in place by the sweep
    double* Vs = ctx->work + (size_t)pn * m_slab;           // the solved rows L^-1 K(x, xs_slab) (see trsm_blocked, Xout)
    double* partial = ctx->work + 2 * (size_t)pn * m_slab;

    for (int64_t q0 = 0; q0 < m; q0 += m_slab) {
        const int64_t mq = (m - q0 < m_slab) ? (m - q0) : m_slab;
        const int64_t pq = mgpi_roundup(mq, MGPI_TILE);
        // V = K(x, xs_slab): rows = sources (padded rows written as 0), cols = queries
        MGPI_TRY(mgpi_launch_cov(ctx, t, X->d, X->ld, n, pn, Xs->d + q0 * Xs->ld, Xs->ld, mq, pq, 0, 0, 0, V, m_slab));
        // V <- L^-1 V
        int in_vs = 0;
        MGPI_TRY(mgpi_trsm_raw(ctx, L, V, m_slab, pq, 0, 1, Vs, &in_vs));     // variance path: inverse diagonal blocks allowed (chol.hip)
        dim3 grid((unsigned)((mq + 63) / 64), (unsigned)rchunks);
        hipLaunchKernelGGL(colsq_kernel, grid, dim3(256), 0, ctx->stream, in_vs ? Vs : V, m_slab, pn, mq, partial);
        MGPI_HIP_TRY(hipGetLastError());
        dim3 g2((unsigned)((mq + 255) / 256));
#define MGPI_VF_CASE(DV)                                                                                         \
    case DV:                                                                                                     \
        hipLaunchKernelGGL(var_finish_kernel<DV>, g2, dim3(256), 0, ctx->stream, t, Xs->d + q0 * Xs->ld, Xs->ld, mq, \
                           partial, (int)rchunks, mq, var->d + q0);                                             \
        break;
        switch (t.d) {
            MGPI_VF_CASE(1)
            MGPI_VF_CASE(2)
            MGPI_VF_CASE(3)
            MGPI_VF_CASE(4)
            default:
                hipLaunchKernelGGL(var_finish_kernel<0>, g2, dim3(256), 0, ctx->stream, t, Xs->d + q0 * Xs->ld, Xs->ld,
                                   mq, partial, (int)rchunks, mq, var->d + q0);
        }
#undef MGPI_VF_CASE
        MGPI_HIP_TRY(hipGetLastError());
    }
    return 0;
}

extern "C" int mgpi_condition_cov_dev(mgpi_ctx* ctx, const mgpi_mat* Xs, const mgpi_mat* X, const mgpi_kernel_desc* terms,
                                      int nterms, const mgpi_mat* L, mgpi_mat* cov) {
    MGPI_REQUIRE(ctx && Xs && X && L && cov, "mgpi_condition_cov_dev: null pointer");
    MGPI_REQUIRE(Xs->cols == X->cols, "conflicting number of dimensions for x1 and x2 (%lld vs %lld)", (long long)Xs->cols,
                 (long long)X->cols);
    MGPI_REQUIRE(L->rows == X->rows && L->rows == L->cols, "mgpi_condition_cov_dev: L does not match X");
    MGPI_REQUIRE(cov->rows == Xs->rows && cov->cols == Xs->rows && cov->pcols % MGPI_TILE == 0,
                 "mgpi_condition_cov_dev: cov must be a tile-padded (m x m) matrix");
    if (!L->factored) {
        mgpi_set_error("mgpi_condition_cov_dev: L has not been factored by mgpi_potrf_dev");
        return MGPI_ERR_STATE;
    }
    mgpi_terms_dev t;
    MGPI_TRY(mgpi_make_terms(terms, nterms, (int)X->cols, &t));
    MGPI_HIP_TRY(hipSetDevice(ctx->device));
    const int64_t m = Xs->rows, n = X->rows, pn = L->prows, pm = cov->prows;
    MGPI_REQUIRE(cov->pcols >= pm, "mgpi_condition_cov_dev: cov padding is not square");
    MGPI_TRY(mgpi_ensure_work(ctx, (size_t)pn * pm * sizeof(double)));
    double* V = ctx->work;
    // cov = K(xs, xs)   (interpolators.py:148)
    MGPI_TRY(mgpi_launch_cov(ctx, t, Xs->d, Xs->ld, m, pm, Xs->d, Xs->ld, m, pm, 1, 0, 0, cov->d, cov->ld));
    // V = L^-1 K(x, xs) (interpolators.py:157, transposed)
    MGPI_TRY(mgpi_launch_cov(ctx, t, X->d, X->ld, n, pn, Xs->d, Xs->ld, m, pm, 0, 0, 0, V, pm));
    MGPI_TRY(mgpi_trsm_raw(ctx, L, V, pm, pm, 0, 1, nullptr, nullptr));
    // cov -= V^T V      (interpolators.py:199)
    return mgpi_launch_gemm(ctx, 1, 1, 0, pm, pm, pn, V, pm, V, pm, cov->d, cov->ld);
}

// ---- throughput probes (bench.py roofline leg) ------------------------------------------------------------
typedef double double4_t __attribute__((ext_vector_type(4)));

// back-to-back fp64 MFMAs on register operands.  The accumulators are pinned to arch VGPRs ("+v"): left to hipcc
// they land in AGPRs, where this instruction issues at 148 instead of 64 cycles (tools/mfma_probe2.hip).
#define MGPI_MFMA_V(acc, a, b) asm volatile("v_mfma_f64_16x16x4_f64 %0, %1, %2, %0" : "+v"(acc) : "v"(a), "v"(b))
__global__ __launch_bounds__(256) void probe_mfma_kernel(int iters, const double* in, double* out) {
    double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0, c4 = c0, c5 = c0, c6 = c0, c7 = c0;
    const double a0 = in[threadIdx.x], a1 = in[256 + threadIdx.x], b0 = a1 + 1.0, b1 = a0 - 1.0, b2 = a0 * 0.5, b3 = a1 * 0.25;
    for (int it = 0; it < iters; ++it) {
        MGPI_MFMA_V(c0, a0, b0); MGPI_MFMA_V(c1, a0, b1); MGPI_MFMA_V(c2, a0, b2); MGPI_MFMA_V(c3, a0, b3);
        MGPI_MFMA_V(c4, a1, b0); MGPI_MFMA_V(c5, a1, b1); MGPI_MFMA_V(c6, a1, b2); MGPI_MFMA_V(c7, a1, b3);
    }
    const double4_t s4 = c0 + c1 + c2 + c3 + c4 + c5 + c6 + c7;
    const double s = s4[0] + s4[1] + s4[2] + s4[3];
    if (s == 12345.678) out[0] = s;   // keep the chain live
}

extern "C" int mgpi_probe_mfma_f64(mgpi_ctx* ctx, int iters, double* tflops) {
    MGPI_REQUIRE(ctx && tflops && iters > 0, "mgpi_probe_mfma_f64: bad argument");
    MGPI_HIP_TRY(hipSetDevice(ctx->device));
    const int blocks = 256 * 2;   // 2 workgroups of 4 waves per CU
    // operands: non-trivial values (the clock the chip holds depends on the data, cdna_hip_programming.md rule 25)
    MGPI_TRY(mgpi_ensure_work(ctx, 512 * sizeof(double)));
    double h[512];
    unsigned long long lcg = 88172645463325252ull;
    for (int i = 0; i < 512; ++i) {
        lcg = lcg * 6364136223846793005ull + 1442695040888963407ull;
        h[i] = (double)(lcg >> 11) / 9007199254740992.0 * 2.0 - 1.0;
    }
    MGPI_HIP_TRY(hipMemcpyAsync(ctx->work, h, sizeof(h), hipMemcpyHostToDevice, ctx->stream));
    MGPI_HIP_TRY(hipStreamSynchronize(ctx->stream));
    hipLaunchKernelGGL(probe_mfma_kernel, dim3(blocks), dim3(256), 0, ctx->stream, 16, ctx->work, ctx->d_scal);   // warm-up
    MGPI_TRY(mgpi_ctx_tic(ctx));
    hipLaunchKernelGGL(probe_mfma_kernel, dim3(blocks), dim3(256), 0, ctx->stream, iters, ctx->work, ctx->d_scal);
    double ms = 0.0;
    MGPI_TRY(mgpi_ctx_toc(ctx, &ms));
    const double flops = (double)blocks * 4.0 * (double)iters * 8.0 * 2048.0;   // 16*16*4*2 per MFMA
    *tflops = flops / (ms * 1e-3) / 1e12;
    return 0;
}

__global__ __launch_bounds__(256) void probe_read_kernel(const double2* __restrict__ p, int64_t n2, double* out) {
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) {
        const double2 v = p[i];
        s += v.x + v.y;
    }
    if (s == 12345.678) out[0] = s;
}
__global__ __launch_bounds__(256) void probe_write_kernel(double2* __restrict__ p, int64_t n2) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256)
        p[i] = make_double2(1.0, 2.0);
}

extern "C" int mgpi_probe_hbm(mgpi_ctx* ctx, int64_t bytes, int iters, double* read_gbs, double* write_gbs) {
    MGPI_REQUIRE(ctx && bytes >= (1 << 20) && iters > 0, "mgpi_probe_hbm: bad argument");
    MGPI_HIP_TRY(hipSetDevice(ctx->device));
    MGPI_TRY(mgpi_ensure_work(ctx, (size_t)bytes));
    const int64_t n2 = bytes / 16;
    double ms = 0.0;
    hipLaunchKernelGGL(probe_write_kernel, dim3(2048), dim3(256), 0, ctx->stream, (double2*)ctx->work, n2);
    MGPI_TRY(mgpi_ctx_tic(ctx));
    for (int i = 0; i < iters; ++i)
        hipLaunchKernelGGL(probe_write_kernel, dim3(2048), dim3(256), 0, ctx->stream, (double2*)ctx->work, n2);
    MGPI_TRY(mgpi_ctx_toc(ctx, &ms));
    if (write_gbs) *write_gbs = (double)bytes * iters / (ms * 1e-3) / 1e9;
    MGPI_TRY(mgpi_ctx_tic(ctx));
    for (int i = 0; i < iters; ++i)
        hipLaunchKernelGGL(probe_read_kernel, dim3(2048), dim3(256), 0, ctx->stream, (const double2*)ctx->work, n2, ctx->d_scal);
    MGPI_TRY(mgpi_ctx_toc(ctx, &ms));
    if (read_gbs) *read_gbs = (double)bytes * iters / (ms * 1e-3) / 1e9;
    return 0;
}
