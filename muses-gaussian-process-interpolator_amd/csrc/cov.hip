// Kernel-matrix build: the device form of Interpolator._x2cov (mgpi/interpolators.py:208-300) and of the
// row-paired Kernel.cov (mgpi/kernels.py:89-93).  The reference materialises the n1*n2 row pairs; here each
// workgroup owns one 64x64 output tile, stages the two coordinate slabs in LDS as SoA and never forms pairs.
#include "cov_device.h"
#include <stdlib.h>

#define COV_T 64   // output tile edge

// out[i][j] = k(x1_i, x2_j) for i < n1, j < n2; padded entries (i >= n1 or j >= n2, up to p1 x p2) are written
// as 0, or as the identity when pad_identity (so that the padded K stays positive definite, its Cholesky
// factor is [[L,0],[0,I]] and logdet is unchanged).
template <int D, int MIX>
__global__ __launch_bounds__(256) void cov_tile_kernel(mgpi_terms_dev t, const double* __restrict__ x1, int64_t ldx1,
                                                       int64_t n1, const double* __restrict__ x2, int64_t ldx2,
                                                       int64_t n2, int lower_only, int pad_identity,
                                                       double* __restrict__ out, int64_t ldo) {
    constexpr int DD = (D > 0) ? D : MGPI_MAX_D;
    __shared__ double s1[DD][COV_T];
    __shared__ __attribute__((aligned(16))) double s2[DD][COV_T];
    const int nd = (D > 0) ? D : t.d;
    int64_t row0, col0;
    if (lower_only) {   // 1-D grid over the tiles on/below the diagonal only (potrf never reads the others): at N = 32768 the
        int ti, tj;     // full grid launched 131 k workgroups just to have them exit
        tri_decode(blockIdx.x, ti, tj);
        row0 = (int64_t)ti * COV_T;
        col0 = (int64_t)tj * COV_T;
    } else {
        row0 = (int64_t)blockIdx.y * COV_T;
        col0 = (int64_t)blockIdx.x * COV_T;
    }

    const int tid = threadIdx.x;
    for (int idx = tid; idx < COV_T * nd; idx += 256) {
        const int r = idx / nd, k = idx - r * nd;
        s1[k][r] = (row0 + r < n1) ? x1[(row0 + r) * ldx1 + k] : 0.0;
        s2[k][r] = (col0 + r < n2) ? x2[(col0 + r) * ldx2 + k] : 0.0;
    }
    __syncthreads();

    const int tx = tid & 15, ty = tid >> 4;   // 16 x 16 threads; each owns 4 rows x (2 x 2) columns
    double a[4][DD], b[4][DD];
#pragma unroll
    for (int k = 0; k < DD; ++k) {
        if (k < nd) {
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r][k] = s1[k][ty * 4 + r];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const double2 v = *reinterpret_cast<const double2*>(&s2[k][2 * tx + 32 * c]);
                b[2 * c][k] = v.x;
                b[2 * c + 1][k] = v.y;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t i = row0 + ty * 4 + r;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int64_t j = col0 + 2 * tx + 32 * c;
            double2 v;
            v.x = mgpi_cov_entry<D, MIX>(t, a[r], b[2 * c]);
            v.y = mgpi_cov_entry<D, MIX>(t, a[r], b[2 * c + 1]);
            if (i >= n1 || j >= n2) v.x = (pad_identity && i == j) ? 1.0 : 0.0;
            if (i >= n1 || j + 1 >= n2) v.y = (pad_identity && i == j + 1) ? 1.0 : 0.0;
            *reinterpret_cast<double2*>(&out[i * ldo + j]) = v;
        }
    }
}

// The large-D form of the kernel-matrix build (north_star: "wavefront shuffles for the D-reduction"), default for D > 8.  Four lanes (a
// DPP quad) share one output row of the tile and split the DIMENSIONS: lane q accumulates sum_{k = q mod 4} (dx_k / l_k)^2
// for FOUR neighbouring columns, a 4 x 4 transpose-reduction inside the quad (three 64-bit DPP exchanges) leaves lane q
// with the complete sum of column q, and every lane then evaluates exp for its own entry - no redundant exp, the same
// number of vector instructions per entry as the register loop plus ~9 for the exchange.  What it buys is registers: the
// general-D register kernel holds 2 x 4 x 16 coordinates per thread (256 VGPRs), this one 4 + 16.  Single
// SquaredExponential (+ WhiteNoise) kernels only; the summation order differs from the left-to-right loop, which for
// D >= 8 differs from numpy's 8-way pairwise order anyway.  Measured (N = 8192, SE + WhiteNoise, full matrix): D = 16
// 0.48 ms against 4.58 ms for the register loop, D = 8 0.31 against 2.26; D = 3 0.23 against 0.12 (tools/cov_shuffle_probe.py).
template <int XOR>
__device__ __forceinline__ double cov_quad_xor(double v) {
    constexpr int ctrl = (XOR == 1) ? 0xB1 /* quad_perm [1,0,3,2] */ : 0x4E /* quad_perm [2,3,0,1] */;
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), ctrl, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), ctrl, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

__global__ __launch_bounds__(256) void cov_tile_shuffle_kernel(mgpi_terms_dev t, const double* __restrict__ x1, int64_t ldx1,
                                                               int64_t n1, const double* __restrict__ x2, int64_t ldx2,
                                                               int64_t n2, int lower_only, int pad_identity,
                                                               double* __restrict__ out, int64_t ldo) {
    __shared__ double s1[MGPI_MAX_D][COV_T];
    __shared__ double s2[MGPI_MAX_D][COV_T];
    const int nd = t.d;
    int64_t row0, col0;
    if (lower_only) {
        int ti, tj;
        tri_decode(blockIdx.x, ti, tj);
        row0 = (int64_t)ti * COV_T;
        col0 = (int64_t)tj * COV_T;
    } else {
        row0 = (int64_t)blockIdx.y * COV_T;
        col0 = (int64_t)blockIdx.x * COV_T;
    }
    const int tid = threadIdx.x;
    for (int idx = tid; idx < COV_T * nd; idx += 256) {
        const int r = idx / nd, k = idx - r * nd;
        s1[k][r] = (row0 + r < n1) ? x1[(row0 + r) * ldx1 + k] : 0.0;
        s2[k][r] = (col0 + r < n2) ? x2[(col0 + r) * ldx2 + k] : 0.0;
    }
    __syncthreads();
    const int q = tid & 3, row = tid >> 2;                       // lane part, tile row
    double a[4], l2[4], il2[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int k = q + 4 * u;
        a[u] = (k < nd) ? s1[k][row] : 0.0;
        l2[u] = (k < nd) ? t.l2[0][k] : 1.0;
        il2[u] = (k < nd) ? t.inv_l2[0][k] : 1.0;
    }
    const bool white = t.nterms == 2;
    const int64_t i = row0 + row;
    for (int cg = 0; cg < COV_T; cg += 4) {
        double p[4];
        bool same[4];
#pragma unroll
        for (int cc = 0; cc < 4; ++cc) {
            double r = 0.0;
            bool sm = true;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int k = q + 4 * u;
                if (k < nd) {
                    const double b = s2[k][cg + cc];
                    const double df = a[u] - b;
                    sm = sm && (a[u] == b);
                    r = __dadd_rn(r, mgpi_div_rn(__dmul_rn(df, df), l2[u], il2[u]));
                }
            }
            p[cc] = r;
            same[cc] = sm;
        }
        // 4 x 4 transpose-reduction: after the xor-1 exchange lane q holds the two columns with the parity of q, after
        // the xor-2 exchange its own column
        // (every exchange is executed by all four lanes, outside any lane-dependent branch: a DPP read from a lane that
        // sits out a divergent branch returns 0)
        const double x0 = cov_quad_xor<1>(p[0]), x1 = cov_quad_xor<1>(p[1]), x2 = cov_quad_xor<1>(p[2]), x3 = cov_quad_xor<1>(p[3]);
        const double e0 = (q & 1) ? p[1] + x1 : p[0] + x0;        // column (q & 1), lanes q and q ^ 1
        const double e1 = (q & 1) ? p[3] + x3 : p[2] + x2;        // column 2 + (q & 1)
        const double y0 = cov_quad_xor<2>(e0), y1 = cov_quad_xor<2>(e1);
        const double f = (q & 2) ? e1 + y1 : e0 + y0;             // column q, all four lanes
        double val = __dmul_rn(t.sig2[0], mgpi_exp_neg(f));
        if (white) {
            // all dimensions equal <=> every part found its share equal: reduce the flags the same way
            const int f0 = same[0], f1 = same[1], f2 = same[2], f3 = same[3];
            const int g0 = __builtin_amdgcn_mov_dpp(f0, 0xB1, 0xf, 0xf, true), g1 = __builtin_amdgcn_mov_dpp(f1, 0xB1, 0xf, 0xf, true);
            const int g2 = __builtin_amdgcn_mov_dpp(f2, 0xB1, 0xf, 0xf, true), g3 = __builtin_amdgcn_mov_dpp(f3, 0xB1, 0xf, 0xf, true);
            const int s0 = (q & 1) ? (f1 & g1) : (f0 & g0), s1_ = (q & 1) ? (f3 & g3) : (f2 & g2);
            const int h0 = __builtin_amdgcn_mov_dpp(s0, 0x4E, 0xf, 0xf, true), h1 = __builtin_amdgcn_mov_dpp(s1_, 0x4E, 0xf, 0xf, true);
            const int all = (q & 2) ? (s1_ & h1) : (s0 & h0);
            val = __dadd_rn(val, all ? t.sig2[1] : 0.0);
        }
        const int64_t j = col0 + cg + q;
        if (i >= n1 || j >= n2) val = (pad_identity && i == j) ? 1.0 : 0.0;
        out[i * ldo + j] = val;
    }
}

// Strided batch of LOWER kernel matrices over ONE coordinate table: problem z = blockIdx.z uses the kernel description
// tb[z] (device memory) and writes out + z * stride.  The hyper-parameter loops of the reference evaluate the
// likelihood of 2 * num_dim walkers per step at fixed x (mgpi/interpolators.py:513-514, closure at :369-387).
template <int D>
__global__ __launch_bounds__(256) void cov_tile_batch_kernel(const mgpi_terms_dev* __restrict__ tb, const double* __restrict__ x,
                                                             int64_t ldx, int64_t n, double* __restrict__ out, int64_t ldo,
                                                             int64_t stride) {
    constexpr int DD = (D > 0) ? D : MGPI_MAX_D;
    __shared__ double s1[DD][COV_T];
    __shared__ __attribute__((aligned(16))) double s2[DD][COV_T];
    const mgpi_terms_dev& t = tb[blockIdx.z];
    const int nd = (D > 0) ? D : t.d;
    const int64_t row0 = (int64_t)blockIdx.y * COV_T, col0 = (int64_t)blockIdx.x * COV_T;
    if (col0 > row0) return;
    out += (int64_t)blockIdx.z * stride;
    const int tid = threadIdx.x;
    for (int idx = tid; idx < COV_T * nd; idx += 256) {
        const int r = idx / nd, k = idx - r * nd;
        s1[k][r] = (row0 + r < n) ? x[(row0 + r) * ldx + k] : 0.0;
        s2[k][r] = (col0 + r < n) ? x[(col0 + r) * ldx + k] : 0.0;
    }
    __syncthreads();
    const int tx = tid & 15, ty = tid >> 4;
    double a[4][DD], b[4][DD];
#pragma unroll
    for (int k = 0; k < DD; ++k) {
        if (k < nd) {
#pragma unroll
            for (int r = 0; r < 4; ++r) a[r][k] = s1[k][ty * 4 + r];
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const double2 v = *reinterpret_cast<const double2*>(&s2[k][2 * tx + 32 * c]);
                b[2 * c][k] = v.x;
                b[2 * c + 1][k] = v.y;
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int64_t i = row0 + ty * 4 + r;
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int64_t j = col0 + 2 * tx + 32 * c;
            double2 v;
            v.x = mgpi_cov_entry<D>(t, a[r], b[2 * c]);
            v.y = mgpi_cov_entry<D>(t, a[r], b[2 * c + 1]);
            if (i >= n || j >= n) v.x = (i == j) ? 1.0 : 0.0;               // unit diagonal in the padding
            if (i >= n || j + 1 >= n) v.y = (i == j + 1) ? 1.0 : 0.0;
            *reinterpret_cast<double2*>(&out[i * ldo + j]) = v;
        }
    }
}

int mgpi_launch_cov_batch(mgpi_ctx* ctx, const mgpi_terms_dev* tb_dev, int d, const double* x, int64_t ldx, int64_t n, int64_t pn,
                          double* out, int64_t ldo, int64_t stride, int batch) {
    MGPI_REQUIRE(pn % COV_T == 0 && batch >= 1 && batch < 65536, "cov batch: pn=%lld batch=%d", (long long)pn, batch);
    dim3 grid((unsigned)(pn / COV_T), (unsigned)(pn / COV_T), (unsigned)batch), block(256);
    switch (d) {
        case 1: hipLaunchKernelGGL(cov_tile_batch_kernel<1>, grid, block, 0, ctx->stream, tb_dev, x, ldx, n, out, ldo, stride); break;
        case 2: hipLaunchKernelGGL(cov_tile_batch_kernel<2>, grid, block, 0, ctx->stream, tb_dev, x, ldx, n, out, ldo, stride); break;
        case 3: hipLaunchKernelGGL(cov_tile_batch_kernel<3>, grid, block, 0, ctx->stream, tb_dev, x, ldx, n, out, ldo, stride); break;
        case 4: hipLaunchKernelGGL(cov_tile_batch_kernel<4>, grid, block, 0, ctx->stream, tb_dev, x, ldx, n, out, ldo, stride); break;
        default: hipLaunchKernelGGL(cov_tile_batch_kernel<0>, grid, block, 0, ctx->stream, tb_dev, x, ldx, n, out, ldo, stride);
    }
    MGPI_HIP_TRY(hipGetLastError());
    return 0;
}

template <int D>
__global__ __launch_bounds__(256) void cov_paired_kernel(mgpi_terms_dev t, const double* __restrict__ x1,
                                                         const double* __restrict__ x2, int64_t p,
                                                         double* __restrict__ out) {
    constexpr int DD = (D > 0) ? D : MGPI_MAX_D;
    const int nd = (D > 0) ? D : t.d;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < p; i += (int64_t)gridDim.x * 256) {
        double a[DD], b[DD];
#pragma unroll
        for (int k = 0; k < DD; ++k)
            if (k < nd) {
                a[k] = x1[i * nd + k];
                b[k] = x2[i * nd + k];
            }
        out[i] = mgpi_cov_entry<D>(t, a, b);
    }
}

int mgpi_launch_cov(mgpi_ctx* ctx, const mgpi_terms_dev& t, const double* x1, int64_t ldx1, int64_t n1, int64_t p1,
                    const double* x2, int64_t ldx2, int64_t n2, int64_t p2, int symmetric, int lower_only,
                    int pad_identity, double* out, int64_t ldo) {
    (void)symmetric;
    MGPI_REQUIRE(p1 % COV_T == 0 && p2 % COV_T == 0, "cov: padded output %lld x %lld is not a multiple of %d",
                 (long long)p1, (long long)p2, COV_T);
    MGPI_REQUIRE(p1 / COV_T <= 65535, "cov: too many row tiles");
    dim3 grid((unsigned)(p2 / COV_T), (unsigned)(p1 / COV_T)), block(256);
    if (lower_only) {
        MGPI_REQUIRE(p1 == p2, "cov(lower): the matrix must be square");
        const int64_t tm = p1 / COV_T;
        grid = dim3((unsigned)(tm * (tm + 1) / 2), 1);
    }
    // the quad-shuffle kernel is the default beyond D = 8, where there is no register kernel with a compile-time D (the
    // run-time-D register loop holds 2 x 4 x 16 coordinates per thread); up to 8 dimensions the register kernels are
    // faster (D = 8: 0.21 ms against 0.31 at N = 8192) and sum in the reference's own order.  MGPI_COV_SHUFFLE=1 forces it for every D, =0 switches it off.
    static const int shuffle_env = getenv("MGPI_COV_SHUFFLE") ? atoi(getenv("MGPI_COV_SHUFFLE")) : -1;
    const bool shuffle_variant = shuffle_env == 1 || (shuffle_env != 0 && t.d > 8);
    if (shuffle_variant && t.fast_mix >= 1) {
        hipLaunchKernelGGL(cov_tile_shuffle_kernel, grid, block, 0, ctx->stream, t, x1, ldx1, n1, x2, ldx2, n2, lower_only, pad_identity,
                           out, ldo);
        MGPI_HIP_TRY(hipGetLastError());
        return 0;
    }
#define MGPI_COV_LAUNCH(DV, MV)                                                                                   \
    hipLaunchKernelGGL((cov_tile_kernel<DV, MV>), grid, block, 0, ctx->stream, t, x1, ldx1, n1, x2, ldx2, n2,     \
                       lower_only, pad_identity, out, ldo)
#define MGPI_COV_CASE(DV)                                                                                         \
    case DV:                                                                                                      \
        if (t.fast_mix == 1) MGPI_COV_LAUNCH(DV, 1);                                                              \
        else if (t.fast_mix == 2) MGPI_COV_LAUNCH(DV, 2);                                                         \
        else MGPI_COV_LAUNCH(DV, 0);                                                                              \
        break;
    switch (t.d) {
        MGPI_COV_CASE(1)
        MGPI_COV_CASE(2)
        MGPI_COV_CASE(3)
        MGPI_COV_CASE(4)
        MGPI_COV_CASE(5)
        MGPI_COV_CASE(6)
        MGPI_COV_CASE(7)
        MGPI_COV_CASE(8)
        default:
            if (t.fast_mix == 1) MGPI_COV_LAUNCH(0, 1);
            else if (t.fast_mix == 2) MGPI_COV_LAUNCH(0, 2);
            else MGPI_COV_LAUNCH(0, 0);
    }
#undef MGPI_COV_LAUNCH
#undef MGPI_COV_CASE
    MGPI_HIP_TRY(hipGetLastError());
    return 0;
}

extern "C" int mgpi_cov_matrix_dev(mgpi_ctx* ctx, const mgpi_mat* X1, const mgpi_mat* X2, const mgpi_kernel_desc* terms,
                                   int nterms, int lower_only, mgpi_mat* out) {
    MGPI_REQUIRE(ctx && X1 && out, "mgpi_cov_matrix_dev: null pointer");
    const mgpi_mat* Xb = X2 ? X2 : X1;
    MGPI_REQUIRE(X1->cols == Xb->cols, "conflicting number of dimensions for x1 and x2 (%lld vs %lld)",
                 (long long)X1->cols, (long long)Xb->cols);   // interpolators.py:219
    MGPI_REQUIRE(out->rows == X1->rows && out->cols == Xb->rows, "mgpi_cov_matrix_dev: out is %lld x %lld, expected %lld x %lld",
                 (long long)out->rows, (long long)out->cols, (long long)X1->rows, (long long)Xb->rows);
    MGPI_REQUIRE(!lower_only || X2 == nullptr, "mgpi_cov_matrix_dev: lower_only needs the symmetric form (X2 == NULL)");
    MGPI_REQUIRE(out->pcols % COV_T == 0, "mgpi_cov_matrix_dev: out must be a wide (tile-padded) matrix, cols > %d", MGPI_MAX_D);
    mgpi_terms_dev t;
    MGPI_TRY(mgpi_make_terms(terms, nterms, (int)X1->cols, &t));
    MGPI_HIP_TRY(hipSetDevice(ctx->device));
    const int sym = (X2 == nullptr);
    mgpi_mat_invalidate(out);
    return mgpi_launch_cov(ctx, t, X1->d, X1->ld, X1->rows, out->prows, Xb->d, Xb->ld, Xb->rows, out->pcols, sym,
                           lower_only, /*pad_identity=*/sym, out->d, out->ld);
}

// K(x, x; sterms) lower tiles into rows [0, n) of KR and K(x*, x; kterms) into rows [pn, pn + m): the matrix mgpi_posv_riders_dev
// factors (the three _x2cov calls of Interpolator.condition, mgpi/interpolators.py:148-169, minus K** whose diagonal is taken later)
extern "C" int mgpi_cov_riders_dev(mgpi_ctx* ctx, const mgpi_mat* X, const mgpi_mat* Xs, const mgpi_kernel_desc* kterms, int nk,
                                   const mgpi_kernel_desc* sterms, int ns, mgpi_mat* KR) {
    MGPI_REQUIRE(ctx && X && Xs && KR && kterms && sterms, "mgpi_cov_riders_dev: null pointer");
    MGPI_REQUIRE(Xs->cols == X->cols, "conflicting number of dimensions for x1 and x2 (%lld vs %lld)", (long long)Xs->cols,
                 (long long)X->cols);   // interpolators.py:219
    const int64_t n = X->rows, m = Xs->rows, pn = KR->pcols;
    MGPI_REQUIRE(KR->cols == n && pn % COV_T == 0 && KR->prows >= pn + mgpi_roundup(m, COV_T) && KR->rows == pn + m,
                 "mgpi_cov_riders_dev: KR must be a wide (pn + m) x n matrix (got %lld x %lld for n = %lld, m = %lld)", (long long)KR->rows,
                 (long long)KR->cols, (long long)n, (long long)m);
    mgpi_terms_dev tk, ts;
    MGPI_TRY(mgpi_make_terms(kterms, nk, (int)X->cols, &tk));
    MGPI_TRY(mgpi_make_terms(sterms, ns, (int)X->cols, &ts));
    MGPI_HIP_TRY(hipSetDevice(ctx->device));
    mgpi_mat_invalidate(KR);
    MGPI_TRY(mgpi_launch_cov(ctx, ts, X->d, X->ld, n, pn, X->d, X->ld, n, pn, 1, 1, 1, KR->d, KR->ld));
    return mgpi_launch_cov(ctx, tk, Xs->d, Xs->ld, m, KR->prows - pn, X->d, X->ld, n, pn, 0, 0, 0, KR->d + pn * KR->ld, KR->ld);
}

extern "C" int mgpi_cov_matrix(mgpi_ctx* ctx, const double* x1, int64_t n1, const double* x2, int64_t n2, int d,
                               const mgpi_kernel_desc* terms, int nterms, double* out) {
    MGPI_REQUIRE(ctx && x1 && out, "mgpi_cov_matrix: null pointer");
    MGPI_REQUIRE(n1 >= 1 && (x2 == nullptr || n2 >= 1), "mgpi_cov_matrix: empty input");
    if (x2 == nullptr) n2 = n1;
    mgpi_mat *X1 = nullptr, *X2 = nullptr, *K = nullptr;
    int st = mgpi_mat_alloc(ctx, n1, d, &X1);
    if (st == 0) st = mgpi_mat_upload(X1, x1, d);
    if (st == 0 && x2) st = mgpi_mat_alloc(ctx, n2, d, &X2);
    if (st == 0 && x2) st = mgpi_mat_upload(X2, x2, d);
    if (st == 0) st = mgpi_mat_alloc_wide(ctx, n1, n2, &K);
    if (st == 0) st = mgpi_cov_matrix_dev(ctx, X1, X2, terms, nterms, 0, K);
    if (st == 0) st = mgpi_mat_download(K, out, n2);
    mgpi_mat_free(X1);
    mgpi_mat_free(X2);
    mgpi_mat_free(K);
    return st;
}

extern "C" int mgpi_cov_paired(mgpi_ctx* ctx, const double* x1, const double* x2, int64_t p, int d,
                               const mgpi_kernel_desc* terms, int nterms, double* out) {
    MGPI_REQUIRE(ctx && x1 && x2 && out, "mgpi_cov_paired: null pointer");
    MGPI_REQUIRE(p >= 1, "mgpi_cov_paired: empty input");
    mgpi_terms_dev t;
    MGPI_TRY(mgpi_make_terms(terms, nterms, d, &t));
    MGPI_HIP_TRY(hipSetDevice(ctx->device));
    const size_t xb = (size_t)p * d * sizeof(double), ob = (size_t)p * sizeof(double);
    MGPI_TRY(mgpi_ensure_work(ctx, 2 * xb + ob));
    double* d1 = ctx->work;
    double* d2 = d1 + (size_t)p * d;
    double* dout = d2 + (size_t)p * d;
    MGPI_HIP_TRY(hipMemcpyAsync(d1, x1, xb, hipMemcpyHostToDevice, ctx->stream));
    MGPI_HIP_TRY(hipMemcpyAsync(d2, x2, xb, hipMemcpyHostToDevice, ctx->stream));
    int64_t blocks = (p + 255) / 256;
    if (blocks > 4096) blocks = 4096;
#define MGPI_PAIR_CASE(DV)                                                                                          \
    case DV:                                                                                                        \
        hipLaunchKernelGGL(cov_paired_kernel<DV>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, t, d1, d2, p, dout); \
        break;
    switch (d) {
        MGPI_PAIR_CASE(1)
        MGPI_PAIR_CASE(2)
        MGPI_PAIR_CASE(3)
        MGPI_PAIR_CASE(4)
        default:
            hipLaunchKernelGGL(cov_paired_kernel<0>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, t, d1, d2, p, dout);
    }
#undef MGPI_PAIR_CASE
    MGPI_HIP_TRY(hipGetLastError());
    MGPI_HIP_TRY(hipMemcpyAsync(out, dout, ob, hipMemcpyDeviceToHost, ctx->stream));
    MGPI_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

// block form for the multi-GPU driver: out (every padded entry) = k(X1[i], X2[j]) for i < n1_valid, j < n2_valid,
// zero elsewhere (padded global indices get their unit diagonal from the driver, which knows the global layout)
extern "C" int mgpi_cov_block_dev(mgpi_ctx* ctx, const mgpi_mat* X1, int64_t n1_valid, const mgpi_mat* X2, int64_t n2_valid,
                                  const mgpi_kernel_desc* terms, int nterms, mgpi_mat* out) {
    MGPI_REQUIRE(ctx && X1 && X2 && out, "mgpi_cov_block_dev: null pointer");
    MGPI_REQUIRE(X1->cols == X2->cols, "conflicting number of dimensions for x1 and x2 (%lld vs %lld)", (long long)X1->cols,
                 (long long)X2->cols);
    MGPI_REQUIRE(n1_valid >= 0 && n2_valid >= 0 && n1_valid <= X1->prows && n2_valid <= X2->prows &&
                     n1_valid <= out->prows && n2_valid <= out->pcols,
                 "mgpi_cov_block_dev: valid extents %lld x %lld exceed the operands", (long long)n1_valid, (long long)n2_valid);
    MGPI_REQUIRE(out->prows % COV_T == 0 && out->pcols % COV_T == 0, "mgpi_cov_block_dev: out must be tile padded");
    mgpi_terms_dev t;
    MGPI_TRY(mgpi_make_terms(terms, nterms, (int)X1->cols, &t));
    MGPI_HIP_TRY(hipSetDevice(ctx->device));
    mgpi_mat_invalidate(out);
    return mgpi_launch_cov(ctx, t, X1->d, X1->ld, n1_valid, out->prows, X2->d, X2->ld, n2_valid, out->pcols, 0, 0, 0,
                           out->d, out->ld);
}
