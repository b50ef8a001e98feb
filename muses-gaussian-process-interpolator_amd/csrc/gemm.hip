// fp64 MFMA GEMM family:  C -= op(A) * op(B)   on v_mfma_f64_16x16x4_f64 (gfx950).
//
// This is where the O(N^3) of the hot path lives: the Cholesky trailing update (syrk, lower tiles only), the
// panel-internal updates, the multi-rhs triangular solves of predict_var and the V^T V of condition().
// It takes the role np.linalg.inv's LAPACK GEMMs play in the reference (mgpi/interpolators.py:92,179,199).
//
// Layout: 256 threads = 4 waves in a 2x2 grid; workgroup tile BM x BN, wave tile (BM/2) x (BN/2) made of
// 16x16 MFMA tiles; BK = 16 contraction elements per LDS stage (4 MFMA k-steps), two LDS stages.
// MFMA f64 16x16x4 operand map (cdna_hip_programming.md section 3): lane l supplies A[l&15][l>>4] and
// B[l>>4][l&15]; it receives D[(l>>4) + 4r][l&15], r = 0..3.
//
// LDS images (doubles; strides chosen so each 32-lane half of a ds_read_b64 touches 32 distinct 8-byte banks):
//   row-major operand  ([row][k], k contiguous in HBM):  stride 18 : row*18 mod 32 is a permutation of the evens
//   k-major operand    ([k][col], col contiguous in HBM): stride BN+16: consecutive k rows land 32 banks apart
#include "mgpi_internal.h"

typedef double double4_t __attribute__((ext_vector_type(4)));

#define GEMM_BK 16
#define GEMM_SR 18            // LDS row stride of a row-major operand tile

template <int BT, bool KMAJOR>
struct TileGeom {
    static constexpr int stride = KMAJOR ? (BT + 16) : GEMM_SR;
    static constexpr int doubles = KMAJOR ? GEMM_BK * (BT + 16) : BT * GEMM_SR;
    static constexpr int vecs = BT * GEMM_BK / 2 / 256;   // 16-byte loads per thread per stage
};

// global -> registers for one BT x BK operand slab
template <int BT, bool KMAJOR>
__device__ __forceinline__ void tile_load(const double* __restrict__ g, int64_t ld, int64_t t0, int64_t k0, int tid,
                                          double2 (&r)[TileGeom<BT, KMAJOR>::vecs]) {
#pragma unroll
    for (int s = 0; s < TileGeom<BT, KMAJOR>::vecs; ++s) {
        const int idx = tid + 256 * s;
        if (KMAJOR) {   // stored (k x BT): row = k index, BT/2 double2 per row
            const int kr = idx / (BT / 2), c2 = idx % (BT / 2);
            r[s] = *reinterpret_cast<const double2*>(g + (k0 + kr) * ld + t0 + 2 * c2);
        } else {        // stored (BT x k): 8 double2 per row
            const int row = idx >> 3, k2 = idx & 7;
            r[s] = *reinterpret_cast<const double2*>(g + (t0 + row) * ld + k0 + 2 * k2);
        }
    }
}

template <int BT, bool KMAJOR>
__device__ __forceinline__ void tile_store(double* __restrict__ lds, int tid,
                                           const double2 (&r)[TileGeom<BT, KMAJOR>::vecs]) {
#pragma unroll
    for (int s = 0; s < TileGeom<BT, KMAJOR>::vecs; ++s) {
        const int idx = tid + 256 * s;
        if (KMAJOR) {
            const int kr = idx / (BT / 2), c2 = idx % (BT / 2);
            *reinterpret_cast<double2*>(lds + kr * (BT + 16) + 2 * c2) = r[s];
        } else {
            const int row = idx >> 3, k2 = idx & 7;
            *reinterpret_cast<double2*>(lds + row * GEMM_SR + 2 * k2) = r[s];
        }
    }
}

// fragment element (t = index along the tile edge, k = contraction index) of an LDS operand image
template <int BT, bool KMAJOR>
__device__ __forceinline__ double frag(const double* __restrict__ lds, int t, int k) {
    return KMAJOR ? lds[k * (BT + 16) + t] : lds[t * GEMM_SR + k];
}

// linear lower-triangle tile index -> (i, j), j <= i
__device__ __forceinline__ void tri_decode(int64_t idx, int& ti, int& tj) {
    int i = (int)((sqrt(8.0 * (double)idx + 1.0) - 1.0) * 0.5);
    while ((int64_t)(i + 1) * (i + 2) / 2 <= idx) ++i;
    while ((int64_t)i * (i + 1) / 2 > idx) --i;
    ti = i;
    tj = (int)(idx - (int64_t)i * (i + 1) / 2);
}

template <int BM, int BN, bool AK, bool BKM, bool LOWER>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel(int64_t k, const double* __restrict__ A, int64_t lda,
                                                          const double* __restrict__ B, int64_t ldb,
                                                          double* __restrict__ C, int64_t ldc, int tiles_n,
                                                          int64_t ntiles) {
    constexpr int WM = BM / 2, WN = BN / 2;     // wave tile
    constexpr int FM = WM / 16, FN = WN / 16;   // MFMA tiles per wave
    using GA = TileGeom<BM, AK>;
    using GB = TileGeom<BN, BKM>;
    __shared__ __attribute__((aligned(16))) double lds[2 * (GA::doubles + GB::doubles)];
    constexpr int STAGE = GA::doubles + GB::doubles;   // stage s: A image at s*STAGE, B image right after it

    // XCD-aware tile order: blocks b, b+8, b+16.. share an XCD (and its L2); give each XCD a contiguous
    // run of the tile list so neighbouring tiles (which share A or B slabs) hit the same L2.
    int64_t bid = blockIdx.x;
    {
        const int64_t q = ntiles / 8, r = ntiles % 8, xcd = bid % 8, loc = bid / 8;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + loc;
    }
    int ti, tj;
    if (LOWER) {
        tri_decode(bid, ti, tj);
    } else {
        ti = (int)(bid / tiles_n);
        tj = (int)(bid % tiles_n);
    }
    const int64_t m0 = (int64_t)ti * BM, n0 = (int64_t)tj * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;

    double4_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j) acc[i][j] = (double4_t){0.0, 0.0, 0.0, 0.0};

    double2 ra[GA::vecs], rb[GB::vecs];
    tile_load<BM, AK>(A, lda, m0, 0, tid, ra);
    tile_load<BN, BKM>(B, ldb, n0, 0, tid, rb);
    tile_store<BM, AK>(lds, tid, ra);
    tile_store<BN, BKM>(lds + GA::doubles, tid, rb);
    __syncthreads();

    // one LDS stage = 4 MFMA k-steps on the wave's FM x FN accumulator tiles
    auto compute_stage = [&](const double* __restrict__ la, const double* __restrict__ lb) {
#pragma unroll
        for (int ks = 0; ks < GEMM_BK / 4; ++ks) {
            double af[FM], bf[FN];
#pragma unroll
            for (int i = 0; i < FM; ++i) af[i] = frag<BM, AK>(la, wr * WM + i * 16 + fr, ks * 4 + fq);
#pragma unroll
            for (int j = 0; j < FN; ++j) bf[j] = frag<BN, BKM>(lb, wc * WN + j * 16 + fr, ks * 4 + fq);
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    };

    const int64_t nk = k / GEMM_BK;
    for (int64_t kt = 0; kt + 1 < nk; ++kt) {
        const int cur = (int)(kt & 1);
        // issue the next stage's HBM/L2 loads; they land while the MFMAs below run
        tile_load<BM, AK>(A, lda, m0, (kt + 1) * GEMM_BK, tid, ra);
        tile_load<BN, BKM>(B, ldb, n0, (kt + 1) * GEMM_BK, tid, rb);
        compute_stage(lds + cur * STAGE, lds + cur * STAGE + GA::doubles);
        tile_store<BM, AK>(lds + (cur ^ 1) * STAGE, tid, ra);
        tile_store<BN, BKM>(lds + (cur ^ 1) * STAGE + GA::doubles, tid, rb);
        __syncthreads();
    }
    {
        const int cur = (int)((nk - 1) & 1);
        compute_stage(lds + cur * STAGE, lds + cur * STAGE + GA::doubles);
    }

    // epilogue: C -= acc.  D[(lane>>4) + 4r][lane&15]
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t row = m0 + wr * WM + i * 16 + fq + 4 * r;
                const int64_t col = n0 + wc * WN + j * 16 + fr;
                double* p = C + row * ldc + col;
                *p = *p - acc[i][j][r];
            }
}

template <int BM, int BN, bool AK, bool BKM, bool LOWER>
static int launch_one(mgpi_ctx* ctx, int64_t m, int64_t n, int64_t k, const double* A, int64_t lda, const double* B,
                      int64_t ldb, double* C, int64_t ldc) {
    const int64_t tm = m / BM, tn = n / BN;
    const int64_t ntiles = LOWER ? tm * (tm + 1) / 2 : tm * tn;
    MGPI_REQUIRE(ntiles > 0 && ntiles < (int64_t)2147483647, "gemm: tile count %lld out of range", (long long)ntiles);
    hipLaunchKernelGGL((gemm_f64_kernel<BM, BN, AK, BKM, LOWER>), dim3((unsigned)ntiles), dim3(256), 0, ctx->stream, k,
                       A, lda, B, ldb, C, ldc, (int)tn, ntiles);
    MGPI_HIP_TRY(hipGetLastError());
    return 0;
}

template <bool AK, bool BKM>
static int launch_shape(mgpi_ctx* ctx, int lower_only, int64_t m, int64_t n, int64_t k, const double* A, int64_t lda,
                        const double* B, int64_t ldb, double* C, int64_t ldc) {
    if (lower_only) {
        MGPI_REQUIRE(m == n && m % 128 == 0, "gemm(lower): needs m == n, multiple of 128 (got %lld x %lld)", (long long)m,
                     (long long)n);
        return launch_one<128, 128, AK, BKM, true>(ctx, m, n, k, A, lda, B, ldb, C, ldc);
    }
    if (m % 128 == 0 && n % 128 == 0) return launch_one<128, 128, AK, BKM, false>(ctx, m, n, k, A, lda, B, ldb, C, ldc);
    if (m % 128 == 0) return launch_one<128, 64, AK, BKM, false>(ctx, m, n, k, A, lda, B, ldb, C, ldc);
    if (n % 128 == 0) return launch_one<64, 128, AK, BKM, false>(ctx, m, n, k, A, lda, B, ldb, C, ldc);
    return launch_one<64, 64, AK, BKM, false>(ctx, m, n, k, A, lda, B, ldb, C, ldc);
}

int mgpi_launch_gemm(mgpi_ctx* ctx, int a_kmajor, int b_kmajor, int lower_only, int64_t m, int64_t n, int64_t k,
                     const double* A, int64_t lda, const double* B, int64_t ldb, double* C, int64_t ldc) {
    if (m == 0 || n == 0 || k == 0) return 0;
    MGPI_REQUIRE(m % 64 == 0 && n % 64 == 0 && k % GEMM_BK == 0,
                 "gemm: m=%lld n=%lld must be multiples of 64 and k=%lld of %d", (long long)m, (long long)n, (long long)k,
                 GEMM_BK);
    MGPI_REQUIRE(((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0) && lda % 2 == 0 && ldb % 2 == 0,
                 "gemm: operands must be 16-byte aligned with even leading dimensions");
    if (!a_kmajor && !b_kmajor) return launch_shape<false, false>(ctx, lower_only, m, n, k, A, lda, B, ldb, C, ldc);
    if (!a_kmajor && b_kmajor) return launch_shape<false, true>(ctx, lower_only, m, n, k, A, lda, B, ldb, C, ldc);
    if (a_kmajor && b_kmajor) return launch_shape<true, true>(ctx, lower_only, m, n, k, A, lda, B, ldb, C, ldc);
    mgpi_set_error("gemm: the (A k-major, B row-major) form is not used by the hot path");
    return MGPI_ERR_ARG;
}

extern "C" int mgpi_gemm_nt_dev(mgpi_ctx* ctx, const mgpi_mat* A, const mgpi_mat* B, mgpi_mat* C, int lower_only) {
    MGPI_REQUIRE(ctx && A && B && C, "mgpi_gemm_nt_dev: null pointer");
    MGPI_REQUIRE(A->cols == B->cols && C->rows == A->rows && C->cols == B->rows,
                 "mgpi_gemm_nt_dev: shapes A %lldx%lld, B %lldx%lld, C %lldx%lld do not conform", (long long)A->rows,
                 (long long)A->cols, (long long)B->rows, (long long)B->cols, (long long)C->rows, (long long)C->cols);
    MGPI_HIP_TRY(hipSetDevice(ctx->device));
    // padded extents: the pads are zero so they contribute nothing
    const int64_t m = C->prows, n = mgpi_roundup(C->cols, 64) <= C->pcols ? (C->pcols) : C->pcols;
    const int64_t k = A->pcols < B->pcols ? A->pcols : B->pcols;
    MGPI_REQUIRE(A->prows >= m && B->prows >= n, "mgpi_gemm_nt_dev: operand padding smaller than the result's");
    mgpi_prof_begin(ctx);
    int st = mgpi_launch_gemm(ctx, 0, 0, lower_only, m, n, k, A->d, A->ld, B->d, B->ld, C->d, C->ld);
    mgpi_prof_end(ctx, 0);
    return st;
}
