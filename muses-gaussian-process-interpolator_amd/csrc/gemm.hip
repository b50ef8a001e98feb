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
// LDS images (conflict-free by construction; SQ_LDS_BANK_CONFLICT measured 0 on the syrk kernel):
//   row-major operand ([row][k], k contiguous in HBM): 16 doubles = 128 B per row, no padding, the eight 16-byte
//     chunks of a row XOR-swizzled by (row>>1)&7.  A lane fetches TWO consecutive k of its row with one
//     ds_read_b128 (chunk 4h+g of lane group g = lane>>4) and feeds them to two successive MFMAs, so MFMA s of a
//     pair contracts over k = 8h + 2g + s.  Within each 16-lane group of a ds_read_b128 the 16 chunk addresses
//     fall on 16 distinct 4-bank slots (rows r, r^1 differ in bank half, the XOR separates the row pairs).
//   k-major operand ([k][col], col contiguous in HBM): stride BT+16 doubles; a lane reads single doubles at
//     k = 8h + 2g + s, 16 consecutive columns per lane group.
#include "mgpi_internal.h"
#include <stdlib.h>

typedef double double4_t __attribute__((ext_vector_type(4)));

#define GEMM_BK 16
#define GEMM_SR 16            // LDS row stride of a row-major operand tile (swizzled, unpadded)

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
            *reinterpret_cast<double2*>(lds + row * GEMM_SR + 2 * (k2 ^ ((row >> 1) & 7))) = r[s];
        }
    }
}

// global -> LDS directly (global_load_lds_dwordx4, no staging registers, no ds_write): the LDS destination of
// a wave instruction is wave-uniform base + lane*16, i.e. linear, so the image's swizzle is applied to the
// per-lane SOURCE address instead (cdna_hip_programming.md 5.4 rule 21).  BT = 128 only: one wave instruction
// (1 KiB) then covers 8 whole swizzled rows of a row-major image, or one whole k-row of a k-major image.
typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) const void gbl_void_t;
template <bool KMAJOR>
__device__ __forceinline__ void tile_glds(const double* __restrict__ g, int64_t ld, int64_t t0, int64_t k0, int tid,
                                          double* __restrict__ lds) {
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int idx = tid + 256 * s;
        const double* src;
        double* dst;                                   // wave-uniform: where lane 0 of this wave lands
        if (KMAJOR) {                                  // idx -> k-row idx/64, lane covers columns 2*(idx%64)..+1
            const int kr = idx >> 6, c2 = idx & 63;
            src = g + (k0 + kr) * ld + t0 + 2 * c2;
            dst = lds + (4 * s + wave) * (128 + 16);
        } else {                                       // idx -> row idx/8, physical chunk idx%8 holds logical chunk p^f(row)
            const int row = idx >> 3, p = idx & 7;
            src = g + (t0 + row) * ld + k0 + 2 * (p ^ ((row >> 1) & 7));
            dst = lds + (32 * s + 8 * wave) * GEMM_SR;
        }
        __builtin_amdgcn_global_load_lds((gbl_void_t*)src, (lds_void_t*)dst, 16, 0, 0);
    }
}

// fragments of one half stage (h = 0, 1 -> k in [8h, 8h+8)) for the tile row/column t: f[0] feeds the first MFMA
// of the pair (k = 8h + 2g), f[1] the second (k = 8h + 2g + 1); g = lane >> 4
template <int BT, bool KMAJOR>
__device__ __forceinline__ void frag_pair(const double* __restrict__ lds, int t, int h, int g, double (&f)[2]) {
    if (KMAJOR) {
        f[0] = lds[(8 * h + 2 * g) * (BT + 16) + t];
        f[1] = lds[(8 * h + 2 * g + 1) * (BT + 16) + t];
    } else {
        const double2 v = *reinterpret_cast<const double2*>(lds + t * GEMM_SR + 2 * ((4 * h + g) ^ ((t >> 1) & 7)));
        f[0] = v.x;
        f[1] = v.y;
    }
}

#include "tile_index.h"

// blockIdx.y selects one problem of a strided batch (operand y starts sA / sB / sC doubles after operand y - 1); the
// single-problem launches pass a 1-deep grid and zero strides
template <int BM, int BN, bool AK, bool BKM, bool LOWER, bool GLDS = false>
__global__ __launch_bounds__(256, 2) void gemm_f64_kernel(int64_t k, const double* __restrict__ A, int64_t lda,
                                                          const double* __restrict__ B, int64_t ldb,
                                                          double* __restrict__ C, int64_t ldc, int tiles_m,
                                                          int tiles_n, int64_t ntiles, gemm_stair st, int64_t sA,
                                                          int64_t sB, int64_t sC) {
    A += (int64_t)blockIdx.y * sA;
    B += (int64_t)blockIdx.y * sB;
    C += (int64_t)blockIdx.y * sC;
    // the 64 x 64 form is the panel's gemm: during the look-ahead it shares its SIMDs with two trailing-update waves and
    // sits on the chain the next big update waits for - let the arbiter issue it first (see chol.hip, MGPI_CHAIN_PRIO)
    if constexpr (BM == 64 && BN == 64) __builtin_amdgcn_s_setprio(3);
    constexpr int WM = BM / 2, WN = BN / 2;     // wave tile
    constexpr int FM = WM / 16, FN = WN / 16;   // MFMA tiles per wave
    using GA = TileGeom<BM, AK>;
    using GB = TileGeom<BN, BKM>;
    __shared__ __attribute__((aligned(16))) double lds[2 * (GA::doubles + GB::doubles)];
    constexpr int STAGE = GA::doubles + GB::doubles;   // stage s: A image at s*STAGE, B image right after it

    // XCD-aware tile order: blocks b, b+8, b+16.. share an XCD (and its L2); give each XCD a contiguous
    // run of the tile list so neighbouring tiles (which share A or B slabs) hit the same L2.
    const int64_t bid = gemm_xcd_remap(blockIdx.x, ntiles);
    int ti, tj;
    if (!LOWER && st.on) {   // multi-GPU trailing update: tiles strictly above the global diagonal are not touched
        if (!gemm_stair_decode(st, bid, tiles_m, ti, tj)) return;
    } else {
        tile_decode<LOWER>(bid, tiles_m, tiles_n, ti, tj);
    }
    const int64_t m0 = (int64_t)ti * BM, n0 = (int64_t)tj * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;

    double2 ra[GA::vecs], rb[GB::vecs];
    if constexpr (GLDS) {
        tile_glds<AK>(A, lda, m0, 0, tid, lds);
        tile_glds<BKM>(B, ldb, n0, 0, tid, lds + GA::doubles);
    } else {
        tile_load<BM, AK>(A, lda, m0, 0, tid, ra);
        tile_load<BN, BKM>(B, ldb, n0, 0, tid, rb);
    }

    // The accumulators start from the C tile and the MFMAs add (-A) B: the result C - A B is then stored straight
    // from the accumulators.  (An epilogue that re-read C cost one exposed HBM round trip per tile.)
    // D[(lane>>4) + 4r][lane&15]
    double4_t acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                acc[i][j][r] = st.zero_c ? 0.0 : C[(m0 + wr * WM + i * 16 + fq + 4 * r) * ldc + n0 + wc * WN + j * 16 + fr];
    if constexpr (!GLDS) {
        tile_store<BM, AK>(lds, tid, ra);
        tile_store<BN, BKM>(lds + GA::doubles, tid, rb);
    }
    __syncthreads();

    // one LDS stage = 2 half stages x 2 MFMA k-steps on the wave's FM x FN accumulator tiles
    auto compute_stage = [&](const double* __restrict__ la, const double* __restrict__ lb) {
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            double af[FM][2], bf[FN][2];
#pragma unroll
            for (int i = 0; i < FM; ++i) {
                frag_pair<BM, AK>(la, wr * WM + i * 16 + fr, h, fq, af[i]);
                af[i][0] = -af[i][0];
                af[i][1] = -af[i][1];
            }
#pragma unroll
            for (int j = 0; j < FN; ++j) frag_pair<BN, BKM>(lb, wc * WN + j * 16 + fr, h, fq, bf[j]);
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
                for (int i = 0; i < FM; ++i)
#pragma unroll
                    for (int j = 0; j < FN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(af[i][s2], bf[j][s2], acc[i][j], 0, 0, 0);
        }
    };

    const int64_t nk = k / GEMM_BK;
    for (int64_t kt = 0; kt + 1 < nk; ++kt) {
        const int cur = (int)(kt & 1);
        // issue the next stage's HBM/L2 loads; they land while the MFMAs below run
        if constexpr (GLDS) {
            tile_glds<AK>(A, lda, m0, (kt + 1) * GEMM_BK, tid, lds + (cur ^ 1) * STAGE);
            tile_glds<BKM>(B, ldb, n0, (kt + 1) * GEMM_BK, tid, lds + (cur ^ 1) * STAGE + GA::doubles);
            __builtin_amdgcn_sched_barrier(0);
            compute_stage(lds + cur * STAGE, lds + cur * STAGE + GA::doubles);
        } else {
            tile_load<BM, AK>(A, lda, m0, (kt + 1) * GEMM_BK, tid, ra);
            tile_load<BN, BKM>(B, ldb, n0, (kt + 1) * GEMM_BK, tid, rb);
            compute_stage(lds + cur * STAGE, lds + cur * STAGE + GA::doubles);
            // keep the staging writes (and the vmcnt waits in front of them) behind the stage's MFMAs: the loads
            // then have the whole stage (64 MFMAs = 4096 cycles) to land
            __builtin_amdgcn_sched_barrier(0);
            tile_store<BM, AK>(lds + (cur ^ 1) * STAGE, tid, ra);
            tile_store<BN, BKM>(lds + (cur ^ 1) * STAGE + GA::doubles, tid, rb);
        }
        __syncthreads();
    }
    {
        const int cur = (int)((nk - 1) & 1);
        compute_stage(lds + cur * STAGE, lds + cur * STAGE + GA::doubles);
    }

    // epilogue: the accumulators hold C - A B.  The row pointers are rebuilt from a laundered thread id: kept live from
    // the prologue they would hold 32 VGPRs across the K loop.
    int tid2 = tid;
    asm volatile("" : "+v"(tid2));
    {
        const int lane2 = tid2 & 63, wave2 = tid2 >> 6;
        double* c0 = C + (m0 + (wave2 >> 1) * WM + (lane2 >> 4)) * ldc + n0 + (wave2 & 1) * WN + (lane2 & 15);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) c0[(i * 16 + 4 * r) * ldc + j * 16] = acc[i][j][r];
    }
}

// ---- software-pipelined 128x128 kernel (the shipped trailing-update kernel) --------------------------------------
// Same tile, LDS image and direct-to-LDS staging as gemm_f64_kernel<128,128,..,GLDS>, with the K loop pipelined by hand
// around ONE barrier per stage:
//     region 1:  ds_read half 1 of stage s -> Y          | 32 MFMAs on X (half 0 of stage s, read one region earlier)
//     s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier            stage s+1 has landed (it was issued a whole stage ago) and
//                                                         every wave is done reading stage s
//     region 2:  global_load_lds stage s+2 -> buffer of s | ds_read half 0 of stage s+1 -> X | 32 MFMAs on Y
// so the HBM/L2 loads get a full stage (64 MFMAs = 4096 cycles) to land instead of half of one, every LDS read is
// covered by 32 MFMAs of the same wave (no reliance on the other workgroup of the CU to fill the pipe), and the
// operand negation moves out of the loop: the accumulators start from -C, the MFMAs add A B, the epilogue stores
// -(acc) = C - A B.  The C row pointers are recomputed in the epilogue (laundered thread id) so that they do not
// occupy 32 VGPRs across the loop - that is what makes room for the second fragment set.
struct gemm_yes { static constexpr bool v = true; };
struct gemm_no  { static constexpr bool v = false; };

// ZEROC: the accumulators start from 0 and C is not read, C = -A B (a compile-time switch: as a run-time select in front of
// the 64 accumulator loads it cost the trailing update 4 % - 0.86 -> 0.82 of peak isolated)
// MERGED (lower form only): the two-region tile order of merged_decode (tile_index.h); the workgroups of the first region -
// the next panel's block column - announce their stored tile on st.mg_counter, which the look-ahead stream waits for
// (wait_count_kernel, chol.hip) instead of running that block column as a gemm of its own beside this kernel
template <bool BKM, bool LOWER, bool ZEROC = false, bool MERGED = false>
__global__ __launch_bounds__(256, 2) void gemm_f64_pipe_kernel(int64_t k, const double* __restrict__ A, int64_t lda,
                                                               const double* __restrict__ B, int64_t ldb,
                                                               double* __restrict__ C, int64_t ldc, int tiles_m,
                                                               int tiles_n, int64_t ntiles, gemm_stair st) {
    constexpr int BM = 128, BN = 128, WM = 64, WN = 64, FM = 4, FN = 4;
    using GA = TileGeom<BM, false>;
    using GB = TileGeom<BN, BKM>;
    __shared__ __attribute__((aligned(16))) double lds[2 * (GA::doubles + GB::doubles)];
    constexpr int STAGE = GA::doubles + GB::doubles;

    int ti, tj;
    if constexpr (MERGED && LOWER) {
        merged_decode(blockIdx.x, tiles_m, st.mg_cw, st.mg_ncol, ntiles, ti, tj);
    } else if constexpr (MERGED) {          // the staircase of the multi-GPU driver, next super-panel's block columns first
        if (!gemm_stair_merged_decode(st, blockIdx.x, ntiles, tiles_m, ti, tj)) return;
    } else {
        const int64_t bid = gemm_xcd_remap(blockIdx.x, ntiles);
        if (!LOWER && st.on) {
            if (!gemm_stair_decode(st, bid, tiles_m, ti, tj)) return;
        } else {
            tile_decode<LOWER>(bid, tiles_m, tiles_n, ti, tj);
        }
    }
    const int64_t m0 = (int64_t)ti * BM, n0 = (int64_t)tj * BN;
    const int tid = threadIdx.x;

    tile_glds<false>(A, lda, m0, 0, tid, lds);
    tile_glds<BKM>(B, ldb, n0, 0, tid, lds + GA::doubles);

    double4_t acc[FM][FN];
    {
        const int lane = tid & 63, wave = tid >> 6;
        const double* c0 = C + (m0 + (wave >> 1) * WM + (lane >> 4)) * ldc + n0 + (wave & 1) * WN + (lane & 15);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i][j][r] = ZEROC ? 0.0 : -c0[(i * 16 + 4 * r) * ldc + j * 16];
    }
    const int lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int fr = lane & 15, fq = lane >> 4;
    const int64_t nk = k / GEMM_BK;
    __syncthreads();                                              // stage 0 landed
    if (nk > 1) {
        tile_glds<false>(A, lda, m0, GEMM_BK, tid, lds + STAGE);
        tile_glds<BKM>(B, ldb, n0, GEMM_BK, tid, lds + STAGE + GA::doubles);
    }

    double xa[FM][2], xb[FN][2], ya[FM][2], yb[FN][2];
    auto read_half = [&](const double* __restrict__ la, const double* __restrict__ lb, int h, double (&fa)[FM][2],
                         double (&fb)[FN][2]) {
#pragma unroll
        for (int i = 0; i < FM; ++i) frag_pair<BM, false>(la, wr * WM + i * 16 + fr, h, fq, fa[i]);
#pragma unroll
        for (int j = 0; j < FN; ++j) frag_pair<BN, BKM>(lb, wc * WN + j * 16 + fr, h, fq, fb[j]);
    };
    // MFMAs lo..hi-1 of a half stage (32 in all: 2 k-steps x 4 x 4 accumulator tiles)
    auto mfma_range = [&](const double (&fa)[FM][2], const double (&fb)[FN][2], int lo, int hi) {
#pragma unroll
        for (int t = 0; t < 2 * FM * FN; ++t) {
            const int s2 = t / (FM * FN), i = (t / FN) % FM, j = t % FN;
            if (t >= lo && t < hi)
                acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(fa[i][s2], fb[j][s2], acc[i][j], 0, 0, 0);
        }
    };
    auto body = [&](int64_t kt, auto refill, auto more) {
        const int cur = (int)(kt & 1);
        const double* la = lds + cur * STAGE;
        const double* lb = la + GA::doubles;
        // The sched_barriers pin "first MFMA, then the loads, then the MFMAs that cover them".  Left alone the scheduler
        // sinks the reads below the MFMAs (one fragment set, fewer registers) and their latency is exposed at the
        // barrier; and a read issued BEFORE the first MFMA drags that MFMA's s_waitcnt down to lgkmcnt(0), i.e. the
        // wave would wait for the reads it has just issued instead of only for the fragments it needs.
        mfma_range(xa, xb, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        read_half(la, lb, 1, ya, yb);
        __builtin_amdgcn_sched_barrier(0);
        mfma_range(xa, xb, 1, 32);
        __builtin_amdgcn_sched_barrier(0);
        __syncthreads();
        __builtin_amdgcn_sched_barrier(0);
        mfma_range(ya, yb, 0, 1);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (decltype(refill)::v) {
            tile_glds<false>(A, lda, m0, (kt + 2) * GEMM_BK, tid, lds + cur * STAGE);
            tile_glds<BKM>(B, ldb, n0, (kt + 2) * GEMM_BK, tid, lds + cur * STAGE + GA::doubles);
        }
        if constexpr (decltype(more)::v) read_half(lds + (cur ^ 1) * STAGE, lds + (cur ^ 1) * STAGE + GA::doubles, 0, xa, xb);
        __builtin_amdgcn_sched_barrier(0);
        mfma_range(ya, yb, 1, 32);
        __builtin_amdgcn_sched_barrier(0);
    };

    read_half(lds, lds + GA::doubles, 0, xa, xb);
    int64_t kt = 0;
    for (; kt + 2 < nk; ++kt) body(kt, gemm_yes(), gemm_yes());
    if (nk >= 2) body(nk - 2, gemm_no(), gemm_yes());
    body(nk - 1, gemm_no(), gemm_no());

    // epilogue: acc = -(C - A B).  The addresses are rebuilt from a laundered thread id (see the header comment).
    int tid2 = tid;
    asm volatile("" : "+v"(tid2));
    const int lane2 = tid2 & 63, wave2 = tid2 >> 6;
    double* c0 = C + (m0 + (wave2 >> 1) * WM + (lane2 >> 4)) * ldc + n0 + (wave2 & 1) * WN + (lane2 & 15);
    if constexpr (MERGED) {
        if ((int)blockIdx.x < st.mg_ncol) {       // workgroup-uniform: a tile of the next panel's block column
            // Another stream's kernels read this tile as soon as the counter says so, on any XCD, while this kernel is still
            // running - so the tile must be in memory, not dirty in this XCD's L2.  A release fence would do that by writing
            // back the WHOLE L2 (buffer_wbl2), once per workgroup: measured 4 ms per factorisation at N = 32768.  Instead the
            // tile is stored write-through (agent-scope relaxed atomic stores = global_store sc1) and the count is bumped
            // once every wave's stores have been acknowledged.
#pragma unroll
            for (int i = 0; i < FM; ++i)
#pragma unroll
                for (int j = 0; j < FN; ++j)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        __hip_atomic_store(&c0[(i * 16 + 4 * r) * ldc + j * 16], -acc[i][j][r], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            if (tid2 == 0) __hip_atomic_fetch_add(st.mg_counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            return;
        }
    }
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
#pragma unroll
            for (int r = 0; r < 4; ++r) c0[(i * 16 + 4 * r) * ldc + j * 16] = -acc[i][j][r];
}

// Measured and dropped (round 3): a 256 x 128 tile - 8 waves in a 4 x 2 grid, ONE workgroup per CU, otherwise this kernel
// (same LDS image, same staging, same pipelined K loop).  25 % fewer L2 requests and 13 % less HBM-side traffic as intended,
// but 59.7 TFLOP/s against 69.5 on the lower update at n = 32256, K = 1024 (NN, 32768 x 4096 x 2048: 64.8 against 70.5):
// MFMA-busy 0.77 against 0.89.  With one barrier domain per CU every wave of the CU waits at the same stage boundary, where two
// independent 4-wave workgroups fill each other's barrier, prologue and epilogue.  Counters: profiles/r03_wide_tile_pmc.md;
// the kernel text: tools/gemm_wide_experiment.inc.

static gemm_stair no_stair() {
    gemm_stair st;
    memset(&st, 0, sizeof(st));
    st.P = st.Q = st.tpb = 1;
    return st;
}

template <int BM, int BN, bool AK, bool BKM, bool LOWER, bool GLDS = false>
static int launch_one(mgpi_ctx* ctx, int64_t m, int64_t n, int64_t k, const double* A, int64_t lda, const double* B,
                      int64_t ldb, double* C, int64_t ldc, gemm_stair st = no_stair()) {
    const int64_t tm = m / BM, tn = n / BN;
    const int64_t ntiles = LOWER ? tm * (tm + 1) / 2 : (st.on ? (int64_t)st.prefix[st.nbands] : tm * tn);
    MGPI_REQUIRE(ntiles > 0 && ntiles < (int64_t)2147483647, "gemm: tile count %lld out of range", (long long)ntiles);
    st.zero_c = ctx->gemm_zero_c;
    hipLaunchKernelGGL((gemm_f64_kernel<BM, BN, AK, BKM, LOWER, GLDS>), dim3((unsigned)ntiles), dim3(256), 0, ctx->stream, k,
                       A, lda, B, ldb, C, ldc, (int)tm, (int)tn, ntiles, st, (int64_t)0, (int64_t)0, (int64_t)0);
    MGPI_HIP_TRY(hipGetLastError());
    return 0;
}

// strided batch of  C_y (m x n) -= A_y (m x k, row-major) * B_y (k x n, row-major)  with 64 x 64 tiles: the shape of
// the blocked inversion of the factor's diagonal blocks (chol.hip), where every problem is small and there are N / nb
int mgpi_launch_gemm_nn_batched(mgpi_ctx* ctx, int64_t m, int64_t n, int64_t k, const double* A, int64_t lda, int64_t sA,
                                const double* B, int64_t ldb, int64_t sB, double* C, int64_t ldc, int64_t sC, int batch) {
    if (m == 0 || n == 0 || k == 0 || batch == 0) return 0;
    MGPI_REQUIRE(m % 64 == 0 && n % 64 == 0 && k % GEMM_BK == 0 && batch > 0 && batch < 65536,
                 "batched gemm: m=%lld n=%lld k=%lld batch=%d", (long long)m, (long long)n, (long long)k, batch);
    const int64_t tm = m / 64, tn = n / 64, ntiles = tm * tn;
    hipLaunchKernelGGL((gemm_f64_kernel<64, 64, false, true, false, false>), dim3((unsigned)ntiles, (unsigned)batch), dim3(256), 0,
                       ctx->stream, k, A, lda, B, ldb, C, ldc, (int)tm, (int)tn, ntiles, no_stair(), sA, sB, sC);
    MGPI_HIP_TRY(hipGetLastError());
    return 0;
}

template <bool BKM, bool LOWER>
static int launch_pipe(mgpi_ctx* ctx, int64_t m, int64_t n, int64_t k, const double* A, int64_t lda, const double* B,
                       int64_t ldb, double* C, int64_t ldc, gemm_stair st = no_stair()) {
    const int64_t tm = m / 128, tn = n / 128;
    const int64_t ntiles = LOWER ? tm * (tm + 1) / 2 : (st.on ? (int64_t)st.prefix[st.nbands] : tm * tn);
    MGPI_REQUIRE(ntiles > 0 && ntiles < (int64_t)2147483647, "gemm: tile count %lld out of range", (long long)ntiles);
    if (ctx->gemm_zero_c)
        hipLaunchKernelGGL((gemm_f64_pipe_kernel<BKM, LOWER, true>), dim3((unsigned)ntiles), dim3(256), 0, ctx->stream, k, A, lda, B, ldb,
                           C, ldc, (int)tm, (int)tn, ntiles, st);
    else
        hipLaunchKernelGGL((gemm_f64_pipe_kernel<BKM, LOWER>), dim3((unsigned)ntiles), dim3(256), 0, ctx->stream, k, A, lda, B, ldb,
                           C, ldc, (int)tm, (int)tn, ntiles, st);
    MGPI_HIP_TRY(hipGetLastError());
    return 0;
}

// A22 (t x t, lower tiles) -= P P^T with the first `cw` tile columns first; *counter grows by merged_column_tiles(t/128, cw)
// as those tiles complete.  The shape of the Cholesky trailing update with the next panel's block column folded in.
int mgpi_launch_syrk_merged(mgpi_ctx* ctx, int64_t t, int64_t k, const double* P, int64_t ldp, double* C, int64_t ldc, int cw,
                            unsigned* counter) {
    MGPI_REQUIRE(t % 128 == 0 && k % GEMM_BK == 0 && cw >= 1 && (int64_t)cw * 128 < t && counter != nullptr,
                 "merged syrk: t=%lld k=%lld cw=%d", (long long)t, (long long)k, cw);
    MGPI_REQUIRE(((uintptr_t)P % 16 == 0) && ldp % 2 == 0, "merged syrk: operand must be 16-byte aligned with an even leading dimension");
    const int64_t tm = t / 128, ntiles = tm * (tm + 1) / 2;
    MGPI_REQUIRE(ntiles < (int64_t)2147483647, "merged syrk: tile count %lld out of range", (long long)ntiles);
    gemm_stair st = no_stair();
    st.mg_cw = cw;
    st.mg_ncol = (int)merged_column_tiles((int)tm, cw);
    st.mg_counter = counter;
    hipLaunchKernelGGL((gemm_f64_pipe_kernel<false, true, false, true>), dim3((unsigned)ntiles), dim3(256), 0, ctx->stream, k, P, ldp, P,
                       ldp, C, ldc, (int)tm, (int)tm, ntiles, st);
    MGPI_HIP_TRY(hipGetLastError());
    return 0;
}

// 128 x 128 tiles: the hand-pipelined kernel for row-major A (every shape of the hot path but the backward multi-rhs solve),
// gemm_f64_kernel with direct-to-LDS staging for k-major A.  Measured and dropped (round 1-2, removed in round 3): a ping-pong
// kernel of two 4-wave groups per workgroup (53.5 against 61.3 TFLOP/s at n = 32256), register-staged tiles (-4 %).
template <bool AK, bool BKM>
static int launch_shape(mgpi_ctx* ctx, int lower_only, int64_t m, int64_t n, int64_t k, const double* A, int64_t lda,
                        const double* B, int64_t ldb, double* C, int64_t ldc) {
    if (ctx->lean && !lower_only) return launch_one<64, 64, AK, BKM, false>(ctx, m, n, k, A, lda, B, ldb, C, ldc);
    if (lower_only) {
        MGPI_REQUIRE(m == n && m % 128 == 0, "gemm(lower): needs m == n, multiple of 128 (got %lld x %lld)", (long long)m,
                     (long long)n);
        // developer switch for counter runs: the merged form of the trailing update alone (its counter ticks, nobody waits)
        static const int merged_cw = getenv("MGPI_SYRK_MERGED_CW") ? atoi(getenv("MGPI_SYRK_MERGED_CW")) : 0;
        if constexpr (!AK && !BKM)
            if (merged_cw > 0 && (int64_t)merged_cw * 128 < m && A == B && lda == ldb)
                return mgpi_launch_syrk_merged(ctx, m, k, A, lda, C, ldc, merged_cw, reinterpret_cast<unsigned*>(ctx->d_scal + 8064));
        if constexpr (!AK) return launch_pipe<BKM, true>(ctx, m, n, k, A, lda, B, ldb, C, ldc);
        return launch_one<128, 128, AK, BKM, true, true>(ctx, m, n, k, A, lda, B, ldb, C, ldc);
    }
    // Skinny updates (the next panel's block column: t x 512 with t <= 8192) give fewer 128 x 128 tiles than the chip has
    // workgroup slots (256 CUs x 2): half the CUs idle for the length of one K = 512 tile, on the chain the next panel waits
    // for.  Narrower tiles fill the chip (N = 8192: 240 tiles -> 480; measured 184 -> ... us per column update).
    // [r3] the threshold follows the size of the factorisation the context last enqueued (chol.hip, potrf_enqueue): 448 beside
    // a large trailing update, which fills the idle slots anyway; 1300 up to N = 16384, where one to two and a half rounds of
    // 128 x 128 tiles leave a ragged last round on the chain (C2: 13.03 -> 12.59 ms per step, C3/C4 +0.1..0.3 % if applied there).
    static const int64_t skinny_env = getenv("MGPI_GEMM_SKINNY_TILES") ? atoll(getenv("MGPI_GEMM_SKINNY_TILES")) : -1;
    const int64_t skinny = skinny_env >= 0 ? skinny_env : ctx->skinny_tiles;
    if (m % 128 == 0 && n % 128 == 0 && (m / 128) * (n / 128) < skinny) {
        if ((m / 128) * (n / 64) < skinny) return launch_one<64, 64, AK, BKM, false>(ctx, m, n, k, A, lda, B, ldb, C, ldc);
        return launch_one<128, 64, AK, BKM, false>(ctx, m, n, k, A, lda, B, ldb, C, ldc);
    }
    if (m % 128 == 0 && n % 128 == 0) {
        if constexpr (!AK) return launch_pipe<BKM, false>(ctx, m, n, k, A, lda, B, ldb, C, ldc);
        return launch_one<128, 128, AK, BKM, false, true>(ctx, m, n, k, A, lda, B, ldb, C, ldc);
    }
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
    mgpi_prof_begin(ctx, 0);
    int st = mgpi_launch_gemm(ctx, 0, 0, lower_only, m, n, k, A->d, A->ld, B->d, B->ld, C->d, C->ld);
    mgpi_prof_end(ctx, 0);
    return st;
}

// C (m x n) -= A (m x k) * B (k x n), all row-major: the update form of the multi-rhs triangular solves
extern "C" int mgpi_gemm_nn_dev(mgpi_ctx* ctx, const mgpi_mat* A, const mgpi_mat* B, mgpi_mat* C) {
    MGPI_REQUIRE(ctx && A && B && C, "mgpi_gemm_nn_dev: null pointer");
    MGPI_REQUIRE(A->cols == B->rows && C->rows == A->rows && C->cols == B->cols,
                 "mgpi_gemm_nn_dev: shapes A %lldx%lld, B %lldx%lld, C %lldx%lld do not conform", (long long)A->rows,
                 (long long)A->cols, (long long)B->rows, (long long)B->cols, (long long)C->rows, (long long)C->cols);
    MGPI_HIP_TRY(hipSetDevice(ctx->device));
    const int64_t m = C->prows, n = C->pcols;
    const int64_t k = A->pcols < B->prows ? A->pcols : B->prows;
    MGPI_REQUIRE(A->prows >= m && B->pcols >= n, "mgpi_gemm_nn_dev: operand padding smaller than the result's");
    return mgpi_launch_gemm(ctx, 0, 1, 0, m, n, k, A->d, A->ld, B->d, B->ld, C->d, C->ld);
}

// C = -(A B): the accumulators start from zero and C is not read (the product with an inverse diagonal block in the variance
// sweep of the multi-GPU driver, where C is the buffer that is broadcast next)
extern "C" int mgpi_gemm_nn_neg_dev(mgpi_ctx* ctx, const mgpi_mat* A, const mgpi_mat* B, mgpi_mat* C) {
    MGPI_REQUIRE(ctx != nullptr, "mgpi_gemm_nn_neg_dev: null ctx");
    ctx->gemm_zero_c = 1;
    const int rc = mgpi_gemm_nn_dev(ctx, A, B, C);
    ctx->gemm_zero_c = 0;
    return rc;
}

// the merged form of mgpi_gemm_nt_stair_dev below: the first `first_cols` local block columns of C (the next super-panel's, which
// the look-ahead context factors next) are updated first, stored write-through and counted on ctx's progress counter - the
// look-ahead context waits for the count (mgpi_ctx_wait_count) instead of running those columns as a gemm of its own beside this
// one.  *live_tiles: how far the counter moves (the tiles of those columns on/below the global diagonal).
extern "C" int mgpi_gemm_nt_stair_merged_dev(mgpi_ctx* ctx, const mgpi_mat* A, const mgpi_mat* B, mgpi_mat* C, int P, int p, int Q,
                                             int q, int bi0, int bj0, int nb, int first_cols, int64_t* live_tiles) {
    MGPI_REQUIRE(ctx && A && B && C && live_tiles, "mgpi_gemm_nt_stair_merged_dev: null pointer");
    MGPI_REQUIRE(A->cols == B->cols && C->rows == A->rows && C->cols == B->rows, "mgpi_gemm_nt_stair_merged_dev: shapes do not conform");
    MGPI_REQUIRE(P >= 1 && Q >= 1 && p >= 0 && p < P && q >= 0 && q < Q && nb >= 128 && nb % 128 == 0 && first_cols >= 1,
                 "mgpi_gemm_nt_stair_merged_dev: bad grid description");
    const int64_t m = C->prows, n = C->pcols, k = A->pcols < B->pcols ? A->pcols : B->pcols;
    MGPI_REQUIRE(m % 128 == 0 && n % 128 == 0 && k % GEMM_BK == 0 && (int64_t)first_cols * nb < n,
                 "mgpi_gemm_nt_stair_merged_dev: extents must be tile multiples and the first columns a proper part");
    MGPI_REQUIRE(((uintptr_t)A->d % 16 == 0) && ((uintptr_t)B->d % 16 == 0) && A->ld % 2 == 0 && B->ld % 2 == 0,
                 "mgpi_gemm_nt_stair_merged_dev: operands must be 16-byte aligned with even leading dimensions");
    MGPI_HIP_TRY(hipSetDevice(ctx->device));
    gemm_stair st = no_stair();
    const int tm = (int)(m / 128), tn = (int)(n / 128);
    *live_tiles = 0;
    const int64_t total = gemm_stair_fill_merged(st, tm, tn, P, p, Q, q, bi0, bj0, nb / 128, first_cols * (nb / 128), live_tiles);
    MGPI_REQUIRE(total >= 0, "mgpi_gemm_nt_stair_merged_dev: local trailing matrix has too many tile rows (%d)", tm);
    if (total == 0) return 0;
    st.mg_counter = reinterpret_cast<unsigned*>(ctx->d_scal + 8064);
    mgpi_prof_begin(ctx, 0);
    hipLaunchKernelGGL((gemm_f64_pipe_kernel<false, false, false, true>), dim3((unsigned)total), dim3(256), 0, ctx->stream, k, A->d, A->ld, B->d,
                       B->ld, C->d, C->ld, tm, tn, total, st);
    mgpi_prof_end(ctx, 0);
    MGPI_HIP_TRY(hipGetLastError());
    return 0;
}

// C -= A B^T restricted to the tiles on/below the GLOBAL diagonal of a 2-D block-cyclic matrix: C is this rank's
// local trailing matrix (local block rows bi0.., local block columns bj0.. of an nb-blocked matrix on a P x Q grid,
// this rank at (p, q)), A (m x k) the row panel, B (n x k) the column panel.  One launch per block step.
extern "C" int mgpi_gemm_nt_stair_dev(mgpi_ctx* ctx, const mgpi_mat* A, const mgpi_mat* B, mgpi_mat* C, int P, int p, int Q,
                                      int q, int bi0, int bj0, int nb) {
    MGPI_REQUIRE(ctx && A && B && C, "mgpi_gemm_nt_stair_dev: null pointer");
    MGPI_REQUIRE(A->cols == B->cols && C->rows == A->rows && C->cols == B->rows, "mgpi_gemm_nt_stair_dev: shapes do not conform");
    MGPI_REQUIRE(P >= 1 && Q >= 1 && p >= 0 && p < P && q >= 0 && q < Q && nb >= 128 && nb % 128 == 0,
                 "mgpi_gemm_nt_stair_dev: bad grid description");
    const int64_t m = C->prows, n = C->pcols, k = A->pcols < B->pcols ? A->pcols : B->pcols;
    MGPI_REQUIRE(m % 128 == 0 && n % 128 == 0 && k % GEMM_BK == 0, "mgpi_gemm_nt_stair_dev: extents must be tile multiples");
    MGPI_HIP_TRY(hipSetDevice(ctx->device));
    gemm_stair st = no_stair();
    const int tm = (int)(m / 128), tn = (int)(n / 128);
    const int64_t total = gemm_stair_fill(st, tm, tn, P, p, Q, q, bi0, bj0, nb / 128);
    MGPI_REQUIRE(total >= 0, "mgpi_gemm_nt_stair_dev: local trailing matrix has too many tile rows (%d)", tm);
    if (total == 0) return 0;
    mgpi_prof_begin(ctx, 0);
    int rc = launch_pipe<false, false>(ctx, m, n, k, A->d, A->ld, B->d, B->ld, C->d, C->ld, st);
    mgpi_prof_end(ctx, 0);
    return rc;
}
