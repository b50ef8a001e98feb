// Device-side building blocks of the panel chain (64 x 64 diagonal-block Cholesky, 64-wide triangular sweeps), shared by
// the kernels of chol.hip and by tools/panel_kernel_probe.hip (which times their phases with the shader clock).
// They take the role of LAPACK's dpotf2 / dtrsm inside np.linalg.inv's factorisation (mgpi/interpolators.py:92,179).
#pragma once
#include "mgpi_internal.h"

#define IB MGPI_IB

// The latency-bound kernels of the panel chain share their SIMDs with two waves of the trailing update during the
// look-ahead.  Their own instruction streams are short dependent chains; raising the wave priority lets the SIMD's
// arbiter issue them as soon as they are ready instead of round-robin behind MFMA issues.
#define MGPI_CHAIN_PRIO() __builtin_amdgcn_s_setprio(3)

#define IB MGPI_IB

// ---- 64x64 diagonal block: unblocked Cholesky in LDS, one workgroup ------------------------------------
// reads only the lower triangle; writes L (lower) and zeros the strict upper part of the block.
// info: first failing pivot (1-based, global index pivot_base + k + 1), written with atomicCAS-min semantics.
// A pivot fails when it is not above MGPI_PIVOT_TOL * K_kk(original): zero/negative pivots (LAPACK's test) and
// pivots that are pure rounding noise.  The second part is what makes an exactly singular K - duplicated source
// rows under the exact-equality white noise, where the reference's inv raises LinAlgError - fail here too
// instead of returning garbage.
// The threshold is ctx->pivot_tol (default 100 eps; mgpi_ctx_set_pivot_tol / MGPI_PIVOT_TOL; 0 = LAPACK's rule alone).

// broadcast lane `l` (compile-time constant after unrolling) of a double to the whole wave: two v_readlane_b32
__device__ __forceinline__ double readlane_f64(double v, int l) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}

template <int SRC>
__device__ __forceinline__ double potf2_quad_bcast(double v) {
    constexpr int ctrl = SRC | (SRC << 2) | (SRC << 4) | (SRC << 6);   // quad_perm: every lane reads quad lane SRC
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), ctrl, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), ctrl, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// 1/sqrt(d): hardware estimate + two Newton steps
__device__ __forceinline__ double rsq_refined(double d) {
    double y = __builtin_amdgcn_rsq(d);
    y = fma(y * 0.5, fma(-d * y, y, 1.0), y);
    y = fma(y * 0.5, fma(-d * y, y, 1.0), y);
    return y;
}

// One ROUND eliminates FOUR columns K..K+3 (all in lane part K/16) with ONE barrier: the owning parts drop their four
// entries per row into LDS, every lane then
//   * reads the 4 x 4 pivot block and factors it (the same arithmetic in every lane: l_ts, 1/l_tt),
//   * forward-substitutes its own row's four entries against that block -> the row's part of L (x_r[0..3]),
//   * does the same for the 16 rows that face its own 16 columns (x_j[0..3], from the raw entries in LDS) - redundant
//     arithmetic instead of a second LDS round trip and barrier - and applies the rank-4 update a_r[j] -= sum_t x_r[t] x_j[t].
// The unblocked version (round 1) paid one barrier + one LDS round trip per COLUMN: 64 x ~900 cycles = 25 us per
// 64 x 64 block where the dependent arithmetic (one reciprocal square root per pivot) needs ~250.  The pivots d_t are
// the same quantities as in the unblocked algorithm (up to rounding), so the acceptance rule d_t > tol K_tt is unchanged.
template <int K>
struct potf2_rounds {
    static __device__ __forceinline__ void run(double (&a)[16], double (*cb)[4][IB], const double* dtol, int r, int q, int& bad) {
        constexpr int QK = K / 16, KL = K % 16, B = (K / 4) & 1;
        if (q == QK) {
#pragma unroll
            for (int t = 0; t < 4; ++t) cb[B][t][r] = (r >= K) ? a[KL + t] : 0.0;      // rows above the block contribute nothing
        }
        __syncthreads();
        // pivot block (lower part): p_it = entry (K + i, K + t)
        const double p00 = cb[B][0][K], p10 = cb[B][0][K + 1], p20 = cb[B][0][K + 2], p30 = cb[B][0][K + 3];
        const double p11 = cb[B][1][K + 1], p21 = cb[B][1][K + 2], p31 = cb[B][1][K + 3];
        const double p22 = cb[B][2][K + 2], p32 = cb[B][2][K + 3], p33 = cb[B][3][K + 3];
        double c[4][16];                                             // raw entries (row 16 q + j, column K + t)
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int j = 0; j < 16; j += 2) {
                const double2 v = *reinterpret_cast<const double2*>(&cb[B][t][16 * q + j]);
                c[t][j] = v.x;
                c[t][j + 1] = v.y;
            }
        const double a0 = potf2_quad_bcast<QK>(a[KL]), a1 = potf2_quad_bcast<QK>(a[KL + 1]);
        const double a2 = potf2_quad_bcast<QK>(a[KL + 2]), a3 = potf2_quad_bcast<QK>(a[KL + 3]);
        // 4 x 4 Cholesky of the pivot block; a failed pivot is replaced by 1 so that nothing downstream turns into NaN
        const bool ok0 = p00 > dtol[K];
        const double d0 = ok0 ? p00 : 1.0, i0 = rsq_refined(d0);
        const double l10 = p10 * i0, l20 = p20 * i0, l30 = p30 * i0;
        const double e1 = fma(-l10, l10, p11);
        const bool ok1 = e1 > dtol[K + 1];
        const double d1 = ok1 ? e1 : 1.0, i1 = rsq_refined(d1);
        const double l21 = fma(-l20, l10, p21) * i1, l31 = fma(-l30, l10, p31) * i1;
        const double e2 = fma(-l21, l21, fma(-l20, l20, p22));
        const bool ok2 = e2 > dtol[K + 2];
        const double d2 = ok2 ? e2 : 1.0, i2 = rsq_refined(d2);
        const double l32 = fma(-l31, l21, fma(-l30, l20, p32)) * i2;
        const double e3 = fma(-l32, l32, fma(-l31, l31, fma(-l30, l30, p33)));
        const bool ok3 = e3 > dtol[K + 3];
        const double d3 = ok3 ? e3 : 1.0, i3 = rsq_refined(d3);
        if (bad == 0) bad = !ok0 ? K + 1 : (!ok1 ? K + 2 : (!ok2 ? K + 3 : (!ok3 ? K + 4 : 0)));
        // this row's entries of columns K..K+3 of L: x_t = (a_t - sum_{s<t} x_s l_ts) / l_tt for rows at or below
        // column K + t's diagonal, 0 above (for the block's own rows this reproduces l_ts and the diagonal d_t / sqrt(d_t))
        const double x0 = (r >= K) ? a0 * i0 : 0.0;
        const double x1 = (r >= K + 1) ? fma(-x0, l10, a1) * i1 : 0.0;
        const double x2 = (r >= K + 2) ? fma(-x1, l21, fma(-x0, l20, a2)) * i2 : 0.0;
        const double x3 = (r >= K + 3) ? fma(-x2, l32, fma(-x1, l31, fma(-x0, l30, a3))) * i3 : 0.0;
        // the same for the rows that face this lane's columns, and the rank-4 update (columns left of the block see zeros)
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const double y0 = c[0][j] * i0;
            const double y1 = fma(-y0, l10, c[1][j]) * i1;
            const double y2 = fma(-y1, l21, fma(-y0, l20, c[2][j])) * i2;
            const double y3 = fma(-y2, l32, fma(-y1, l31, fma(-y0, l30, c[3][j]))) * i3;
            a[j] = fma(-x3, y3, fma(-x2, y2, fma(-x1, y1, fma(-x0, y0, a[j]))));
        }
        // the owning part keeps the factor entries
        if (q == QK) {
            a[KL] = x0;
            a[KL + 1] = x1;
            a[KL + 2] = x2;
            a[KL + 3] = x3;
        }
        (void)d0; (void)d1; (void)d2; (void)d3;
        if constexpr (K + 4 < IB) potf2_rounds<K + 4>::run(a, cb, dtol, r, q, bad);
    }
};

// ---- 64-wide triangular block solves ---------------------------------------------------------------------
// Right-looking (column-sweep) substitution: once x_k is final every later unknown subtracts its coefficient
// times x_k, so the updates of a step are independent FMAs; the division is a multiply by a reciprocal kept on
// the diagonal of the LDS coefficient image (LAPACK's dtrsm does the same).
//
// Parallel shape: FOUR lanes (one DPP quad) share one row/column of unknowns, lane part q owning unknowns
// [16q, 16q+16).  The pivot x_k is broadcast inside the quad with one quad_perm DPP move per 32 bits.
// Variants measured with rocprofv3 at N=32768 (us per launch, multi-rhs solve / panel trsm):
//   v1 one thread per row, coefficients from LDS in batches of 32 ............ 37 / 22  (one wave per SIMD, 69 % of
//      the wave time in s_waitcnt on LDS latency, SQ_WAIT_ANY)
//   v2 (this) four lanes per row, 16 waves per CU ............................. 19 / 20  (LDS-instruction bound: every
//      quad re-reads the coefficient triangle)
//   v3 one thread per row, coefficients by SCALAR loads from the factor in L2 . 39 / 44  (s_load latency exposed 178x)
// fp64 facts behind this (tools/clock_probe.hip): a dependent v_fma_f64 has 32 cycles of latency (4 to issue); a
// division is a ~15-instruction dependent chain, hence the reciprocals.
//
// LDS image C[k][m]: coefficient of x_k in equation m; zero where equation m does not involve x_k (so the FMAs
// need no predicate), 1/c_kk at m == k.

template <int SRC>
__device__ __forceinline__ double quad_bcast(double v) {
    constexpr int ctrl = SRC | (SRC << 2) | (SRC << 4) | (SRC << 6);   // quad_perm: every lane reads quad lane SRC
    const int lo = __builtin_amdgcn_mov_dpp(__double2loint(v), ctrl, 0xf, 0xf, true);
    const int hi = __builtin_amdgcn_mov_dpp(__double2hiint(v), ctrl, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}

// One ROUND solves four unknowns (all in lane part K/16): the owner's four-step substitution runs on its own registers
// with the 4 x 4 diagonal coefficients read at wave-uniform LDS addresses, the four solved values are broadcast inside the
// quad, and every lane applies a rank-4 update to its 16 unknowns.  The coefficient reads do not depend on the solution,
// so the fully unrolled sweep keeps them in flight ahead of the arithmetic; what is left on the dependent chain per round
// is four (fma, multiply) pairs instead of four times (LDS read -> multiply -> broadcast -> fma) as in the
// one-unknown-per-step sweep of round 1 (22 us per 64 columns, 64 x ~340 cycles).
// FWD: unknowns K, K+1, K+2, K+3 in that order (C[k][m] non-zero for m > k); otherwise K+3 .. K (non-zero for m < k).
template <int K, bool FWD>
__device__ __forceinline__ void quad_round(double (&x)[16], const double (*C)[IB], int q) {
    constexpr int QK = K / 16, KL = K % 16;
    constexpr int U0 = FWD ? K : K + 3, U1 = FWD ? K + 1 : K + 2, U2 = FWD ? K + 2 : K + 1, U3 = FWD ? K + 3 : K;
    constexpr int L0 = U0 % 16, L1 = U1 % 16, L2 = U2 % 16, L3 = U3 % 16;
    double c[4][16];
#pragma unroll
    for (int j = 0; j < 16; j += 2) {
        const double2 v0 = *reinterpret_cast<const double2*>(&C[U0][16 * q + j]);
        const double2 v1 = *reinterpret_cast<const double2*>(&C[U1][16 * q + j]);
        const double2 v2 = *reinterpret_cast<const double2*>(&C[U2][16 * q + j]);
        const double2 v3 = *reinterpret_cast<const double2*>(&C[U3][16 * q + j]);
        c[0][j] = v0.x; c[0][j + 1] = v0.y;
        c[1][j] = v1.x; c[1][j + 1] = v1.y;
        c[2][j] = v2.x; c[2][j + 1] = v2.y;
        c[3][j] = v3.x; c[3][j + 1] = v3.y;
    }
    const double y0 = x[L0] * C[U0][U0];
    const double y1 = fma(-y0, C[U0][U1], x[L1]) * C[U1][U1];
    const double y2 = fma(-y1, C[U1][U2], fma(-y0, C[U0][U2], x[L2])) * C[U2][U2];
    const double y3 = fma(-y2, C[U2][U3], fma(-y1, C[U1][U3], fma(-y0, C[U0][U3], x[L3]))) * C[U3][U3];
    const double z0 = quad_bcast<QK>(y0), z1 = quad_bcast<QK>(y1), z2 = quad_bcast<QK>(y2), z3 = quad_bcast<QK>(y3);
#pragma unroll
    for (int j = 0; j < 16; ++j) x[j] = fma(-c[3][j], z3, fma(-c[2][j], z2, fma(-c[1][j], z1, fma(-c[0][j], z0, x[j]))));
    if (q == QK) {                                                  // the owner keeps the solved values
        x[L0] = z0;
        x[L1] = z1;
        x[L2] = z2;
        x[L3] = z3;
    }
    (void)KL;
}

template <int R, bool FWD>          // R = rounds still to do; FWD: blocks 0, 4, .., 60; else 60, 56, .., 0
struct quad_sweep4 {
    static __device__ __forceinline__ void run(double (&x)[16], const double (*C)[IB], int q) {
        quad_round<FWD ? IB - 4 * R : 4 * (R - 1), FWD>(x, C, q);
        if constexpr (R > 1) quad_sweep4<R - 1, FWD>::run(x, C, q);
    }
};

// the sweep of all 64 unknowns
template <int KDUMMY, bool FWD>
struct quad_sweep {
    static __device__ __forceinline__ void run(double (&x)[16], const double (*C)[IB], int q) { quad_sweep4<IB / 4, FWD>::run(x, C, q); }
};

// stage the 64x64 diagonal block as the coefficient image.  TRANSPOSE: C[k][m] = L[m][k] (else C[k][m] = L[k][m]);
// KEEP_M_GT_K selects the triangle that is kept (m > k for the forward sweep, m < k for the backward one)
template <bool TRANSPOSE, bool KEEP_M_GT_K>
__device__ __forceinline__ void stage_coeff(const double* __restrict__ Ljj, int64_t ldl, double (*C)[IB], int tid) {
#pragma unroll
    for (int u = 0; u < IB * IB / 256; ++u) {
        const int idx = tid + 256 * u, r = idx >> 6, cidx = idx & 63;   // L[r][cidx], coalesced
        const double v = Ljj[r * ldl + cidx];
        const int k = TRANSPOSE ? cidx : r, m = TRANSPOSE ? r : cidx;
        const bool keep = KEEP_M_GT_K ? (m > k) : (m < k);
        C[k][m] = (m == k) ? 1.0 / v : (keep ? v : 0.0);
    }
}


// =====================================================================================================================
// Panel primitives in MFMA ACCUMULATOR layout (round 2).
//
// A 64 x 64 block lives in the accumulators of one 256-thread workgroup: wave w owns tile row w (rows 16 w .. 16 w + 15),
// acc[n][r] = X[16 w + g + 4 r][16 n + c] with g = lane >> 4, c = lane & 15 - the D layout of v_mfma_f64_16x16x4_f64,
// so the contraction-64 products that feed the panel (gemm from the previous step's columns) land where the panel
// arithmetic wants them, and the rank-4 updates inside the 64 x 64 factorisation / substitution are ONE MFMA per
// 16 x 16 tile (the instruction's contraction length is exactly 4).
//
// Why: the quad-lane kernels above are bound by LDS bandwidth - every lane re-reads its 16 coefficients per eliminated
// column, 32 KB per wave per four columns - and by one barrier + LDS round trip per column (measured stand-alone with the
// shader clock, tools/panel_kernel_probe.hip: 64 x 64 Cholesky 48 k cycles, 64-row substitution 20 k).  Here a round
// of four columns moves ~4 KB per wave through LDS: the four raw columns go to LDS once, every lane solves ONE row's
// four entries (6 fma) to produce its MFMA operand, and the update itself is matrix-core work.
// =====================================================================================================================
typedef double panel_double4_t __attribute__((ext_vector_type(4)));
#define PANEL_MFMA(a, b, c) __builtin_amdgcn_mfma_f64_16x16x4f64((a), (b), (c), 0, 0, 0)

struct __attribute__((aligned(32))) panel_lds {
    double raw[2][IB][4];      // potf2: raw[buf][row][t] = entry (row, K + t) of the current round, two rounds in flight
    double zraw[4][16][4];     // trsm: the same, private to each wave (its 16 rows)
    double Lt[IB][IB];         // Lt[k][m] = L[m][k] for m >= k (the factor, transposed: B operand of the substitution), 0 above
    double invd[IB];           // 1 / L[k][k]
    double dtol[IB];           // pivot thresholds
    // matrix-pipe substitution (trsm_acc_mx): per round R = K / 4 the 4 x 4 diagonal block of L as two A-operand images of
    // v_mfma_f64_4x4x4 - dgA[R][i][k] = (i == k) / L[K+i][K+i],  tA[R][i][k] = (k < i) * -L[K+i][K+k] / L[K+i][K+i]
    double dgA[IB / 4][4][4];
    double tA[IB / 4][4][4];
    // matrix-pipe factorisation (potf2_acc_mx): a zero region the size of raw, and per wave the round's multipliers, negated
    // reciprocals and pivots as A-operand images / plain vectors; the pivots of all rounds for the check at the end
    double zeros[2][IB][4];
    double mA[4][4][4];
    double rdA[4][4][4];
    double dsel[4][4];
    double dall[IB];
};

__device__ __forceinline__ panel_double4_t panel_row4(const double (&p)[4]) {
    panel_double4_t v;
    v[0] = p[0]; v[1] = p[1]; v[2] = p[2]; v[3] = p[3];
    return v;
}

// 1/d: hardware estimate (~2^-27) + one Newton step in fma form
__device__ __forceinline__ double rcp_refined(double d) {
    const double p = __builtin_amdgcn_rcp(d);
    return fma(p, fma(-d, p, 1.0), p);
}

// acc[n] -= sum_k Xa[row 16 w + c][k] * Xb[row 16 n + c][k], k = 0..63, for the tile columns lo <= n < hi.
// Operands straight from global memory (they were written by the previous launch and sit in L2) in fragment order:
// lane (c, g) reads the two consecutive k = 8 h + 2 g, + 1 of its row with one 16-byte load and feeds two successive MFMAs.
// arow = &Xa[(16 w + c) * ld + 2 g], bbase = &Xb[c * ld + 2 g].
__device__ __forceinline__ void panel_gemm_nt(panel_double4_t (&acc)[4], const double* __restrict__ arow,
                                              const double* __restrict__ bbase, int64_t ld, int lo, int hi) {
    double2 af[8];
#pragma unroll
    for (int h = 0; h < 8; ++h) af[h] = *reinterpret_cast<const double2*>(arow + 8 * h);
#pragma unroll
    for (int n = 0; n < 4; ++n) {
        if (n >= lo && n < hi) {
            double2 bf[8];
#pragma unroll
            for (int h = 0; h < 8; ++h) bf[h] = *reinterpret_cast<const double2*>(bbase + (int64_t)(16 * n) * ld + 8 * h);
#pragma unroll
            for (int h = 0; h < 8; ++h) {
                acc[n] = PANEL_MFMA(-af[h].x, bf[h].x, acc[n]);
                acc[n] = PANEL_MFMA(-af[h].y, bf[h].y, acc[n]);
            }
        }
    }
}

// ---- 64 x 64 Cholesky in accumulator layout ---------------------------------------------------------------------------
// Round K eliminates columns K..K+3 (tile column QK = K / 16) with ONE barrier:
//   1. the lanes that hold those columns (c in [KL, KL + 4) of tile column QK, waves w >= QK) write them to raw[][row][t];
//   2. every lane factors the 4 x 4 pivot block in LDL^T form - on the dependent chain only a reciprocal (estimate + one
//      Newton step) per pivot; the reciprocal square roots that scale the STORED entries run beside it;
//   3. lane (c, g) solves row 16 w + c against the block:  w_0 = a_0, w_t = a_t - sum_{s<t} w_s m_ts  (m = multipliers of the
//      block), keeps component g:  -w_g is its A operand, and for every tile column n <= w the same solve of row 16 n + c
//      gives the B operand w_g / d_g;  acc[n] -= W V^T is one MFMA per tile;
//   4. the same lane stores the finished entry L[row][K + g] = w_g / sqrt(d_g): to the transposed LDS image Lt (what the
//      substitution that follows reads), its reciprocal diagonal, and (S != nullptr) the row-major 64-wide global strip.
// Rows above the block (row < K + t) produce zeros everywhere.  The pivots d_t are the unblocked algorithm's pivots up to
// rounding, so the acceptance rule d_t > tol * K_tt(original) is unchanged.
template <int K>
struct potf2_acc_rounds {
    static __device__ __forceinline__ void run(panel_double4_t (&acc)[4], panel_lds& s, int w, int g, int c, int& bad,
                                               double* __restrict__ S) {
        constexpr int QK = K / 16, KL = K % 16, B = (K / 4) & 1;
#ifdef PANEL_STAMPS
        long long st0 = 0, st1 = 0, st2 = 0, st3 = 0;
        if (K == 32) st0 = clock64();
#endif
        if (w >= QK && c >= KL && c < KL + 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r) s.raw[B][16 * w + g + 4 * r][c - KL] = acc[QK][r];
        }
        __syncthreads();
#ifdef PANEL_STAMPS
        if (K == 32) st1 = clock64();
#endif
        const double p00 = s.raw[B][K][0], p10 = s.raw[B][K + 1][0], p20 = s.raw[B][K + 2][0], p30 = s.raw[B][K + 3][0];
        const double p11 = s.raw[B][K + 1][1], p21 = s.raw[B][K + 2][1], p31 = s.raw[B][K + 3][1];
        const double p22 = s.raw[B][K + 2][2], p32 = s.raw[B][K + 3][2], p33 = s.raw[B][K + 3][3];
        const int row = 16 * w + c;
        const panel_double4_t ar = panel_row4(s.raw[B][row]);
        // pivot block, LDL^T
        // (a failed pivot is reported, not repaired: what follows it is undefined - NaN / inf - as the header says; keeping
        // the compare-and-select off the dependent chain is worth ~8 dependent instructions per round)
        const double d0 = p00, rd0 = rcp_refined(d0);
        const bool ok0 = d0 > s.dtol[K];
        const double m10 = p10 * rd0, m20 = p20 * rd0, m30 = p30 * rd0;
        const double d1 = fma(-m10, p10, p11), rd1 = rcp_refined(d1);
        const bool ok1 = d1 > s.dtol[K + 1];
        const double q21 = fma(-m20, p10, p21), q31 = fma(-m30, p10, p31);
        const double m21 = q21 * rd1, m31 = q31 * rd1;
        const double d2 = fma(-m21, q21, fma(-m20, p20, p22)), rd2 = rcp_refined(d2);
        const bool ok2 = d2 > s.dtol[K + 2];
        const double q32 = fma(-m31, q21, fma(-m30, p20, p32));
        const double m32 = q32 * rd2;
        const double d3 = fma(-m32, q32, fma(-m31, q31, fma(-m30, p30, p33))), rd3 = rcp_refined(d3);
        const bool ok3 = d3 > s.dtol[K + 3];
#ifdef PANEL_STAMPS
        if (K == 32) { asm volatile("" :: "v"(rd3)); st2 = clock64(); }
#endif
        if (bad == 0) bad = !ok0 ? K + 1 : (!ok1 ? K + 2 : (!ok2 ? K + 3 : (!ok3 ? K + 4 : 0)));
#ifdef PANEL_DEBUG
        if (K == 40 && threadIdx.x == 0) {
            panel_debug[0] = d0; panel_debug[1] = d1; panel_debug[2] = d2; panel_debug[3] = d3;
            panel_debug[4] = s.dtol[K]; panel_debug[5] = s.dtol[K + 1]; panel_debug[6] = s.dtol[K + 2]; panel_debug[7] = s.dtol[K + 3];
            panel_debug[8] = ok0; panel_debug[9] = ok1; panel_debug[10] = ok2; panel_debug[11] = ok3; panel_debug[12] = bad;
        }
#endif
        // component g of the solved row (w_g), its scale for the stored factor (1/sqrt d_g) and for the B operand (1/d_g)
        const double rdg = (g == 0) ? rd0 : (g == 1) ? rd1 : (g == 2) ? rd2 : rd3;
        const double dg = (g == 0) ? d0 : (g == 1) ? d1 : (g == 2) ? d2 : d3;
        const double ig = rsq_refined(dg);
        auto solve = [&](const panel_double4_t& a, int rw) -> double {
            const double w0 = a[0];
            double w1 = fma(-w0, m10, a[1]);
            double w2 = fma(-w1, m21, fma(-w0, m20, a[2]));
            double w3 = fma(-w2, m32, fma(-w1, m31, fma(-w0, m30, a[3])));
            // all four components in every lane, THEN the pick: left alone the compiler sinks each fma into a branch on g -
            // every wave holds all four g, so the branches save nothing and cost compares, EXEC-masked regions and copies
            __asm__ volatile("" : "+v"(w1), "+v"(w2), "+v"(w3));
            const double wg = (g == 0) ? w0 : (g == 1) ? w1 : (g == 2) ? w2 : w3;
            return (rw >= K + g) ? wg : 0.0;                         // rows above column K + g's diagonal: nothing
        };
        const double wa = solve(ar, row);
#ifdef PANEL_STAMPS
        long long st2a = 0, st2b = 0;
        if (K == 32) { asm volatile("" :: "v"(wa)); st2a = clock64(); }
#endif
        // The update itself is UNCONDITIONAL for the tile columns n >= QK (compile time): where it does not apply - tile
        // columns right of the wave's own (upper triangle), waves above the block - the B operand is zero (resp. the rows are
        // finished and their accumulators dead).  With the matrix instruction inside the run-time conditions every merge point
        // copied the accumulators it had touched: ~40 v_mov_b64 per round, vector instructions a starved issue stage pays for.
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            if (n >= QK) {
                double vb = 0.0;
                if (w >= QK && n <= w) {
                    if (n == w) {
                        vb = wa * rdg;
                    } else {
                        const panel_double4_t br = panel_row4(s.raw[B][16 * n + c]);
                        vb = solve(br, 16 * n + c) * rdg;
                    }
                }
                acc[n] = PANEL_MFMA(-wa, vb, acc[n]);
            }
        }
#ifdef PANEL_STAMPS
        if (K == 32) { asm volatile("" :: "v"(acc[2][0]), "v"(acc[3][0])); st2b = clock64(); }
#endif
        // finished entries of columns K..K+3 (zeros above the diagonal, also for the waves above the block)
        const double x = (w >= QK) ? wa * ig : 0.0;
        if (w >= QK) s.Lt[K + g][row] = x;
        if (row == K + g) s.invd[K + g] = ig;
        if (S != nullptr) S[(int64_t)row * IB + K + g] = x;
#ifdef PANEL_STAMPS
        if (K == 32) {
            asm volatile("" :: "v"(acc[2][0]), "v"(acc[3][0]));
            st3 = clock64();
            if (threadIdx.x == 192) { panel_stamps[0] = st1 - st0; panel_stamps[1] = st2 - st1; panel_stamps[2] = st3 - st2;
                                      panel_stamps[3] = st2a - st2; panel_stamps[4] = st2b - st2a; panel_stamps[5] = st3 - st2b; }
        }
#endif
        if constexpr (K + 4 < IB) potf2_acc_rounds<K + 4>::run(acc, s, w, g, c, bad, S);
    }
};

// ---- 64-row substitution  X L^T = A  in accumulator layout --------------------------------------------------------------
// L comes from the LDS images potf2_acc_rounds left (Lt, invd).  No barrier: every wave works on its own 16 rows.
// Round K: the four raw entries per row go through the wave's private LDS strip, lane (c, g) solves row 16 w + c's four
// unknowns against the 4 x 4 diagonal coefficients (wave-uniform reads), keeps y_g: stores it as the final X[row][K + g]
// and supplies -y_g as the A operand; the B operand of tile column n is Lt[K + g][16 n + c] = L[16 n + c][K + g].
template <int K>
struct trsm_acc_rounds {
    static __device__ __forceinline__ void run(panel_double4_t (&acc)[4], panel_lds& s, int w, int g, int c,
                                               double* __restrict__ outrow /* &X[(row of 16 w + c)][0] */) {
        constexpr int QK = K / 16, KL = K % 16;
        if (c >= KL && c < KL + 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r) s.zraw[w][g + 4 * r][c - KL] = acc[QK][r];
        }
        // The strip is written and read by the same wave with no barrier in between: LDS executes one wave's
        // instructions in order, the compiler must not move the loads above the stores either (it did: lanes write
        // single doubles, the row is read back as a whole)
        __asm__ volatile("" ::: "memory");
        const panel_double4_t a = panel_row4(s.zraw[w][c]);
        const double l10 = s.Lt[K][K + 1], l20 = s.Lt[K][K + 2], l30 = s.Lt[K][K + 3];
        const double l21 = s.Lt[K + 1][K + 2], l31 = s.Lt[K + 1][K + 3], l32 = s.Lt[K + 2][K + 3];
        const double y0 = a[0] * s.invd[K];
        const double y1 = fma(-y0, l10, a[1]) * s.invd[K + 1];
        const double y2 = fma(-y1, l21, fma(-y0, l20, a[2])) * s.invd[K + 2];
        const double y3 = fma(-y2, l32, fma(-y1, l31, fma(-y0, l30, a[3]))) * s.invd[K + 3];
        const double yg = (g == 0) ? y0 : (g == 1) ? y1 : (g == 2) ? y2 : y3;
        outrow[K + g] = yg;
#ifdef PANEL_DEBUG
        if (K == 4) {
            double* dbg = panel_debug2 + threadIdx.x * 8;
            dbg[0] = a[0]; dbg[1] = a[1]; dbg[2] = a[2]; dbg[3] = a[3]; dbg[4] = yg; dbg[5] = s.invd[K]; dbg[6] = l10; dbg[7] = acc[0][0];
        }
#endif
        if constexpr (K + 4 < IB) {
#pragma unroll
            for (int n = 0; n < 4; ++n)
                if (n >= QK) acc[n] = PANEL_MFMA(-yg, s.Lt[K + g][16 * n + c], acc[n]);
            trsm_acc_rounds<K + 4>::run(acc, s, w, g, c, outrow);
        }
    }
};

// ---- the same substitution with its dependent arithmetic on the MATRIX pipe ------------------------------------------------
// Why: beside the trailing update the vector instructions of a panel wave are starved at issue (a dependent v_fma_f64 costs
// 430 - 490 cycles there instead of 40, tools/dp_contention_probe.hip), its matrix instructions are not (a dependent
// v_mfma_f64_4x4x4 costs 77 - 80 instead of 44 - 48).  The 4 x 4 forward substitution of a round is therefore written as
// matrix products.  v_mfma_f64_4x4x4 multiplies, for every group of four rows (c >> 2), A (4 x 4) by B (4 x 4) with
//     A[i][k] on lane (c = 4 blk + i, g = k),   B[k][j] and D[i][j] on lane (c = 4 blk + j, g = k resp. i)
// (tools/mfma4_layout_probe.hip), so with lane (c, g) holding a_g(row c) - the round's raw entry of row c in column K + g -
// as B, a UNIFORM 4 x 4 matrix T as A gives D(c, g) = sum_k T[g][k] a_k(row c): a combination across the four columns of the
// same row.  Forward substitution  y = Dg (a - Ls y)  (Dg = 1 / diag, Ls = strictly lower part of the 4 x 4 block) is the
// fixed point of  y <- y0 + T y  with  y0 = Dg a,  T = -Dg Ls, reached exactly after three steps (T is nilpotent): four
// dependent matrix instructions per round instead of seven dependent vector instructions, the same arithmetic as the
// substitution with pre-divided multipliers (no explicit inverse anywhere).  The result lands on lane (c, g) as y_g(row c) -
// what the store and the A operand of the rank-4 update want, no select.
#define PANEL_MFMA4(a, b, c) __builtin_amdgcn_mfma_f64_4x4x4f64((a), (b), (c), 0, 0, 0)

// the two operand images of every round from the factor's LDS images (Lt, invd); all 256 threads, then a barrier
__device__ __forceinline__ void panel_trsm_images(panel_lds& s, int tid) {
    for (int e = tid; e < (IB / 4) * 16; e += 256) {
        const int R = e >> 4, i = (e >> 2) & 3, k = e & 3, K = 4 * R;
        const double inv = s.invd[K + i];
        s.dgA[R][i][k] = (i == k) ? inv : 0.0;
        s.tA[R][i][k] = (k < i) ? -s.Lt[K + k][K + i] * inv : 0.0;
    }
    __syncthreads();
}

template <int K>
struct trsm_acc_mx {
    static __device__ __forceinline__ void run(panel_double4_t (&acc)[4], panel_lds& s, int w, int g, int c,
                                               double* __restrict__ outrow /* &X[(row of 16 w + c)][0] */) {
        constexpr int QK = K / 16, KL = K % 16, R = K / 4;
        if (c >= KL && c < KL + 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r) s.zraw[w][g + 4 * r][c - KL] = acc[QK][r];
        }
        __asm__ volatile("" ::: "memory");      // same wave writes and reads the strip: keep the order (see trsm_acc_rounds)
        const double a = s.zraw[w][c][g];
        const double dg = s.dgA[R][c & 3][g], t = s.tA[R][c & 3][g];
        const double y0 = PANEL_MFMA4(dg, a, 0.0);
        double y = PANEL_MFMA4(t, y0, y0);
        y = PANEL_MFMA4(t, y, y0);
        y = PANEL_MFMA4(t, y, y0);
        outrow[K + g] = y;
        if constexpr (K + 4 < IB) {
#pragma unroll
            for (int n = 0; n < 4; ++n)
                if (n >= QK) acc[n] = PANEL_MFMA(y, -s.Lt[K + g][16 * n + c], acc[n]);
            __asm__ volatile("" ::: "memory");  // the next round's strip writes stay behind this round's read
            trsm_acc_mx<K + 4>::run(acc, s, w, g, c, outrow);
        }
    }
};

// ---- 64 x 64 Cholesky with the pivot recurrence and the row solves on the MATRIX pipe ---------------------------------------
// Same rounds, same LDS exchange and barrier, same rank-4 update as potf2_acc_rounds; what moves is every dependent vector
// instruction of a round except four reciprocal seeds.  The recurrence of the 4 x 4 pivot block is arithmetic on wave-uniform
// scalars; a scalar s is carried as the matrix s I_4 of every four-row group - value on the lanes with (c & 3) == g, zero on
// the others, the same lanes in the A, B and D layouts of v_mfma_f64_4x4x4 - so a product / fused multiply-add of two such
// scalars is ONE matrix instruction.  Signs are arranged so that nothing is ever negated: the reciprocal is carried as
// nrd = -1 / d (seed v_rcp_f64 of -d on the diagonal lanes only, one Newton step  nx <- nx + nx (1 + d nx)), the multipliers as
// nm = p * nrd = -m, which is what the recurrence  d' = p' + nm p  and the row solves want.
//   * pivot-block entries arrive as s I_4 straight from LDS: the diagonal lanes read the entry, the others read a zero region
//     at the same offsets (per-lane base pointer, compile-time offsets);
//   * lane 0 of every wave drops the six multipliers, the four reciprocals and the four pivots into small per-wave LDS images
//     in A-operand layout; every lane then reads ITS entry of T = -Ms (strictly lower multipliers) and of diag(nrd);
//   * row solve  M w = a  as the fixed point of  w <- a + T w  (exact after three steps), own rows and the rows that face the
//     wave's tile columns alike; B operand of the update  nvb = diag(nrd) w, so  acc += w nvb^T  (no negation);
//   * rows above a column's diagonal produce garbage instead of zeros in w - it only ever reaches accumulator entries that are
//     dead (finished rows / finished columns); the STORED entries are masked as before;
//   * the pivots are checked against their thresholds once, after the last round (wave 0, one ballot).
struct panel_mx_state {
    const double* pb;      // this lane's base for pivot-block reads: &raw[0][0][0] on the diagonal lanes, the zero region elsewhere
    double one;            // I_4: 1 on the diagonal lanes
    double ns;             // the seed register: written on the diagonal lanes only, zero elsewhere for ever
    bool diag;
};

// MATRIX_CHAIN = false: the HYBRID - the pivot recurrence stays on the vector pipe (uniform arithmetic in every lane, the same
// sign conventions: nrd = -1 / d, nm = -m), everything per row goes to the matrix pipe.  The recurrence is the shorter chain
// there when the kernel runs alone (18 dependent v_fma_f64 at 40 cycles), and none of the selects, compares and EXEC-masked
// regions the compiler makes of the vector row solves remain.
template <int K, bool MATRIX_CHAIN>
struct potf2_acc_mx {
    static __device__ __forceinline__ void run(panel_double4_t (&acc)[4], panel_lds& s, int w, int g, int c, int& bad,
                                               double* __restrict__ S, panel_mx_state& st) {
        constexpr int QK = K / 16, KL = K % 16, B = (K / 4) & 1;
        if (w >= QK && c >= KL && c < KL + 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r) s.raw[B][16 * w + g + 4 * r][c - KL] = acc[QK][r];
        }
        __syncthreads();
        const int row = 16 * w + c;
        const double a = s.raw[B][row][g];                       // own row, component g
        double d0, d1, d2, d3, nrd0, nrd1, nrd2, nrd3, nm10, nm20, nm30, nm21, nm31, nm32;
        if constexpr (MATRIX_CHAIN) {
            // pivot block as s I_4 (lower part), raw[B][K + i][j] at offset ((B * IB + K + i) * 4 + j)
            const double* pb = st.pb + (B * IB + K) * 4;
            const double P00 = pb[0], P10 = pb[4], P11 = pb[5], P20 = pb[8], P21 = pb[9], P22 = pb[10];
            const double P30 = pb[12], P31 = pb[13], P32 = pb[14], P33 = pb[15];
            auto nrecip = [&](double d) -> double {             // -1 / d on the diagonal lanes (0 elsewhere)
                double ns = st.ns;
                if (st.diag) ns = __builtin_amdgcn_rcp(-d);
                const double e = PANEL_MFMA4(d, ns, st.one);     // 1 + d ns
                return PANEL_MFMA4(e, ns, ns);                   // ns + ns e
            };
            d0 = P00;
            nrd0 = nrecip(d0);
            nm10 = PANEL_MFMA4(P10, nrd0, 0.0); nm20 = PANEL_MFMA4(P20, nrd0, 0.0); nm30 = PANEL_MFMA4(P30, nrd0, 0.0);
            d1 = PANEL_MFMA4(nm10, P10, P11);
            const double q21 = PANEL_MFMA4(nm20, P10, P21), q31 = PANEL_MFMA4(nm30, P10, P31);
            const double u22 = PANEL_MFMA4(nm20, P20, P22), u32 = PANEL_MFMA4(nm30, P20, P32), u33 = PANEL_MFMA4(nm30, P30, P33);
            nrd1 = nrecip(d1);
            nm21 = PANEL_MFMA4(q21, nrd1, 0.0); nm31 = PANEL_MFMA4(q31, nrd1, 0.0);
            d2 = PANEL_MFMA4(nm21, q21, u22);
            const double q32 = PANEL_MFMA4(nm31, q21, u32);
            const double v33 = PANEL_MFMA4(nm31, q31, u33);
            nrd2 = nrecip(d2);
            nm32 = PANEL_MFMA4(q32, nrd2, 0.0);
            d3 = PANEL_MFMA4(nm32, q32, v33);
            nrd3 = nrecip(d3);
        } else {
            const double p00 = s.raw[B][K][0], p10 = s.raw[B][K + 1][0], p20 = s.raw[B][K + 2][0], p30 = s.raw[B][K + 3][0];
            const double p11 = s.raw[B][K + 1][1], p21 = s.raw[B][K + 2][1], p31 = s.raw[B][K + 3][1];
            const double p22 = s.raw[B][K + 2][2], p32 = s.raw[B][K + 3][2], p33 = s.raw[B][K + 3][3];
            auto nrecip = [](double d) -> double {              // -1 / d: estimate of 1 / (-d) + one Newton step
                const double p = __builtin_amdgcn_rcp(-d);
                return fma(p, fma(d, p, 1.0), p);
            };
            d0 = p00;
            nrd0 = nrecip(d0);
            nm10 = p10 * nrd0; nm20 = p20 * nrd0; nm30 = p30 * nrd0;
            d1 = fma(nm10, p10, p11);
            nrd1 = nrecip(d1);
            const double q21 = fma(nm20, p10, p21), q31 = fma(nm30, p10, p31);
            nm21 = q21 * nrd1; nm31 = q31 * nrd1;
            d2 = fma(nm21, q21, fma(nm20, p20, p22));
            nrd2 = nrecip(d2);
            const double q32 = fma(nm31, q21, fma(nm30, p20, p32));
            nm32 = q32 * nrd2;
            d3 = fma(nm32, q32, fma(nm31, q31, fma(nm30, p30, p33)));
            nrd3 = nrecip(d3);
        }
        // the images: lane 0 of the wave holds every one of these scalars (it is a diagonal lane)
        if ((threadIdx.x & 63) == 0) {
            s.mA[w][1][0] = nm10; s.mA[w][2][0] = nm20; s.mA[w][2][1] = nm21;
            s.mA[w][3][0] = nm30; s.mA[w][3][1] = nm31; s.mA[w][3][2] = nm32;
            s.rdA[w][0][0] = nrd0; s.rdA[w][1][1] = nrd1; s.rdA[w][2][2] = nrd2; s.rdA[w][3][3] = nrd3;
            s.dsel[w][0] = d0; s.dsel[w][1] = d1; s.dsel[w][2] = d2; s.dsel[w][3] = d3;
            if (w == 0) { s.dall[K] = d0; s.dall[K + 1] = d1; s.dall[K + 2] = d2; s.dall[K + 3] = d3; }
        }
        __asm__ volatile("" ::: "memory");                       // the wave reads back what its lane 0 has just written (LDS is in order)
        const double tA = s.mA[w][c & 3][g], rA = s.rdA[w][c & 3][g];
        const double dg = s.dsel[w][g];
        auto solve = [&](double b) -> double {                   // M w = b for the four columns of one row, component g on lane (c, g)
            double v = PANEL_MFMA4(tA, b, b);
            v = PANEL_MFMA4(tA, v, b);
            return PANEL_MFMA4(tA, v, b);
        };
        const double wa = solve(a);
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            if (n >= QK) {                                       // unconditional matrix instruction, zero B where it does not apply
                double nvb = 0.0;
                if (w >= QK && n <= w) {
                    const double wn = (n == w) ? wa : solve(s.raw[B][16 * n + c][g]);
                    nvb = PANEL_MFMA4(rA, wn, 0.0);                                    // -w / d, component g
                }
                acc[n] = PANEL_MFMA(wa, nvb, acc[n]);
            }
        }
        // finished entries of columns K..K+3: zeros above the diagonal and for the waves above the block
        const double ig = rsq_refined(dg);
        const double x = (w >= QK && row >= K + g) ? wa * ig : 0.0;
        if (w >= QK) s.Lt[K + g][row] = x;
        if (row == K + g) s.invd[K + g] = ig;
        if (S != nullptr) S[(int64_t)row * IB + K + g] = x;
        if constexpr (K + 4 < IB) {
            potf2_acc_mx<K + 4, MATRIX_CHAIN>::run(acc, s, w, g, c, bad, S, st);
        } else {
            if (w == 0) {                                        // the pivots against their thresholds, once
                const int l = threadIdx.x & 63;
                const bool fail = !(s.dall[l] > s.dtol[l]);
                const unsigned long long m = __ballot(fail);
                bad = m ? (int)__ffsll((long long)m) : 0;
            }
        }
    }
};

// zero region, images and per-lane state of the matrix-pipe factorisation; all 256 threads, before the first round
__device__ __forceinline__ panel_mx_state panel_mx_begin(panel_lds& s, int tid, int g, int c) {
    for (int e = tid; e < 2 * IB * 4; e += 256) (&s.zeros[0][0][0])[e] = 0.0;
    for (int e = tid; e < 4 * 16; e += 256) { (&s.mA[0][0][0])[e] = 0.0; (&s.rdA[0][0][0])[e] = 0.0; }
    panel_mx_state st;
    st.diag = (c & 3) == g;
    st.pb = st.diag ? &s.raw[0][0][0] : &s.zeros[0][0][0];
    st.one = st.diag ? 1.0 : 0.0;
    st.ns = 0.0;
    return st;
}

// load / store one wave's tile row of a 64 x 64 block: acc[n][r] <-> X[(16 w + g + 4 r) * ld + 16 n + c]
__device__ __forceinline__ void panel_load(panel_double4_t (&acc)[4], const double* __restrict__ X, int64_t ld, int w, int g, int c,
                                           bool lower_only) {
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * w + g + 4 * r, colx = 16 * n + c;
            const double v = X[(int64_t)row * ld + colx];
            acc[n][r] = (!lower_only || colx <= row) ? v : 0.0;
        }
}
__device__ __forceinline__ void panel_store(const panel_double4_t (&acc)[4], double* __restrict__ X, int64_t ld, int w, int g, int c) {
#pragma unroll
    for (int n = 0; n < 4; ++n)
#pragma unroll
        for (int r = 0; r < 4; ++r) X[(int64_t)(16 * w + g + 4 * r) * ld + 16 * n + c] = acc[n][r];
}
