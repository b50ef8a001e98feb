// Device-side building blocks of the panel chain (64 x 64 diagonal-block Cholesky, 64-wide triangular sweeps), shared by
// the kernels of chol.hip and by tools/panel_kernel_probe.hip (which times their phases with the shader clock).
// They take the role of LAPACK's dpotf2 / dtrsm inside np.linalg.inv's factorisation (mgpi/interpolators.py:92,179).
#pragma once
#include "mgpi_internal.h"

// The latency-bound kernels of the panel chain share their SIMDs with two waves of the trailing update during the
// look-ahead.  Their own instruction streams are short dependent chains; raising the wave priority lets the SIMD's
// arbiter issue them as soon as they are ready instead of round-robin behind MFMA issues.
#define MGPI_CHAIN_PRIO() __builtin_amdgcn_s_setprio(3)

#define IB MGPI_IB

// ---- 64x64 diagonal block: the pivot rule ---------------------------------------------------------------
// The factorisation (potf2_acc_rounds below) reads only the lower triangle and writes L (lower).
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

// 1/sqrt(d): hardware estimate + two Newton steps
__device__ __forceinline__ double rsq_refined(double d) {
    double y = __builtin_amdgcn_rsq(d);
    y = fma(y * 0.5, fma(-d * y, y, 1.0), y);
    y = fma(y * 0.5, fma(-d * y, y, 1.0), y);
    return y;
}

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
    double zrawn[3][4][16][4]; // ... of the 2nd - 4th row block of a workgroup that holds several (trsm_acc_mxn)
    double Lt[IB][IB];         // Lt[k][m] = L[m][k] for m >= k (the factor, transposed: B operand of the substitution), 0 above
    double invd[IB];           // 1 / L[k][k]
    double dtol[IB];           // pivot thresholds
    // matrix-pipe substitution (trsm_acc_mx): per round R = K / 4 the 4 x 4 diagonal block of L as two A-operand images of
    // v_mfma_f64_4x4x4 - dgA[R][i][k] = (i == k) / L[K+i][K+i],  tA[R][i][k] = (k < i) * -L[K+i][K+k] / L[K+i][K+i]
    double dgA[IB / 4][4][4];
    double tA[IB / 4][4][4];
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

// ---- 64-row substitution  X L^T = A  in accumulator layout, dependent arithmetic on the MATRIX pipe -----------------------
// L comes from the LDS images potf2_acc_rounds left (Lt, invd).  No barrier inside the sweep: every wave works on its own
// 16 rows; per round the four raw entries per row go through the wave's private LDS strip (zraw).
// Why the matrix pipe: beside the trailing update the vector instructions of a panel wave are starved at issue (a dependent v_fma_f64 costs
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
        __asm__ volatile("" ::: "memory");      // the same wave writes single doubles and reads them back: LDS executes a wave's instructions in order, the compiler must keep it too
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

// [r3] NBLK independent 64-row blocks per workgroup, their dependent chains interleaved: the rows launches of the split steps
// (large matrices, beside the trailing update) are latency-bound chains of four dependent matrix instructions per round, so
// further blocks ride along in the same issue slots - 1/NBLK of the workgroups and of the LDS image loads, and less time in
// which those workgroups keep a CU slot from the trailing update.
template <int K, int NBLK>
struct trsm_acc_mxn {
    static __device__ __forceinline__ void run(panel_double4_t (&acc)[NBLK][4], panel_lds& s, int w, int g, int c, double* (&out)[NBLK]) {
        constexpr int QK = K / 16, KL = K % 16, R = K / 4;
        if (c >= KL && c < KL + 4) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                s.zraw[w][g + 4 * r][c - KL] = acc[0][QK][r];
#pragma unroll
                for (int b = 1; b < NBLK; ++b) s.zrawn[b - 1][w][g + 4 * r][c - KL] = acc[b][QK][r];
            }
        }
        __asm__ volatile("" ::: "memory");
        double a[NBLK], y0[NBLK], y[NBLK];
        a[0] = s.zraw[w][c][g];
#pragma unroll
        for (int b = 1; b < NBLK; ++b) a[b] = s.zrawn[b - 1][w][c][g];
        const double dg = s.dgA[R][c & 3][g], t = s.tA[R][c & 3][g];
#pragma unroll
        for (int b = 0; b < NBLK; ++b) y0[b] = PANEL_MFMA4(dg, a[b], 0.0);
#pragma unroll
        for (int b = 0; b < NBLK; ++b) y[b] = PANEL_MFMA4(t, y0[b], y0[b]);
#pragma unroll
        for (int b = 0; b < NBLK; ++b) y[b] = PANEL_MFMA4(t, y[b], y0[b]);
#pragma unroll
        for (int b = 0; b < NBLK; ++b) y[b] = PANEL_MFMA4(t, y[b], y0[b]);
#pragma unroll
        for (int b = 0; b < NBLK; ++b) out[b][K + g] = y[b];
        if constexpr (K + 4 < IB) {
#pragma unroll
            for (int n = 0; n < 4; ++n)
                if (n >= QK) {
                    const double lt = -s.Lt[K + g][16 * n + c];
#pragma unroll
                    for (int b = 0; b < NBLK; ++b) acc[b][n] = PANEL_MFMA(y[b], lt, acc[b][n]);
                }
            __asm__ volatile("" ::: "memory");
            trsm_acc_mxn<K + 4, NBLK>::run(acc, s, w, g, c, out);
        }
    }
};

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
