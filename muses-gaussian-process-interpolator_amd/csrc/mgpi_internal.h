// Internal declarations shared by the HIP translation units of libmgpi_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include "mgpi_hip.h"

#define MGPI_TILE   128      // padding granule of matrix dims and the gemm macro-tile
#define MGPI_IB      64      // inner (panel) block of the two-level Cholesky / trsm

#include <vector>
// per-slot HIP-event pairs recorded around launches (no host sync while recording); read by get_profile
struct mgpi_prof_slot {
    std::vector<hipEvent_t> begin, end;
    size_t used;
    double ms;          // filled by mgpi_prof_collect
    int64_t launches;
    double flops;       // algorithmic flops of the launches recorded in this slot (set by the launcher)
};

#define MGPI_PROF_SLOTS 7
#define MGPI_PINNED_BYTES (64 * 1024)
struct mgpi_pool;

struct mgpi_ctx {
    int device;
    int nb;                       // outer Cholesky block
    hipStream_t stream;           // where launchers enqueue (the main stream, except during a look-ahead panel)
    hipStream_t stream2;          // high-priority stream of the look-ahead panel factorisation
    hipStream_t stream3;          // the forward sweep of a solve, beside the factorisation (chol.hip, potrf_enqueue)
    hipEvent_t ev_fwd;            // panel columns final -> stream3; stream3 done -> main stream
    hipEvent_t ev_join;           // mgpi_ctx_wait: "everything queued on this ctx so far"
    hipEvent_t ev_upd, ev_panel;  // main-stream update of a step done / look-ahead stream's panel (block) done
    int no_lookahead;             // MGPI_NO_LOOKAHEAD=1 (developer switch)
    int gemm_zero_c;              // the next gemm launches compute C = -A B (accumulators start from 0, C is not read)
    int lean;                     // launch gemms as 64x64 tiles (32 KB LDS, < 104 VGPRs): they can then start next to
                                  // two resident trailing-update workgroups during the look-ahead
    hipEvent_t ev0, ev1;          // tic/toc
    int profile;
    mgpi_prof_slot* prof;         // [MGPI_PROF_SLOTS]: 0 trailing update (lower syrk), 1 panel, 2 whole potrf, 3 next-panel
                                  // column update, 4 big update of the multi-rhs forward solve (variance), 5 fused predict mean,
                                  // 6 backward sweep of mgpi_posv_dev (what the solve still costs after the factorisation)
    int* d_info;                  // device: first failing pivot (0 = none)
    double* d_scal;               // device scratch, 8192 doubles: [0] logdet, [1] status accumulator, [64, 64 + 64*64 + 64) panel images, [8000] barrier token, [8064] progress counter of the merged trailing update, [8072] stream-overlap handshake, [8080] piece counter of the fused backward substitution
    void* h_pinned;               // 64 KB of pinned host memory: small read-backs (status words, log-likelihood values) land here
                                  // without the staging copy a pageable destination costs (MGPI_PINNED_BYTES)
    double* d_diag;               // diag(K) before factoring (pivot threshold), grown on demand
    int64_t diag_len;
    double* work;                 // device workspace, grown on demand
    size_t work_bytes;
    mgpi_pool* pool;              // freed matrix buffers awaiting reuse (core.hip)
    int streams_overlap;          // -1 not asked yet; 1 / 0: kernels of stream and stream2 do / do not run side by side (chol.hip)
    int64_t skinny_tiles;         // plain gemms with fewer 128 x 128 tiles than this take narrower tiles (gemm.hip; set per factorisation)
    double pivot_tol;             // a pivot must exceed pivot_tol * K_kk(original); 0 = LAPACK's "pivot > 0" rule
};

struct mgpi_mat {
    mgpi_ctx* ctx;
    int64_t rows, cols;           // logical shape
    int64_t prows, pcols;         // padded (allocated) shape; pcols == ld for owners
    int64_t ld;                   // row stride in doubles
    double* d;                    // device pointer to element (0,0)
    int owns;
    int factored;                 // set by potrf
    double diag_min, diag_max;    // smallest / largest L_ii of the factor as the last status read saw them (0: unknown) - chol.hip, trsv_sweeps
    size_t alloc_bytes;           // owners: size of the device block (>= prows*pcols*8 when it came from the pool)
    double* dinv;                 // -(L_bb)^-1 of the factor's nb x nb diagonal blocks, block b at b nb (nb + 1), row stride nb = dinv_nb;
    int64_t dinv_nb;              //   built on demand by the variance path (chol.hip), dropped when the matrix is refactored
    size_t dinv_bytes;
};

#include "tile_index.h"
struct mgpi_pool {
    std::vector<mgpi_pool_block> blocks;   // oldest first
    size_t held = 0, cap = 0;
};

// ---- error plumbing -------------------------------------------------------------------
void mgpi_set_error(const char* fmt, ...);
#define MGPI_HIP_TRY(expr)                                                                   \
    do {                                                                                     \
        hipError_t _e = (expr);                                                              \
        if (_e != hipSuccess) {                                                              \
            mgpi_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,  \
                           __LINE__);                                                        \
            return MGPI_ERR_HIP;                                                             \
        }                                                                                    \
    } while (0)
#define MGPI_REQUIRE(cond, ...)                                                              \
    do {                                                                                     \
        if (!(cond)) {                                                                       \
            mgpi_set_error(__VA_ARGS__);                                                     \
            return MGPI_ERR_ARG;                                                             \
        }                                                                                    \
    } while (0)
#define MGPI_TRY(expr)                                                                       \
    do {                                                                                     \
        int _s = (expr);                                                                     \
        if (_s != 0) return _s;                                                              \
    } while (0)

static inline int64_t mgpi_roundup(int64_t v, int64_t m) { return (v + m - 1) / m * m; }

int mgpi_ensure_work(mgpi_ctx* ctx, size_t bytes);
void mgpi_mat_invalidate(mgpi_mat* m);     // contents change: not a factor any more, cached inverse blocks dropped
int mgpi_pool_alloc(mgpi_ctx* ctx, size_t bytes, double** out, size_t* got);
void mgpi_pool_free(mgpi_ctx* ctx, void* ptr, size_t bytes);

// ---- device-side kernel descriptor (passed by value as a kernel argument) -------------------
#define MGPI_MATERN_POLY 11
struct mgpi_terms_dev {
    int nterms;
    int d;
    int type[MGPI_MAX_TERMS];
    double sig2[MGPI_MAX_TERMS];                    // sigma**2              (kernels.py:131,185)
    double l2[MGPI_MAX_TERMS][MGPI_MAX_D];          // length_d**2           (kernels.py:185,161)
    double inv_l2[MGPI_MAX_TERMS][MGPI_MAX_D];      // RN(1/length_d**2): with one fma correction the quotient (x1-x2)**2/length**2
                                                    // comes out correctly rounded, i.e. equal to the reference's division
    int exact_div;                                  // every length**2 is a normal finite number: the corrected product is valid
    int fast_mix;                                   // 1: one SE term; 2: SE + WhiteNoise (kernel + nugget); 0: the general loop
    // Matern terms: order o = nl + mu with |mu| <= 1/2, and the x-independent constants of Temme's series
    int mat_nl[MGPI_MAX_TERMS];
    double mat_mu[MGPI_MAX_TERMS], mat_two_o[MGPI_MAX_TERMS], mat_norm[MGPI_MAX_TERMS], mat_order[MGPI_MAX_TERMS];
    double mat_gam1[MGPI_MAX_TERMS], mat_gam2[MGPI_MAX_TERMS], mat_gampl[MGPI_MAX_TERMS], mat_gammi[MGPI_MAX_TERMS];
    // half-integer orders o = p + 1/2 <= 10.5: the shape function is exp(-x) * sum_i mat_poly[i] x^i (closed form)
    int mat_half_p[MGPI_MAX_TERMS];                  // p, or -1 for the general path
    double mat_poly[MGPI_MAX_TERMS][MGPI_MATERN_POLY];
};
int mgpi_make_terms(const mgpi_kernel_desc* terms, int nterms, int d, mgpi_terms_dev* out);

// ---- launchers (each enqueues on ctx->stream, no host sync) ----------------------------------
// raw-pointer gemm:  C(m x n, ldc) -= opA(A) * opB(B), contraction length k
//   a_kmajor == 0: A is (m x k) row-major (lda);  a_kmajor == 1: A is stored (k x m) row-major
//   b_kmajor == 0: B is (n x k) row-major (ldb);  b_kmajor == 1: B is stored (k x n) row-major
//   m multiple of 64, n multiple of 64, k multiple of 16.  lower_only: m == n, tiles with j <= i.
int mgpi_launch_gemm(mgpi_ctx* ctx, int a_kmajor, int b_kmajor, int lower_only,
                     int64_t m, int64_t n, int64_t k,
                     const double* A, int64_t lda, const double* B, int64_t ldb,
                     double* C, int64_t ldc);

// trailing update with the next panel's block column first (gemm.hip, chol.hip): C (t x t, lower tiles) -= P P^T
int mgpi_launch_syrk_merged(mgpi_ctx* ctx, int64_t t, int64_t k, const double* P, int64_t ldp, double* C, int64_t ldc, int cw,
                            unsigned* counter);

int mgpi_launch_cov(mgpi_ctx* ctx, const mgpi_terms_dev& t,
                    const double* x1, int64_t ldx1, int64_t n1, int64_t p1,
                    const double* x2, int64_t ldx2, int64_t n2, int64_t p2,
                    int symmetric, int lower_only, int pad_identity,
                    double* out, int64_t ldo);

int mgpi_launch_cov_batch(mgpi_ctx* ctx, const mgpi_terms_dev* tb_dev, int d, const double* x, int64_t ldx, int64_t n, int64_t pn,
                          double* out, int64_t ldo, int64_t stride, int batch);

// panel pieces of the Cholesky (chol.hip)
// multi-rhs 64-row block solves on B (64 x n): trans=0: L_jj^-1 B, trans=1: L_jj^-T B
int mgpi_launch_trsm_cols(mgpi_ctx* ctx, const double* Ljj, int64_t ldl, double* B, int64_t ldb, int64_t n, int trans);
int mgpi_launch_gemm_nn_batched(mgpi_ctx* ctx, int64_t m, int64_t n, int64_t k, const double* A, int64_t lda, int64_t sA,
                                const double* B, int64_t ldb, int64_t sB, double* C, int64_t ldc, int64_t sC, int batch);

void mgpi_prof_begin(mgpi_ctx* ctx, int slot);
void mgpi_prof_end(mgpi_ctx* ctx, int slot);
void mgpi_prof_flops(mgpi_ctx* ctx, int slot, double flops);
