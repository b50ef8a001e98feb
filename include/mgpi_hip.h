/*
 * mgpi_hip.h - C ABI of libmgpi_hip.so: the MI355X (gfx950) device path for the dense
 * Gaussian-process hot path of reedessick/muses-gaussian-process-interpolator ("mgpi").
 *
 * The reference is pure Python + numpy and has no FFI of its own; the
 * entry points below are what a ctypes binding at the reference's three numeric call sites
 * would bind (INTEGRATION.md shows that binding):
 *
 *   (i)   Interpolator._x2cov            mgpi/interpolators.py:208-300  -> mgpi_cov_matrix*
 *         Kernel.cov (row-paired)        mgpi/kernels.py:89-93,128-131,180-185,287-293 -> mgpi_cov_paired
 *   (ii)  np.linalg.inv / slogdet        mgpi/interpolators.py:92,179,339,343 -> mgpi_potrf*, mgpi_potrs*
 *   (iii) the `@` contractions           mgpi/interpolators.py:100,125,189,199 -> mgpi_predict_mean*,
 *                                                                        mgpi_predict_var*, mgpi_condition_cov*
 *
 * Conventions
 *   - everything is IEEE fp64; matrices are row-major (numpy C order), exactly as the
 *     reference's ndarrays (interpolators.py:247,294).
 *   - plain pointers and sizes only; no exceptions cross the ABI.
 *   - every function returns an int status:  0 ok;  > 0 = 1-based index of the first
 *     non-positive pivot (matrix not positive definite; the reference's
 *     `assert s > 0, 'covariance is not positive definite!'`, interpolators.py:340, and
 *     LinAlgError from inv);  < 0 = MGPI_ERR_* below, text via mgpi_last_error().
 *   - host buffers are borrowed for the duration of the call.  Device buffers are owned by
 *     the library and referenced through opaque mgpi_mat handles, so K / L / alpha stay in
 *     HBM between compress() and predict().
 *   - a ctx is bound to one GPU and is not re-entrant; distinct ctxs may be used from
 *     distinct threads.
 *   - there is NO CPU fallback in this library: without a HIP device mgpi_ctx_create fails.
 */
#ifndef MGPI_HIP_H
#define MGPI_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MGPI_MAX_D      16   /* max input dimensionality (NDKernel length0..length{D-1}, kernels.py:97-118) */
#define MGPI_MAX_TERMS   8   /* max summands of a CombinedKernel (kernels.py:191-293) */

/* kernel summand types (a1-a4 of SURVEY.md section 8) */
#define MGPI_TERM_SE     0   /* SquaredExponentialKernel: sigma^2 exp(-sum_d (dx_d/l_d)^2), NO 1/2 (kernels.py:185) */
#define MGPI_TERM_WHITE  1   /* WhiteNoiseKernel: sigma^2 [x1 == x2 exactly in every coordinate] (kernels.py:131) */
#define MGPI_TERM_MATERN 2   /* MaternKernel: sigma^2 2^(1-o)/Gamma(o) r^o K_o(r), r = sqrt(2 o sum_d (dx_d/l_d)^2) (kernels.py:155-162);
                              * at r == 0 the limit sigma^2 is returned where the reference evaluates 0*inf = NaN */

#define MGPI_ERR_ARG     (-1)  /* bad argument (shape mismatch, D or nterms too large, null pointer) */
#define MGPI_ERR_HIP     (-2)  /* HIP runtime error (incl. no device) */
#define MGPI_ERR_ALLOC   (-3)  /* device allocation failed */
#define MGPI_ERR_STATE   (-4)  /* handle used in the wrong state (e.g. potrs on an unfactored matrix) */
#define MGPI_ERR_COMM    (-5)  /* RCCL error (librccl not loadable, communicator failure); text via mgpi_last_error() */

/* one summand of the (flattened) CombinedKernel, built from Kernel.params in summand order */
typedef struct mgpi_kernel_desc {
    int32_t type;                 /* MGPI_TERM_* */
    int32_t reserved;
    double  sigma;                /* params[0] of SE / WhiteNoise, params[1] of Matern */
    double  order;                /* Matern order, 0 < o <= 120 (params[0] of MaternKernel); unused otherwise.  Half-integer
                                   * orders up to 10.5 are evaluated by their closed form, the rest by Temme / Steed */
    double  length[MGPI_MAX_D];   /* SE: params[1:1+D]; Matern: params[2:2+D] */
} mgpi_kernel_desc;

typedef struct mgpi_ctx mgpi_ctx;   /* one GPU + its streams and workspace */
typedef struct mgpi_mat mgpi_mat;   /* a device-resident row-major fp64 matrix (rows x cols) */

/* ---- library / context ------------------------------------------------------------- */
const char* mgpi_last_error(void);                 /* thread-local text of the last failure */
int  mgpi_device_count(int* count);
/* nb = outer Cholesky block (multiple of 128; 0 -> default 512, BASELINE.json config 3) */
int  mgpi_ctx_create(int device_id, int nb, mgpi_ctx** out);
int  mgpi_ctx_destroy(mgpi_ctx* ctx);
int  mgpi_ctx_sync(mgpi_ctx* ctx);                 /* wait for all queued device work */
/* Device memory of freed matrices is kept in a per-context pool and handed to later mgpi_mat_alloc calls of a
 * fitting size: the hyper-parameter loops of the reference (interpolators.py:382, :514) rebuild K at a fixed N
 * thousands of times, and a hipMalloc/hipFree pair per call costs more than the factorisation below N ~ 16384.
 * The pool is capped at a quarter of the device memory (MGPI_POOL_MB overrides, 0 disables); trim releases it. */
int  mgpi_ctx_trim(mgpi_ctx* ctx);
/* Positive-definiteness rule of the factorisations (dense and NNGP): pivot k must exceed tol * K_kk(original).
 * Default 100 eps (MGPI_PIVOT_TOL overrides at ctx creation): it makes the exactly singular K of duplicated source
 * rows fail, as the reference's inv does (LinAlgError), instead of factoring rounding noise.  It is STRICTER than
 * LAPACK: an ill-conditioned but positive definite K (cond ~ 1/(100 eps)) that np.linalg.inv accepts
 * (interpolators.py:92,179) is rejected.  tol = 0 gives LAPACK's plain "pivot > 0" rule. */
int  mgpi_ctx_set_pivot_tol(mgpi_ctx* ctx, double tol);
/* raw hipStream_t of the ctx (as void*), for callers that time with HIP events */
void* mgpi_ctx_stream(mgpi_ctx* ctx);
/* event timing on the ctx stream: tic records, toc records+synchronises and returns ms */
int  mgpi_ctx_tic(mgpi_ctx* ctx);
int  mgpi_ctx_toc(mgpi_ctx* ctx, double* ms);

/* ---- device matrices --------------------------------------------------------------- */
/* rows are padded to a multiple of 128.  Matrices with cols <= MGPI_MAX_D (coordinate tables, vectors) keep
 * their natural row stride (ld == cols); wider ones get ld padded to a multiple of 128.  The pads are zero. */
int  mgpi_mat_alloc(mgpi_ctx* ctx, int64_t rows, int64_t cols, mgpi_mat** out);
/* same, but always tile-padded (ld multiple of 128) whatever cols is: the form cov/potrf/trsm/gemm operate on */
int  mgpi_mat_alloc_wide(mgpi_ctx* ctx, int64_t rows, int64_t cols, mgpi_mat** out);
int  mgpi_mat_free(mgpi_mat* m);
int  mgpi_mat_upload(mgpi_mat* m, const double* host, int64_t host_ld);     /* host (rows x cols), row stride host_ld */
int  mgpi_mat_download(const mgpi_mat* m, double* host, int64_t host_ld);
int  mgpi_mat_shape(const mgpi_mat* m, int64_t* rows, int64_t* cols, int64_t* ld);
void* mgpi_mat_devptr(const mgpi_mat* m);                                   /* raw device pointer (double*) */
/* declare that m holds (factored != 0) or no longer holds a Cholesky factor: for buffers whose content arrived by a
 * broadcast from the rank that factored it */
int  mgpi_mat_mark_factored(mgpi_mat* m, int factored);
/* copy src into dst (same logical shape), e.g. keep K before factoring in place */
int  mgpi_mat_copy(mgpi_ctx* ctx, const mgpi_mat* src, mgpi_mat* dst);
/* fill the strict upper triangle from the lower one (square), for downloads of a lower-only build */
int  mgpi_mat_symmetrize(mgpi_ctx* ctx, mgpi_mat* m);

/* zero the strict upper triangle (square): turns an in-place Cholesky result into a clean lower factor */
int  mgpi_mat_tril(mgpi_ctx* ctx, mgpi_mat* m);
/* diag (n x 1 vector) <- diagonal of the square matrix m */
int  mgpi_mat_diag(mgpi_ctx* ctx, const mgpi_mat* m, mgpi_mat* diag);
/* m[i][i] += shift (square): the jitter of rvs()'s retry loop stays on the device */
int  mgpi_mat_add_diag(mgpi_ctx* ctx, mgpi_mat* m, double shift);
/* m[i][j] <- m[i][j] / (s[i] * s[j]), s an (n x 1) vector: covariance -> correlation (interpolators.py:321) */
int  mgpi_mat_unscale(mgpi_ctx* ctx, mgpi_mat* m, const mgpi_mat* s);

/* ---- (i) kernel operators ---------------------------------------------------------- */
/* Kernel.cov: row-paired, x1 and x2 both (p, d) row-major host arrays -> out (p,) host */
int  mgpi_cov_paired(mgpi_ctx* ctx, const double* x1, const double* x2, int64_t p, int d,
                     const mgpi_kernel_desc* terms, int nterms, double* out);

/* Interpolator._x2cov on host arrays: out (n1, n2) row-major host.  x2 == NULL means x2 = x1. */
int  mgpi_cov_matrix(mgpi_ctx* ctx, const double* x1, int64_t n1, const double* x2, int64_t n2, int d,
                     const mgpi_kernel_desc* terms, int nterms, double* out);

/* device-resident form: X1 (n1 x d), X2 (n2 x d) or NULL (= X1) -> out (n1 x n2), all mgpi_mat.
 * lower_only != 0 (requires X2 == NULL): only tiles on/below the diagonal are written - the
 * form potrf consumes; halves the work. */
int  mgpi_cov_matrix_dev(mgpi_ctx* ctx, const mgpi_mat* X1, const mgpi_mat* X2,
                         const mgpi_kernel_desc* terms, int nterms, int lower_only, mgpi_mat* out);

/* ---- (ii) factor / solve  (the role of np.linalg.inv + slogdet) ---------------------- */
/* in-place blocked right-looking Cholesky K = L L^T, lower triangle, row-major; the strict
 * upper triangle is not referenced.  *logdet = 2 sum log L_ii (may be NULL).
 * returns 0, or k > 0 if the k-th (1-based) pivot is not positive (matrix content then undefined) */
int  mgpi_potrf_dev(mgpi_ctx* ctx, mgpi_mat* K, double* logdet);
/* solve L L^T X = B in place, B (n x nrhs) device matrix.
 * One right-hand side: block substitution with the factor's nb x nb diagonal blocks; for the BACKWARD sweep the diagonal-block
 * solves are products with the inverted blocks (built once per factor, cached on the handle, shared with mgpi_predict_var_dev)
 * when the factor's diagonal spans at most 1e4 (MGPI_TRSV_DINV_RATIO) as the status read-back of mgpi_potrf_dev /
 * mgpi_posv_dev saw it - cond K up to ~4e10 for squared-exponential + nugget matrices, profiles/r03_dinv_accuracy.md - and plain substitutions otherwise (ill-conditioned factors,
 * handles whose status this library never read, MGPI_TRSV_DINV=0).  The two forms agree to ~cond(L_bb) eps. */
int  mgpi_potrs_dev(mgpi_ctx* ctx, const mgpi_mat* L, mgpi_mat* B);
/* factor AND solve (LAPACK's dposv):  K <- L,  B <- K^-1 B,  *logdet as above.  The reference's
 * inv(K) @ f (mgpi/interpolators.py:92-100, :179-189).  For one right-hand side (B an (n x 1) vector) the
 * forward sweep runs beside the factorisation, so the solve costs the backward sweep only; the arithmetic
 * is that of mgpi_potrf_dev followed by mgpi_potrs_dev.  Same return convention as mgpi_potrf_dev. */
int  mgpi_posv_dev(mgpi_ctx* ctx, mgpi_mat* K, mgpi_mat* B, double* logdet);
/* factor, solve and forward-solve "riders" in one call - Interpolator.condition's three steps (mgpi/interpolators.py:148-199)
 * fused: KR is a wide matrix of n columns and pn + m rows (pn = n rounded up to 128); mgpi_cov_riders_dev writes
 * K(x, x; kernel + nugget) (lower tiles) into rows [0, n) and the riders K(x*, x; kernel) into rows [pn, pn + m);
 * mgpi_posv_riders_dev leaves L in the top square, V^T = K* L^-T in the rider rows (solved by the panels' own substitutions,
 * updated inside the trailing-update launches) and K^-1 B in B (an (n x 1) vector, may be NULL); mgpi_riders_var_dev turns the
 * rider rows into the posterior variance var_i = k(x*_i, x*_i) - |V^T_i|^2.  mgpi_mat_square_view hands out the factor as a
 * square handle for the entry points that take L. */
int  mgpi_cov_riders_dev(mgpi_ctx* ctx, const mgpi_mat* X, const mgpi_mat* Xs, const mgpi_kernel_desc* kterms, int nk,
                         const mgpi_kernel_desc* sterms, int ns, mgpi_mat* KR);
int  mgpi_posv_riders_dev(mgpi_ctx* ctx, mgpi_mat* KR, mgpi_mat* B, double* logdet);
int  mgpi_riders_var_dev(mgpi_ctx* ctx, const mgpi_mat* KR, const mgpi_mat* Xs, const mgpi_kernel_desc* terms, int nterms,
                         mgpi_mat* var);
int  mgpi_mat_square_view(const mgpi_mat* m, mgpi_mat** out);
/* host convenience: b (n, nrhs) row-major host in, x out (may alias) */
int  mgpi_potrs(mgpi_ctx* ctx, const mgpi_mat* L, const double* b, int64_t n, int nrhs, double* x);
/* Interpolator.loglikelihood (interpolators.py:331-343) for `batch` kernel parameter sets at once, x and f resident:
 * terms holds batch * nterms descriptions (problem b: terms[b * nterms .. ]), every problem the same summand structure.
 * out[b] = -1/2 f^T K_b^-1 f - 1/2 logdet K_b - n/2 log 2 pi.  A K_b that is not positive definite gives out[b] = -inf and
 * info[b] = its failing pivot (info may be NULL) without disturbing the other members.  This is the inner loop of
 * optimize_kernel / sample_kernel (:369-387, :474-478, :513-514, :587-601): for n <= 4096 all problems share every launch
 * (one kernel-matrix launch, n/64 panel-step launches, one reduction); larger n run one after the other. */
int  mgpi_loglikelihood_batch(mgpi_ctx* ctx, const mgpi_mat* X, const mgpi_mat* f, const mgpi_kernel_desc* terms, int nterms,
                              int batch, double* out, int32_t* info);
/* triangular half-solves used by predict_var / rvs:  L X = B  (trans=0)  or  L^T X = B (trans=1) */
int  mgpi_trsm_dev(mgpi_ctx* ctx, const mgpi_mat* L, mgpi_mat* B, int trans);

/* ---- (iii) posterior contractions -------------------------------------------------- */
/* mean[i] = sum_j k(xs_i, x_j) alpha_j   (interpolators.py:117,125) - K* is never materialised.
 * Xs (m x d), X (n x d), alpha (n x 1), mean (m x 1) */
int  mgpi_predict_mean_dev(mgpi_ctx* ctx, const mgpi_mat* Xs, const mgpi_mat* X,
                           const mgpi_kernel_desc* terms, int nterms,
                           const mgpi_mat* alpha, mgpi_mat* mean);
/* var[i] = k(xs_i,xs_i) - || L^-1 k(x, xs_i) ||^2  = diag of interpolators.py:199.
 * m is processed in slabs of at most m_slab queries (0 -> library default) so the n x m_slab
 * workspace fits HBM. */
int  mgpi_predict_var_dev(mgpi_ctx* ctx, const mgpi_mat* Xs, const mgpi_mat* X,
                          const mgpi_kernel_desc* terms, int nterms,
                          const mgpi_mat* L, int64_t m_slab, mgpi_mat* var);
/* full posterior covariance cov = K** - (L^-1 K*^T)^T (L^-1 K*^T)  (m x m), for small m
 * (interpolators.py:148,157,199) */
int  mgpi_condition_cov_dev(mgpi_ctx* ctx, const mgpi_mat* Xs, const mgpi_mat* X,
                            const mgpi_kernel_desc* terms, int nterms,
                            const mgpi_mat* L, mgpi_mat* cov);

/* ---- building blocks exported for tests / the multi-GPU driver ----------------------- */
/* C (m x n) -= A (m x k) * B^T   with B (n x k)            (trailing update, interpolators.py:179's role)
 * lower_only != 0: m == n, only tiles on/below the diagonal are updated (syrk when A == B) */
int  mgpi_gemm_nt_dev(mgpi_ctx* ctx, const mgpi_mat* A, const mgpi_mat* B, mgpi_mat* C, int lower_only);
/* views: a (rows x cols) window of an existing matrix starting at (r0, c0); shares storage,
 * r0/c0/rows/cols must be multiples of 64 or reach the padded edge */
int  mgpi_mat_view(const mgpi_mat* m, int64_t r0, int64_t c0, int64_t rows, int64_t cols, mgpi_mat** out);

/* non-owning handle over external device memory (rows x cols doubles, row stride ld, 16-byte aligned), e.g. a
 * communication buffer of the multi-GPU driver; factored != 0 marks it as holding a Cholesky factor */
int  mgpi_mat_wrap(mgpi_ctx* ctx, void* devptr, int64_t rows, int64_t cols, int64_t ld, int factored, mgpi_mat** out);
/* out (all padded entries) = k(X1[i], X2[j]) for i < n1_valid, j < n2_valid, zero elsewhere: a local block of the
 * block-cyclic K; the driver sets the unit diagonal of padded global indices */
int  mgpi_cov_block_dev(mgpi_ctx* ctx, const mgpi_mat* X1, int64_t n1_valid, const mgpi_mat* X2, int64_t n2_valid,
                        const mgpi_kernel_desc* terms, int nterms, mgpi_mat* out);
/* C (m x n) -= A (m x k) * B (k x n)   (row-major B: the update of the multi-rhs triangular solve) */
int  mgpi_gemm_nn_dev(mgpi_ctx* ctx, const mgpi_mat* A, const mgpi_mat* B, mgpi_mat* C);
/* C = -(A B): C is written, not read */
int  mgpi_gemm_nn_neg_dev(mgpi_ctx* ctx, const mgpi_mat* A, const mgpi_mat* B, mgpi_mat* C);
/* C -= A B^T restricted to the tiles on/below the GLOBAL diagonal of a 2-D block-cyclic matrix: C = this rank's local
 * trailing matrix starting at local block (bi0, bj0) of an nb-blocked matrix on a P x Q grid, this rank at (p, q) */
int  mgpi_gemm_nt_stair_dev(mgpi_ctx* ctx, const mgpi_mat* A, const mgpi_mat* B, mgpi_mat* C, int P, int p, int Q, int q,
                            int bi0, int bj0, int nb);
/* A <- A L^-T : panel trsm with a factored diagonal block L (w x w) on a block of rows A (m x w) */
int  mgpi_trsm_right_dev(mgpi_ctx* ctx, const mgpi_mat* L, mgpi_mat* A);
/* out rows [j w, (j+1) w) := -(A[r0 + j row_step : + w, c0 + j col_step : + w])^-1 for `batch` lower-triangular w x w
 * blocks at a constant stride (the diagonal blocks a rank owns in its local block-cyclic factor).  Used by the
 * VARIANCE sweep only, where a diagonal-block solve then becomes one gemm (error budget: DESIGN.md); w % 64 == 0 */
int  mgpi_trtri_neg_dev(mgpi_ctx* ctx, const mgpi_mat* A, int64_t r0, int64_t c0, int64_t row_step, int64_t col_step, int64_t w,
                        int batch, mgpi_mat* out);
/* --- more pieces of the block-cyclic driver: everything it does to device memory goes through this library ---------- */
/* factor K in place WITHOUT reading the result back: the first failing pivot (reported as pivot_base + k, 1-based) is
 * min-accumulated in the ctx and sum log L_ii is added to a ctx accumulator; one status_read at the end of the whole
 * factorisation replaces a host round trip per block step.  status_reset clears both. */
int  mgpi_potrf_enqueue_dev(mgpi_ctx* ctx, mgpi_mat* K, int64_t pivot_base, const mgpi_mat* diag0);
/* the same for a TALL block column (m x w window, m >= w, multiples of 64) whose top w x w block is the diagonal block:
 * that block is factored and the rows below are solved against it with the fused step kernels of the single-GPU path -
 * the panel of a block step on grids with one process row, where the whole block column is local
 * (role: the dpotrf + dtrsm pair inside np.linalg.inv's factorisation, mgpi/interpolators.py:92,179) */
int  mgpi_potrf_panel_enqueue_dev(mgpi_ctx* ctx, mgpi_mat* panel, int64_t pivot_base, const mgpi_mat* diag0);
/* diag0 (both): NULL, or a device vector with the ORIGINAL diagonal entries K_ii of the rows being factored - what a pivot is
 * measured against (pivot > tol * K_ii).  A diagonal block that earlier block steps have updated no longer shows them (the
 * entry of a duplicated source row is rounding noise by then), so the driver keeps diag(K) from the build and hands it in. */
int  mgpi_ctx_status_reset(mgpi_ctx* ctx);
int  mgpi_ctx_status_read(mgpi_ctx* ctx, int64_t* pivot, double* logdiag);      /* synchronises ctx's stream */
/* order two contexts of one device: what `signaler` has queued so far precedes what `waiter` queues from now on */
int  mgpi_ctx_wait(mgpi_ctx* waiter, mgpi_ctx* signaler);
/* the merged form of the driver's trailing update: mgpi_gemm_nt_stair_merged_dev updates the first `first_cols` local block columns
 * of C first, stores them write-through and counts their tiles on ctx's progress counter (*live_tiles = how far it moves);
 * mgpi_ctx_wait_count makes another context of the same device wait for a count instead of a kernel boundary, so that the next
 * super-panel is factored while the rest of the same launch runs; mgpi_ctx_count_reset zeroes the counter (once per factorisation);
 * mgpi_ctx_streams_overlap asks the device whether kernels of the two contexts run side by side at all (bounded handshake) -
 * where they do not (profilers, serialising debug switches) the driver keeps the two-launch form. */
int  mgpi_gemm_nt_stair_merged_dev(mgpi_ctx* ctx, const mgpi_mat* A, const mgpi_mat* B, mgpi_mat* C, int P, int p, int Q, int q,
                                   int bi0, int bj0, int nb, int first_cols, int64_t* live_tiles);
int  mgpi_ctx_wait_count(mgpi_ctx* waiter, mgpi_ctx* owner, int64_t target);
int  mgpi_ctx_count_reset(mgpi_ctx* ctx);
int  mgpi_ctx_streams_overlap(mgpi_ctx* a, mgpi_ctx* b, int* yes);
/* from now on this context's launches go to its high-priority stream (the look-ahead context of the driver: its panels and
 * broadcasts must not queue behind the main context's trailing update); call on an idle context */
int  mgpi_ctx_prefer_high_priority(mgpi_ctx* ctx);
/* m[i][i] = value, i < min(rows, cols) of the handle (use a view to address part of a diagonal) */
int  mgpi_mat_fill_diag(mgpi_ctx* ctx, mgpi_mat* m, double value);
/* y += a x, (n x 1) vectors */
int  mgpi_axpy_dev(mgpi_ctx* ctx, double a, const mgpi_mat* x, mgpi_mat* y);
/* `count` blocks of `rows` full-width rows: dst rows [dst_r0 + j dst_step, ..) <- src rows [src_r0 + j src_step, ..) */
int  mgpi_copy_blocks_dev(mgpi_ctx* ctx, const mgpi_mat* src, int64_t src_r0, int64_t src_step, mgpi_mat* dst, int64_t dst_r0,
                          int64_t dst_step, int64_t rows, int64_t count);
/* out[c] = sum_r V[r][c]^2 over the padded rows of V; out a (cols x 1) vector */
int  mgpi_colsq_dev(mgpi_ctx* ctx, const mgpi_mat* V, mgpi_mat* out);

/* single-vector half-solve with a factored diagonal block (<= 4096 rows): L y = b (trans=0) / L^T x = b (trans=1) */
int  mgpi_trsv_dev(mgpi_ctx* ctx, const mgpi_mat* L, mgpi_mat* v, int trans);
/* y -= A x (trans=0) or y -= A^T x (trans=1); x, y vectors (ld == 1) */
int  mgpi_gemv_dev(mgpi_ctx* ctx, const mgpi_mat* A, const mgpi_mat* x, mgpi_mat* y, int trans);

/* ---- RCCL over xGMI (north_star: "2D block-cyclic panels with RCCL broadcast/reduce over xGMI") ---
 * One process per GPU.  Rank 0 makes a unique id and ships its MGPI_COMM_ID_BYTES bytes to the other ranks by any means
 * (mgpi_amd.rendezvous uses a file: the ranks of one node share a filesystem); every rank then calls mgpi_comm_create
 * with its ctx (the communicator lives on that ctx's GPU).  mgpi_comm_split makes the process-row / process-column
 * communicators of the P x Q grid.  Collectives take device pointers to contiguous doubles and are enqueued on the
 * stream of the ctx passed with them - a second ctx of the same device gives the look-ahead stream - with no host
 * synchronisation.  librccl.so.1 is dlopen'ed on first use (MGPI_RCCL_LIB overrides the name); failures return
 * MGPI_ERR_COMM.  Broadcasts are grouped point-to-point sends from the root (xGMI is point to point: the 1-3 peers of a
 * grid row / column sit on separate links); MGPI_BCAST=rccl selects ncclBroadcast.
 * The reference has no communication layer; these calls exist only because K no longer fits the role of ONE
 * np.linalg.inv operand (interpolators.py:92,179) when it is spread over GPUs. */
#define MGPI_COMM_ID_BYTES 128
typedef struct mgpi_comm mgpi_comm;
int  mgpi_comm_unique_id(void* out /* MGPI_COMM_ID_BYTES */);
int  mgpi_comm_create(mgpi_ctx* ctx, const void* unique_id, int rank, int world, mgpi_comm** out);
int  mgpi_comm_split(mgpi_comm* comm, int color, int key, mgpi_comm** out);   /* same color = same new communicator, ranked by key */
int  mgpi_comm_destroy(mgpi_comm* comm);
int  mgpi_comm_info(const mgpi_comm* comm, int* rank, int* size);
int  mgpi_comm_bcast(mgpi_comm* comm, mgpi_ctx* ctx, double* buf, int64_t count, int root);
int  mgpi_comm_reduce_sum(mgpi_comm* comm, mgpi_ctx* ctx, double* buf, int64_t count, int root);   /* in place on root */
int  mgpi_comm_allreduce(mgpi_comm* comm, mgpi_ctx* ctx, double* buf, int64_t count, int op /* 0 sum, 1 max */);
int  mgpi_comm_allgather(mgpi_comm* comm, mgpi_ctx* ctx, const double* send, double* recv, int64_t count);
int  mgpi_comm_barrier(mgpi_comm* comm, mgpi_ctx* ctx);   /* all ranks arrived and ctx's stream drained */

/* ---- NearestNeighborInterpolator (mgpi/interpolators.py:612-871) ----------------------
 * Host pointers in and out; the reference's own neighbour sets are host-side index lists too (:744, :833). */
#define MGPI_NN_MAX_NEIGHBORS 62
/* _2neighbors (:654-744).  source_x (n x d) must already be sorted by rank (_2sorted, :645-652; source_rank ascending).
 * target_x == NULL: reference-set mode, sample i against the samples in front of it (m is ignored, n targets).
 * rank_is_sum: ranks are sum(x) (order_by_index None) rather than one coordinate; it selects the lower bound on the
 * distance that lets the scan stop early (the scan is O(N sqrt(k N)) in 2-D instead of the reference's O(N^2)).
 * out_index: (m x k) closest first, -1 padded; out_count: (m).  Euclidean distance, samples ranked above the target or
 * coinciding with it are skipped, as in the reference; equal distances go to the lower index. */
int  mgpi_nn_neighbors(mgpi_ctx* ctx, const double* source_x, const double* source_rank, int64_t n, int d,
                       const double* target_x, const double* target_rank, int64_t m, int k, int rank_is_sum,
                       int32_t* out_index, int32_t* out_count);
/* _2diag / _sample2diag (:751-792): mean and variance of every target given its neighbour set; terms = the kernel's
 * nkernel summands followed by the nugget's.  A point without neighbours gets (0, k(x,x) + nugget(x,x)); a neighbour
 * matrix that is not positive definite gives NaN for that point. */
int  mgpi_nn_condition(mgpi_ctx* ctx, const double* target_x, int64_t m, const double* source_x, const double* source_f,
                       int64_t n, int d, const int32_t* index, const int32_t* count, int k, const mgpi_kernel_desc* terms,
                       int nterms, int nkernel, double* out_mean, double* out_diag);

/* measured throughput probes used by bench.py's roofline leg (not part of the hot path) */
int  mgpi_probe_mfma_f64(mgpi_ctx* ctx, int iters, double* tflops);
int  mgpi_probe_hbm(mgpi_ctx* ctx, int64_t bytes, int iters, double* read_gbs, double* write_gbs);

/* look-ahead on/off (default on; MGPI_NO_LOOKAHEAD=1 starts with it off).  Off = the panel and the trailing update run
 * back to back on one stream: used by bench.py to time the trailing-update kernel without a concurrent panel */
int  mgpi_ctx_set_lookahead(mgpi_ctx* ctx, int on);
/* per-kernel accumulated HIP-event time of the last mgpi_potrf_dev when profiling is enabled */
int  mgpi_ctx_set_profile(mgpi_ctx* ctx, int on);
/* the same for a subset of the slots below (bit s = slot s): every recorded event is a marker between two kernels of its
 * stream; bench.py keeps only the dominant kernel's slot on inside its timed region (no reference counterpart: measurement) */
int  mgpi_ctx_set_profile_slots(mgpi_ctx* ctx, unsigned mask);
/* which: 0 = trailing update (lower syrk launches), 1 = panel (potf2+trsm+inner gemm; only when look-ahead is off),
 * 2 = whole potrf, 3 = look-ahead update of the next panel's block column, 4 = the big (main-stream) update launches of
 * the multi-rhs forward solve V = L^-1 K*^T of the variance path, 5 = the fused predict-mean kernel; ms and launches */
int  mgpi_ctx_get_profile(mgpi_ctx* ctx, int which, double* ms, int64_t* launches);
/* algorithmic flops of the launches recorded in slot `which` since profiling was switched on: for slot 0 the sum of
 * t (t + 1) K over the lower-triangular trailing updates (t rows, contraction length K); slot 4: 2 t m K of the
 * solve's update launches; slot 5: the number of (query, source) kernel evaluations */
int  mgpi_ctx_get_profile_flops(mgpi_ctx* ctx, int which, double* flops);

#ifdef __cplusplus
}
#endif
#endif /* MGPI_HIP_H */
