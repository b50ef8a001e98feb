// NearestNeighborInterpolator on the device (mgpi/interpolators.py:612-871; Datta et al. 2016).
//
// Two kernels, both embarrassingly parallel over the points being conditioned:
//
//  nn_search_kernel     neighbour selection (_2neighbors, :654-744): one thread per target scans the reference set
//                       (already sorted by rank on the host, :645-652) in LDS-staged slabs of 256 samples, from the
//                       target's own rank downwards, and keeps the KMAX closest admissible samples in registers
//                       (insertion network, fully unrolled); it stops as soon as the rank gap alone exceeds the k-th
//                       best distance.  fp64 VALU work, ~(3 D + 8) flops per (target, sample) pair visited.
//  nn_condition_kernel  the per-point GP (_sample2diag, :751-776): a group of G = 16/32/64 lanes builds the
//                       (c+2) x c matrix [K(nbrs,nbrs; kernel+nugget); k(x, nbrs; kernel); f(nbrs)] in LDS, lane r owning
//                       row r, and runs a left-looking Cholesky on it.  The two extra rows come out as L^-1 k* and
//                       L^-1 f, so mean = <L^-1 k*, L^-1 f> and var = k(x,x) - |L^-1 k*|^2 need no separate solve.
//                       Latency bound (c ~ 10): what matters is that 10^5..10^6 of them run at once.
//
// Selection rules kept from the reference, quirks included: distances are plain Euclidean (not scaled by the kernel
// lengths, :725); only samples ranked at or below the target are admissible (:695-696) - in reference-set mode, the
// samples in front of it; samples that coincide exactly with the target are skipped (:706-718).  One deliberate
// choice: equal distances are ordered by index in the sorted set, where numpy's default argsort (:726) leaves the
// order unspecified.
#include "mgpi_internal.h"
#include "cov_device.h"

#define NN_SLAB 256

// sum_k v[k] in numpy's order for a contiguous run of n <= 16 doubles (pairwise_sum: plain loop below 8 terms, eight
// partial sums above), so that near-ties in distance or rank fall the same way as in the reference
__device__ __forceinline__ double nn_numpy_sum(const double (&v)[MGPI_MAX_D], int n) {
    if (n < 8) {
        double res = 0.0;
#pragma unroll
        for (int k = 0; k < 7; ++k)
            if (k < n) res = __dadd_rn(res, v[k]);
        return res;
    }
    double r[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) r[k] = v[k];
    if (n == 16) {
#pragma unroll
        for (int k = 0; k < 8; ++k) r[k] = __dadd_rn(r[k], v[8 + k]);
    }
    double res = __dadd_rn(__dadd_rn(__dadd_rn(r[0], r[1]), __dadd_rn(r[2], r[3])),
                           __dadd_rn(__dadd_rn(r[4], r[5]), __dadd_rn(r[6], r[7])));
    if (n < 16) {
#pragma unroll
        for (int k = 8; k < 15; ++k)
            if (k < n) res = __dadd_rn(res, v[k]);
    }
    return res;
}

// Search order and the early exit.  A target's admissible samples are those ranked at or below it, i.e. a prefix
// [0, ub) of the sorted reference set.  The prefix is scanned BACKWARDS, closest ranks first, because the rank bounds
// the distance from below:  ordered by one coordinate, (rank_t - rank_j)^2 is one of the summands of the distance;
// ordered by sum(x), Cauchy-Schwarz gives (rank_t - rank_j)^2 <= D * dist.  Once that bound exceeds the k-th best
// distance found so far no sample further down can enter the list and the thread is done; the block stops when all
// its threads are.  That turns the O(N^2) scan of the reference's loop (:686-744) into O(N * band), band ~ sqrt(k N)
// samples in 2-D.  Targets arrive sorted by rank (the host permutes them back afterwards), so the threads of a
// block want the same slabs.  Ties: a backward scan meets the higher index first, so a later candidate at an equal
// distance is placed IN FRONT of its equals - the list stays ordered by (distance, index), the stable rule.
template <int KMAX>
__global__ __launch_bounds__(NN_SLAB) void nn_search_kernel(const double* __restrict__ sx, const double* __restrict__ srank,
                                                            int64_t n, int d, const double* __restrict__ tx,
                                                            const double* __restrict__ trank, int64_t m, int within,
                                                            int k, int bound_div, int32_t* __restrict__ idx,
                                                            int32_t* __restrict__ cnt) {
    __shared__ double slab[NN_SLAB * MGPI_MAX_D];
    __shared__ double slab_rank[NN_SLAB];
    __shared__ unsigned long long blk_hi;
    const int tid = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * NN_SLAB + tid;
    const bool active = i < m;

    double xt[MGPI_MAX_D];
#pragma unroll
    for (int c = 0; c < MGPI_MAX_D; ++c) xt[c] = (active && c < d) ? tx[i * d + c] : 0.0;
    const double my_rank = active ? trank[i] : 0.0;

    // ub: number of admissible leading samples.  Reference-set mode: the samples in front of the target that are
    // ranked at or below it (all of them when the set is sorted, :695-696 with discard_index = i);
    // target mode: upper_bound(source_rank, my_rank).
    int64_t ub = 0;
    if (active) {
        int64_t lo = 0, hi = within ? i : n;
        if (within && (i == 0 || srank[i - 1] <= my_rank)) {
            lo = hi;
        } else {
            while (lo < hi) {
                const int64_t mid = (lo + hi) >> 1;
                if (srank[mid] <= my_rank) lo = mid + 1; else hi = mid;
            }
        }
        ub = lo;
    }
    if (tid == 0) blk_hi = 0ull;
    __syncthreads();
    if (active && ub > 0) atomicMax(&blk_hi, (unsigned long long)ub);
    __syncthreads();
    const int64_t top = (int64_t)blk_hi;          // the block scans [0, top) downwards

    double bd[KMAX];
    int32_t bi[KMAX];
#pragma unroll
    for (int p = 0; p < KMAX; ++p) {
        bd[p] = __builtin_huge_val();
        bi[p] = -1;
    }
    double kth = __builtin_huge_val();            // bd[k - 1]: the distance a candidate has to beat (or tie)
    bool done = !active || ub == 0;
    // slack of the rank bound under sum(x) ordering: the ranks carry rounding of up to D ulps of their magnitude
    const double rank_eps = (bound_div > 1) ? 8.0 * bound_div * 2.220446049250313e-16 : 0.0;

    for (int64_t j0 = (top > 0 ? (top - 1) / NN_SLAB * NN_SLAB : 0); top > 0 && j0 >= 0; j0 -= NN_SLAB) {
        __syncthreads();
        {
            const int64_t j = j0 + tid;
            if (j < top) {
                for (int c = 0; c < d; ++c) slab[tid * d + c] = sx[j * d + c];
                slab_rank[tid] = srank[j];
            }
        }
        __syncthreads();
        const int jn = (int)((top - j0 < NN_SLAB) ? (top - j0) : NN_SLAB);
        if (!done) {
            for (int jj = jn - 1; jj >= 0; --jj) {
                const int64_t j = j0 + jj;
                if (j >= ub) continue;
                const double sr = slab_rank[jj];
                double gap = my_rank - sr;                                   // >= 0 inside the admissible prefix
                gap -= rank_eps * (fabs(my_rank) + fabs(sr));
                if (gap > 0.0 && gap * gap > kth * (double)bound_div * (1.0 + 1e-12)) {
                    done = true;                                             // nothing further down can be closer
                    break;
                }
                double sq[MGPI_MAX_D];
                bool same = true;
#pragma unroll
                for (int c = 0; c < MGPI_MAX_D; ++c) {
                    if (c < d) {
                        const double s = slab[jj * d + c];
                        const double df = s - xt[c];                         // (source_x - x)**2, :725
                        same = same && (s == xt[c]);
                        sq[c] = __dmul_rn(df, df);
                    } else {
                        sq[c] = 0.0;
                    }
                }
                const double dist = nn_numpy_sum(sq, d);
                if (!same && dist <= bd[KMAX - 1]) {
                    double cd = dist;
                    int32_t ci = (int32_t)j;
                    bool placed = false;
#pragma unroll
                    for (int p = 0; p < KMAX; ++p) {
                        const bool sw = placed || (cd <= bd[p]);
                        const double td = bd[p];
                        const int32_t ti = bi[p];
                        if (sw) {
                            bd[p] = cd;
                            bi[p] = ci;
                            cd = td;
                            ci = ti;
                        }
                        placed = sw;
                    }
#pragma unroll
                    for (int p = 0; p < KMAX; ++p)
                        if (p == k - 1) kth = bd[p];
                }
            }
            if (j0 == 0) done = true;
        }
        if (__syncthreads_and(done ? 1 : 0)) break;
    }
    if (active) {
        int32_t c = 0;
#pragma unroll
        for (int p = 0; p < KMAX; ++p) {
            if (p < k) {
                idx[i * k + p] = bi[p];
                c += (bi[p] >= 0) ? 1 : 0;
            }
        }
        cnt[i] = c;
    }
}

// One G-lane group per target.  LDS image per group: up to G rows with row stride G + 1 doubles - rows 0..c-1 the
// neighbours, row c the target's cross-covariances, row c + 1 the neighbours' function values; c <= G - 2.
template <int G>
__global__ __launch_bounds__(G == 64 ? 64 : 128) void nn_condition_kernel(mgpi_terms_dev t, int nkernel, const double* __restrict__ tx,
                                                           int64_t m, const double* __restrict__ sx,
                                                           const double* __restrict__ sf, const int32_t* __restrict__ idx,
                                                           const int32_t* __restrict__ cnt, int k,
                                                           double* __restrict__ mean, double* __restrict__ diag, double ptol) {
    constexpr int GROUPS = (G == 64 ? 64 : 128) / G;
    constexpr int LD = G + 1;
    __shared__ double A[GROUPS][G * LD];
    __shared__ int32_t nbr[GROUPS][G];
    const int g = threadIdx.x / G, r = threadIdx.x % G;
    const int64_t i = (int64_t)blockIdx.x * GROUPS + g;
    const bool live = i < m;
    const int c = live ? cnt[i] : 0;
    const int d = t.d;
    double* a = A[g];

    if (r < c) nbr[g][r] = idx[i * k + r];
    __syncthreads();

    double xr[MGPI_MAX_D];                      // this lane's point: neighbour r, or the target for the two extra rows
#pragma unroll
    for (int q = 0; q < MGPI_MAX_D; ++q) {
        double v = 0.0;
        if (live && q < d) v = (r < c) ? sx[(int64_t)nbr[g][r] * d + q] : tx[i * d + q];
        xr[q] = v;
    }
    double kss = 0.0, a0 = 0.0;
    if (live) {
        if (r < c) {
            // row r of K(nbrs, nbrs; kernel + nugget), lower part      (:166-169 via :761)
            for (int j = 0; j <= r; ++j) {
                double xj[MGPI_MAX_D];
#pragma unroll
                for (int q = 0; q < MGPI_MAX_D; ++q) xj[q] = (q < d) ? sx[(int64_t)nbr[g][j] * d + q] : 0.0;
                a[r * LD + j] = mgpi_cov_entry_range<0>(t, xr, xj, 0, t.nterms);
            }
            a0 = a[r * LD + r];
        } else if (r == c) {
            // k(x, nbrs; kernel) and k(x, x; kernel)                   (:148, :157)
            for (int j = 0; j < c; ++j) {
                double xj[MGPI_MAX_D];
#pragma unroll
                for (int q = 0; q < MGPI_MAX_D; ++q) xj[q] = (q < d) ? sx[(int64_t)nbr[g][j] * d + q] : 0.0;
                a[r * LD + j] = mgpi_cov_entry_range<0>(t, xr, xj, 0, nkernel);
            }
            kss = mgpi_cov_entry_range<0>(t, xr, xr, 0, nkernel);
        } else if (r == c + 1) {
            for (int j = 0; j < c; ++j) a[r * LD + j] = sf[nbr[g][j]];
        }
    }
    __syncthreads();

    // left-looking Cholesky over the columns; rows c and c + 1 ride along (forward substitution for free)
    for (int j = 0; j < k; ++j) {               // k is block-uniform; groups with fewer neighbours idle through the tail
        if (live && j < c && r == j) {
            double s = a[j * LD + j];
            for (int q = 0; q < j; ++q) s -= a[j * LD + q] * a[j * LD + q];
            // same acceptance rule as the dense factorisation (chol.hip): a pivot not above ptol K_jj is not positive definite
            a[j * LD + j] = (s > ptol * a0) ? sqrt(s) : __builtin_nan("");
        }
        __syncthreads();
        if (live && j < c && r > j && r < c + 2) {
            double s = a[r * LD + j];
            for (int q = 0; q < j; ++q) s -= a[r * LD + q] * a[j * LD + q];
            a[r * LD + j] = s / a[j * LD + j];
        }
        __syncthreads();
    }
    if (live && r == c) {
        if (c == 0) {
            // no neighbours: the prior at this point, kernel then nugget   (:753-758)
            double dg = kss;
            if (nkernel < t.nterms) dg += mgpi_cov_entry_range<0>(t, xr, xr, nkernel, t.nterms);
            mean[i] = 0.0;
            diag[i] = dg;
        } else {
            double mu = 0.0, vv = 0.0;
            for (int j = 0; j < c; ++j) {
                const double v = a[c * LD + j], w = a[(c + 1) * LD + j];
                mu += v * w;
                vv += v * v;
            }
            mean[i] = mu;
            diag[i] = kss - vv;
        }
    }
}

template <int KMAX>
static void launch_search(mgpi_ctx* ctx, const double* sx, const double* srank, int64_t n, int d, const double* tx,
                          const double* trank, int64_t m, int within, int k, int bound_div, int32_t* idx, int32_t* cnt) {
    const int64_t blocks = (m + NN_SLAB - 1) / NN_SLAB;
    hipLaunchKernelGGL(nn_search_kernel<KMAX>, dim3((unsigned)blocks), dim3(NN_SLAB), 0, ctx->stream, sx, srank, n, d, tx,
                       trank, m, within, k, bound_div, idx, cnt);
}

// Neighbour sets (interpolators.py:654-744).  source_x: the reference set ALREADY SORTED by rank (n x d, row-major),
// source_rank its ranks (ascending, checked); target_x / target_rank: the m points to find neighbours for (any order;
// sorted by rank they make the blocks' early exit effective), or NULL for the reference-set mode (each sample against
// the samples in front of it).  rank_is_sum: the rank is sum(x) (order_by_index None) rather than one coordinate -
// it selects the bound the early exit uses.  out_index: m x k, closest first, -1 padded; out_count: m.  Host pointers.
extern "C" int mgpi_nn_neighbors(mgpi_ctx* ctx, const double* source_x, const double* source_rank, int64_t n, int d,
                                 const double* target_x, const double* target_rank, int64_t m, int k, int rank_is_sum,
                                 int32_t* out_index, int32_t* out_count) {
    MGPI_REQUIRE(ctx && source_x && source_rank && out_index && out_count, "mgpi_nn_neighbors: null pointer");
    MGPI_REQUIRE((target_x == nullptr) == (target_rank == nullptr), "mgpi_nn_neighbors: target_x and target_rank go together");
    MGPI_REQUIRE(n >= 1 && d >= 1 && d <= MGPI_MAX_D, "mgpi_nn_neighbors: n=%lld d=%d out of range (d <= %d)", (long long)n, d,
                 MGPI_MAX_D);
    MGPI_REQUIRE(k >= 1 && k <= MGPI_NN_MAX_NEIGHBORS, "mgpi_nn_neighbors: num_neighbors %d outside [1, %d]", k,
                 MGPI_NN_MAX_NEIGHBORS);
    MGPI_REQUIRE(n < (int64_t)2147483647, "mgpi_nn_neighbors: more than 2^31 - 1 samples");
    const int within = target_x == nullptr;
    if (within) m = n;
    MGPI_REQUIRE(m >= 1, "mgpi_nn_neighbors: no targets");
    // every host-side check comes before the first enqueue: an early return must not leave copies from the caller's
    // buffers in flight
    for (int64_t j = 1; j < n; ++j)
        MGPI_REQUIRE(source_rank[j - 1] <= source_rank[j], "mgpi_nn_neighbors: source_rank is not ascending at %lld (sort the "
                     "reference set first, interpolators.py:645-652)", (long long)j);
    MGPI_HIP_TRY(hipSetDevice(ctx->device));
    const size_t sxb = (size_t)n * d * sizeof(double), srb = (size_t)n * sizeof(double);
    const size_t txb = within ? 0 : (size_t)m * d * sizeof(double), trb = within ? 0 : (size_t)m * sizeof(double);
    const size_t ib = (size_t)m * k * sizeof(int32_t), cb = (size_t)m * sizeof(int32_t);
    auto up8 = [](size_t b) { return (b + 255) / 256 * 256; };
    MGPI_TRY(mgpi_ensure_work(ctx, up8(sxb) + up8(srb) + up8(txb) + up8(trb) + up8(ib) + up8(cb)));
    char* w = (char*)ctx->work;
    double* dsx = (double*)w;            w += up8(sxb);
    double* dsr = (double*)w;            w += up8(srb);
    double* dtx = within ? dsx : (double*)w;  w += up8(txb);
    double* dtr = within ? dsr : (double*)w;  w += up8(trb);
    int32_t* didx = (int32_t*)w;         w += up8(ib);
    int32_t* dcnt = (int32_t*)w;
    MGPI_HIP_TRY(hipMemcpyAsync(dsx, source_x, sxb, hipMemcpyHostToDevice, ctx->stream));
    MGPI_HIP_TRY(hipMemcpyAsync(dsr, source_rank, srb, hipMemcpyHostToDevice, ctx->stream));
    if (!within) {
        MGPI_HIP_TRY(hipMemcpyAsync(dtx, target_x, txb, hipMemcpyHostToDevice, ctx->stream));
        MGPI_HIP_TRY(hipMemcpyAsync(dtr, target_rank, trb, hipMemcpyHostToDevice, ctx->stream));
    }
    const int bound_div = rank_is_sum ? d : 1;     // (rank_t - rank_j)^2 <= bound_div * dist
    if (k <= 8) launch_search<8>(ctx, dsx, dsr, n, d, dtx, dtr, m, within, k, bound_div, didx, dcnt);
    else if (k <= 16) launch_search<16>(ctx, dsx, dsr, n, d, dtx, dtr, m, within, k, bound_div, didx, dcnt);
    else if (k <= 32) launch_search<32>(ctx, dsx, dsr, n, d, dtx, dtr, m, within, k, bound_div, didx, dcnt);
    else launch_search<MGPI_NN_MAX_NEIGHBORS>(ctx, dsx, dsr, n, d, dtx, dtr, m, within, k, bound_div, didx, dcnt);
    MGPI_HIP_TRY(hipGetLastError());
    MGPI_HIP_TRY(hipMemcpyAsync(out_index, didx, ib, hipMemcpyDeviceToHost, ctx->stream));
    MGPI_HIP_TRY(hipMemcpyAsync(out_count, dcnt, cb, hipMemcpyDeviceToHost, ctx->stream));
    MGPI_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}

// Per-point conditioning on the neighbour sets (interpolators.py:751-792): mean[i], diag[i] of target i given the
// samples index[i, 0:count[i]] of (source_x, source_f).  terms = the nkernel summands of the kernel followed by the
// nugget's (the order of kernel + nugget, kernels.py:54-59).  A neighbour matrix that is not positive definite
// yields NaN for that point (the host raises, like the reference's `safe` checks :766-772).  Host pointers.
extern "C" int mgpi_nn_condition(mgpi_ctx* ctx, const double* target_x, int64_t m, const double* source_x,
                                 const double* source_f, int64_t n, int d, const int32_t* index, const int32_t* count, int k,
                                 const mgpi_kernel_desc* terms, int nterms, int nkernel, double* out_mean, double* out_diag) {
    MGPI_REQUIRE(ctx && target_x && source_x && source_f && index && count && out_mean && out_diag,
                 "mgpi_nn_condition: null pointer");
    MGPI_REQUIRE(m >= 1 && n >= 1, "mgpi_nn_condition: empty input");
    MGPI_REQUIRE(k >= 1 && k <= MGPI_NN_MAX_NEIGHBORS, "mgpi_nn_condition: num_neighbors %d outside [1, %d]", k,
                 MGPI_NN_MAX_NEIGHBORS);
    MGPI_REQUIRE(nkernel >= 1 && nkernel <= nterms, "mgpi_nn_condition: nkernel %d outside [1, nterms = %d]", nkernel, nterms);
    MGPI_REQUIRE(n < (int64_t)2147483647, "mgpi_nn_condition: more than 2^31 - 1 samples");
    for (int64_t i = 0; i < m; ++i) {
        MGPI_REQUIRE(count[i] >= 0 && count[i] <= k, "mgpi_nn_condition: count[%lld] = %d outside [0, %d]", (long long)i,
                     count[i], k);
        for (int p = 0; p < count[i]; ++p)
            MGPI_REQUIRE(index[i * k + p] >= 0 && index[i * k + p] < n, "mgpi_nn_condition: index[%lld, %d] = %d outside [0, %lld)",
                         (long long)i, p, index[i * k + p], (long long)n);
    }
    mgpi_terms_dev t;
    MGPI_TRY(mgpi_make_terms(terms, nterms, d, &t));
    MGPI_HIP_TRY(hipSetDevice(ctx->device));
    const size_t txb = (size_t)m * d * sizeof(double), sxb = (size_t)n * d * sizeof(double), sfb = (size_t)n * sizeof(double);
    const size_t ib = (size_t)m * k * sizeof(int32_t), cb = (size_t)m * sizeof(int32_t), ob = (size_t)m * sizeof(double);
    auto up8 = [](size_t b) { return (b + 255) / 256 * 256; };
    MGPI_TRY(mgpi_ensure_work(ctx, up8(txb) + up8(sxb) + up8(sfb) + up8(ib) + up8(cb) + 2 * up8(ob)));
    char* w = (char*)ctx->work;
    double* dtx = (double*)w;            w += up8(txb);
    double* dsx = (double*)w;            w += up8(sxb);
    double* dsf = (double*)w;            w += up8(sfb);
    int32_t* didx = (int32_t*)w;         w += up8(ib);
    int32_t* dcnt = (int32_t*)w;         w += up8(cb);
    double* dmean = (double*)w;          w += up8(ob);
    double* ddiag = (double*)w;
    MGPI_HIP_TRY(hipMemcpyAsync(dtx, target_x, txb, hipMemcpyHostToDevice, ctx->stream));
    MGPI_HIP_TRY(hipMemcpyAsync(dsx, source_x, sxb, hipMemcpyHostToDevice, ctx->stream));
    MGPI_HIP_TRY(hipMemcpyAsync(dsf, source_f, sfb, hipMemcpyHostToDevice, ctx->stream));
    MGPI_HIP_TRY(hipMemcpyAsync(didx, index, ib, hipMemcpyHostToDevice, ctx->stream));
    MGPI_HIP_TRY(hipMemcpyAsync(dcnt, count, cb, hipMemcpyHostToDevice, ctx->stream));
    if (k <= 14) {
        hipLaunchKernelGGL(nn_condition_kernel<16>, dim3((unsigned)((m + 7) / 8)), dim3(128), 0, ctx->stream, t, nkernel, dtx, m,
                           dsx, dsf, didx, dcnt, k, dmean, ddiag, ctx->pivot_tol);
    } else if (k <= 30) {
        hipLaunchKernelGGL(nn_condition_kernel<32>, dim3((unsigned)((m + 3) / 4)), dim3(128), 0, ctx->stream, t, nkernel, dtx, m,
                           dsx, dsf, didx, dcnt, k, dmean, ddiag, ctx->pivot_tol);
    } else {
        hipLaunchKernelGGL(nn_condition_kernel<64>, dim3((unsigned)m), dim3(64), 0, ctx->stream, t, nkernel, dtx, m,
                           dsx, dsf, didx, dcnt, k, dmean, ddiag, ctx->pivot_tol);
    }
    MGPI_HIP_TRY(hipGetLastError());
    MGPI_HIP_TRY(hipMemcpyAsync(out_mean, dmean, ob, hipMemcpyDeviceToHost, ctx->stream));
    MGPI_HIP_TRY(hipMemcpyAsync(out_diag, ddiag, ob, hipMemcpyDeviceToHost, ctx->stream));
    MGPI_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return 0;
}
