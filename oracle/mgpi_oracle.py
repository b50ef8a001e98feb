"""CPU oracle for the mgpi dense-GP hot path.  TEST INFRASTRUCTURE ONLY.

This file is a numpy restatement of the reference algorithm
(reedessick/muses-gaussian-process-interpolator @ 2024_08_07).  It is the
*checker* for the HIP path: only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.  Nothing under the product
package (``muses-gaussian-process-interpolator_amd/``) imports or calls it, and
the product has no CPU fallback.

Parity status: PINNED.  ``tests/golden/make_fixtures.py`` imports the real
reference in the build container and stores its outputs as ``tests/golden/*.npz``;
``tests/test_oracle_vs_golden.py`` checks every function below against them.
(The reference's own test scripts hold no numeric fixtures - SURVEY.md section 4.)

Each function cites the reference file:line whose arithmetic it restates.  The
arithmetic is kept operation-for-operation (same association order, LU inverse
via ``np.linalg.inv`` rather than Cholesky, exact-equality white noise, no
factor 1/2 in the squared exponential) so that it reproduces the reference's
rounding behaviour, not just its mathematics.
"""

import numpy as np

# kernel term type codes: the same integers as include/mgpi_hip.h
TERM_SE = 0
TERM_WHITE = 1
TERM_MATERN = 2

#-------------------------------------------------
# kernel terms: a kernel is a list of (type, sigma, lengths) tuples that are summed
# (mgpi/kernels.py:287-293 CombinedKernel.cov sums the member covariances in order)


def term_se(sigma, *lengths):
    return (TERM_SE, float(sigma), np.asarray(lengths, dtype=float))


def term_white(sigma):
    return (TERM_WHITE, float(sigma), np.zeros(0, dtype=float))


def term_matern(order, sigma, *lengths):
    """(type, sigma, lengths, order): the argument order of MaternKernel(order, sigma, *lengths), kernels.py:139"""
    return (TERM_MATERN, float(sigma), np.asarray(lengths, dtype=float), float(order))


# The reference's Matern is 0**o * K_o(0) = 0 * inf = NaN at zero separation, which makes
# every K(x,x) unusable.  With MATERN_ZERO_LIMIT = True the oracle substitutes the r -> 0 limit sigma**2 there -
# the product's documented deviation - so that full solves with a Matern term can be cross-checked; every entry at
# non-zero separation is the reference's formula either way.  Tests set this flag explicitly when they need it.
MATERN_ZERO_LIMIT = False


def paired_cov(terms, x1, x2):
    """row-paired covariance, x1 and x2 both (P, D) -> (P,)

    SE:     mgpi/kernels.py:180-185   sigma**2 * exp(-sum((x1-x2)**2/lengths**2, axis=1))   (no 1/2)
    white:  mgpi/kernels.py:128-131   sigma**2 * all(x1==x2, axis=1)
    Matern: mgpi/kernels.py:155-162   diff = (2o)**0.5 * sum(...)**0.5; s**2 * (2**(1-o)/gamma(o)) * diff**o * kv(o, diff)
    sum:    mgpi/kernels.py:287-293   ans = 0.0; ans += term   (left to right)
    """
    x1 = np.asarray(x1, dtype=float)
    x2 = np.asarray(x2, dtype=float)
    total = 0.0
    for term in terms:
        kind, sigma, lengths = term[0], term[1], term[2]
        if kind == TERM_MATERN:
            from scipy.special import gamma as _gamma, kv as _bessel_k      # kernels.py:15-16
            o = term[3]
            with np.errstate(invalid='ignore', over='ignore'):
                diff = (2 * o)**0.5 * np.sum((x1 - x2)**2 / lengths**2, axis=1)**0.5
                val = sigma**2 * (2**(1 - o) / _gamma(o)) * diff**o * _bessel_k(o, diff)
            if MATERN_ZERO_LIMIT:
                val = np.where(diff == 0, sigma**2, val)
            total = total + val
        elif kind == TERM_SE:
            total = total + sigma**2 * np.exp(-np.sum((x1 - x2)**2 / lengths**2, axis=1))
        elif kind == TERM_WHITE:
            total = total + sigma**2 * np.all(x1 == x2, axis=1)
        else:
            raise ValueError('unknown term type %r' % (kind,))
    return total


def x2cov(terms, x1, x2):
    """pairwise matrix K[i,j] = cov(x1[i], x2[j]); (n1,D),(n2,D) -> (n1,n2) row-major

    mgpi/interpolators.py:208-300: materialise the n1*n2 row pairs (x1 rows each repeated n2
    times, x2 tiled n1 times), evaluate the paired covariance, reshape.  The materialisation is
    kept (np.repeat / np.tile) because it is what the reference spends its time on
     and so belongs in the timed CPU baseline.
    """
    x1 = np.asarray(x1, dtype=float)
    x2 = np.asarray(x2, dtype=float)
    n1, n2 = len(x1), len(x2)
    assert x1.shape[1] == x2.shape[1], 'conflicting number of dimensions for x1 and x2'
    big1 = np.repeat(x1, n2, axis=0)      # interpolators.py:247-249
    big2 = np.tile(x2, (n1, 1))           # interpolators.py:280
    return paired_cov(terms, big1, big2).reshape((n1, n2))   # interpolators.py:293-294


def x2cov_chunked(terms, x1, x2, rows=256, workers=1):
    """same values as x2cov, built ``rows`` rows of x1 at a time (memory-bounded variant for
    N >= 16384 where the reference's temporaries no longer fit).
    workers > 1: the row chunks are evaluated by a thread pool (numpy releases the GIL inside its
    ufuncs); every entry is computed by the same expression, so the result is bit-identical."""
    x1 = np.asarray(x1, dtype=float)
    x2 = np.asarray(x2, dtype=float)
    out = np.empty((len(x1), len(x2)), dtype=float)

    def chunk(lo):
        out[lo:lo + rows] = x2cov(terms, x1[lo:lo + rows], x2)

    starts = range(0, len(x1), rows)
    if workers <= 1:
        for lo in starts:
            chunk(lo)
    else:
        from concurrent.futures import ThreadPoolExecutor
        with ThreadPoolExecutor(max_workers=int(workers)) as pool:
            list(pool.map(chunk, starts))
    return out

#-------------------------------------------------
# Interpolator methods


def compress(kernel, nugget, source_x, source_f, timings=None):
    """alpha = inv(K(x,x; kernel+nugget)) @ f        mgpi/interpolators.py:72-105
    timings: a dict that receives the seconds of the two stages the reference's verbose mode reports
    (:78-102: "computing covariance", "inverting"); same arithmetic either way."""
    import time
    terms = list(kernel) + (list(nugget) if nugget else [])   # kernel+nugget, kernels.py:54-59
    t0 = time.perf_counter()
    cov_src_src = x2cov(terms, source_x, source_x)            # :81-84
    t1 = time.perf_counter()
    inv = np.linalg.inv(cov_src_src)                          # :92
    t2 = time.perf_counter()
    if timings is not None:
        timings['x2cov'] = t1 - t0
        timings['inv'] = t2 - t1
    return inv @ np.asarray(source_f, dtype=float)            # :100


def predict(kernel, target_x, source_x, compressed):
    """mean = K(x*,x; kernel) @ alpha  (kernel WITHOUT nugget)   mgpi/interpolators.py:109-130"""
    return x2cov(list(kernel), target_x, source_x) @ compressed            # :117, :125


def condition(kernel, nugget, target_x, source_x, source_f):
    """(mean, cov) of the conditioned process   mgpi/interpolators.py:136-204 (R&W eq 2.19)

    K**, K* use ``kernel``; K uses ``kernel+nugget`` (:148,:157,:166-169); LU inverse (:179);
    left-associated products (:189, :199).
    """
    kterms = list(kernel)
    sterms = kterms + (list(nugget) if nugget else [])
    cov_tar_tar = x2cov(kterms, target_x, target_x)
    cov_tar_src = x2cov(kterms, target_x, source_x)
    cov_src_src = x2cov(sterms, source_x, source_x)
    inv_cov_src_src = np.linalg.inv(cov_src_src)
    mean = cov_tar_src @ inv_cov_src_src @ np.asarray(source_f, dtype=float)
    cov = cov_tar_tar - cov_tar_src @ inv_cov_src_src @ np.transpose(cov_tar_src)
    return mean, cov


def loglikelihood(kernel, nugget, source_x, source_f):
    """-f K^-1 f/2 - logdet K/2 - N log(2 pi)/2     mgpi/interpolators.py:331-343"""
    terms = list(kernel) + (list(nugget) if nugget else [])
    cov_src_src = x2cov(terms, source_x, source_x)
    s, logdet = np.linalg.slogdet(cov_src_src)                # :339
    assert s > 0, 'covariance is not positive definite!'      # :340
    f = np.asarray(source_f, dtype=float)
    return -0.5 * f @ np.linalg.inv(cov_src_src) @ f - 0.5 * logdet - 0.5 * len(f) * np.log(2 * np.pi)


def rvs_from_conditioned(mean, cov, size=1, rng=None):
    """mean + MVN(0, cov/outer(s,s))*s with s = sqrt(diag cov)   mgpi/interpolators.py:314-324
    (the reference's own call at :310 raises TypeError - SURVEY.md section 0 - so no golden output
    exists for this function: parity unpinned, statistical tests only)"""
    rng = np.random.default_rng() if rng is None else rng
    scales = np.diag(cov)**0.5
    zeros = np.zeros_like(mean, dtype=float)
    return mean + rng.multivariate_normal(zeros, cov / np.outer(scales, scales), size=size) * scales

def rvs_cholesky(mean, cov, z):
    """the same documented intent (mgpi/interpolators.py:314-324: ``mean + MVN(0, cov/outer(s,s)) * s``,
    s = sqrt(diag cov)) with the normal deviates GIVEN: a sample of MVN(0, C) is chol(C) z, so with z (size, M)
    the draws are  mean + (chol(cov / outer(s, s)) @ z.T).T * s.  A deterministic statement of the arithmetic the
    product's device sampler performs (the reference's own call raises TypeError, SURVEY.md section 0, and numpy's
    SVD-based multivariate_normal consumes its deviates differently) - jitter-free case only."""
    mean = np.asarray(mean, dtype=float)
    cov = np.asarray(cov, dtype=float)
    scales = np.diag(cov)**0.5
    chol = np.linalg.cholesky(cov / np.outer(scales, scales))
    return mean + (chol @ np.asarray(z, dtype=float).T).T * scales

#-------------------------------------------------
# large-N helpers (used where the LU-inverse statement above cannot run: N >= 16384)


def cholesky_lower(K):
    """LAPACK dpotrf via numpy; the mathematical restatement north_star assigns to the solve
    (SURVEY.md section 0 row 3).  Only used for residual / tile-level checks."""
    return np.linalg.cholesky(K)


def solve_cholesky(L, b):
    from scipy.linalg import solve_triangular
    y = solve_triangular(L, b, lower=True)
    return solve_triangular(L, y, lower=True, trans='T')


def diag_variance_cholesky(kernel, L, target_x, source_x, chunk=1024):
    """var_i = K**_ii - || L^-1 K*_i^T ||^2 : the diagonal of interpolators.py:199"""
    from scipy.linalg import solve_triangular
    kterms = list(kernel)
    target_x = np.asarray(target_x, dtype=float)
    out = np.empty(len(target_x))
    for lo in range(0, len(target_x), chunk):
        t = target_x[lo:lo + chunk]
        ks = x2cov_chunked(kterms, t, source_x)
        v = solve_triangular(L, ks.T, lower=True)
        out[lo:lo + chunk] = paired_cov(kterms, t, t) - np.sum(v * v, axis=0)
    return out

#-------------------------------------------------
# NearestNeighborInterpolator (NNGP, Datta et al. 2016): mgpi/interpolators.py:612-871
# restated without the progress printing; every selection rule and its quirks are kept:
#   * samples are ordered by sum(x) (or one coordinate), and a point's neighbours are the num_neighbors closest
#     samples - plain Euclidean distance, NOT scaled by the kernel lengths - among those ranked at or below it;
#   * samples that coincide exactly with the point are never its neighbours, so an NNGP prediction AT a training
#     point does not use that point;
#   * numpy's default argsort decides ties in distance (unspecified order).  ``stable=True`` is the product's rule
#     (ties go to the lower index in the sorted reference set); the two agree whenever distances are distinct.


def nn_rank(x, order_by_index=None):
    """:633-640"""
    return np.sum(x) if order_by_index is None else x[order_by_index]


def nn_ranks(x, order_by_index=None):
    """:642-643"""
    return np.array([nn_rank(row, order_by_index) for row in x])


def nn_sorted(source_x, source_f=None, order_by_index=None):
    """:645-652"""
    order = np.argsort(nn_ranks(source_x, order_by_index))
    return source_x[order], (source_f[order] if source_f is not None else None)


def nn_neighbors(source_x, target_x=None, num_neighbors=10, order_by_index=None, stable=False):
    """list of index arrays into the SORTED reference set, one per target, closest first   (:654-744)"""
    source_x = np.asarray(source_x, dtype=float)
    within = target_x is None
    targets = source_x if within else np.asarray(target_x, dtype=float)
    ranks = nn_ranks(source_x, order_by_index)
    every = np.arange(len(source_x))
    visible = 0 if within else len(source_x)        # :663-670 samples [0, visible) may be neighbours
    out = []
    for x in targets:
        ok = np.zeros(len(source_x), dtype=bool)
        ok[:visible] = ranks[:visible] <= nn_rank(x, order_by_index)          # :695-696
        ok[every[ok][np.all(source_x[ok] == x, axis=1)]] = False               # :706-718 exact matches dropped
        if ok.any():
            dist = np.sum((source_x[ok] - x)**2, axis=1)                       # :725
            order = np.argsort(dist, kind='stable') if stable else np.argsort(dist)
            out.append(every[ok][order][:num_neighbors])                       # :730
        else:
            out.append([])                                                     # :741
        visible += 1                                                           # :744
    return out


def nn_sample2diag(kernel, nugget, x, ref_x, ref_f):
    """(mean, variance) of one point given its neighbour set   (:751-776)"""
    if len(ref_x) == 0:
        diag = paired_cov(list(kernel), x, x)[0]
        if nugget:
            diag += paired_cov(list(nugget), x, x)[0]
        return 0.0, diag
    m, c = condition(kernel, nugget, x, ref_x, ref_f)
    mean, diag = m[0], c[0, 0]
    assert mean == mean, 'mean is nan'
    assert diag == diag, 'diag is nan'
    assert diag > 0, 'marginal variance is negative!'
    return mean, diag


def nn_2diag(kernel, nugget, target_x, source_x, source_f, neighbors):
    """(2, M): means and variances   (:780-792)"""
    source_x = np.asarray(source_x, dtype=float)
    return np.transpose([nn_sample2diag(kernel, nugget, target_x[[i]], source_x[np.asarray(neighbors[i], dtype=int)],
                                        source_f[np.asarray(neighbors[i], dtype=int)]) for i in range(len(target_x))])


def nn_loglikelihood(kernel, nugget, source_x, source_f, num_neighbors=10, order_by_index=None, neighbors=None, stable=False):
    """sum of independent 1-D Gaussians   (:833-846)"""
    if neighbors is None:
        source_x, source_f = nn_sorted(source_x, source_f, order_by_index)
        neighbors = nn_neighbors(source_x, None, num_neighbors, order_by_index, stable)
    mean, diag = nn_2diag(kernel, nugget, source_x, source_x, source_f, neighbors)
    return -0.5 * np.sum((mean - source_f)**2 / diag) - 0.5 * np.sum(np.log(diag)) - 0.5 * len(source_x) * np.log(2 * np.pi)


def nn_condition(kernel, nugget, target_x, source_x, source_f, num_neighbors=10, order_by_index=None, stable=False):
    """(mean, diagonal covariance matrix)   (:850-884)"""
    source_x, source_f = nn_sorted(source_x, source_f, order_by_index)
    neighbors = nn_neighbors(source_x, target_x, num_neighbors, order_by_index, stable)
    mean, diag = nn_2diag(kernel, nugget, np.asarray(target_x, dtype=float), source_x, source_f, neighbors)
    return mean, np.diag(diag)
