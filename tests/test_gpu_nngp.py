"""NearestNeighborInterpolator on the GPU (csrc/nngp.hip) against the reference's golden vectors and the oracle.
Needs an MI355X: run with -m gpu.

Tolerances: neighbour index lists are integer work - bit-exact whenever no two candidate distances are equal (the
tie-free fixtures; on the regular-grid fixture the reference's numpy argsort breaks ties in an unspecified order, there
the DISTANCES of the chosen neighbours must agree and the sets must equal the oracle's under the product's stable
rule).  Means: max-norm-relative 1e-10 while cond <= 1e7 (conftest.mean_tol; cond bounded per fixture); variances: 1e-8 of the prior variance.  Loglikelihood:
its terms (mean - f)**2 / var divide by conditional variances 1e6 below the prior variance when the nugget is small,
and the reference's LU inverse loses ~cond * eps in each of them: its own value is off by up to 3e-8 relative from an
80-bit evaluation (stored in the fixtures as loglike_longdouble), a fp64 Cholesky by 1e-12.  So: rtol 1e-11 against
the extended-precision value, and no further from the reference than the reference is from that value."""
import numpy as np
import pytest

import mgpi_amd as mgpi
from conftest import load_golden, mean_tol, neighbor_lists, nngp_names, oracle_terms, prior_var, product_kernel, relmax
from oracle import mgpi_oracle as oracle

pytestmark = pytest.mark.gpu
EPS = 2.220446049250313e-16


def make(g):
    k, obi = int(g['k']), (None if int(g['obi']) < 0 else int(g['obi']))
    return mgpi.NearestNeighborInterpolator(product_kernel(g, 'kernel'), nugget=product_kernel(g, 'nugget'),
                                            num_neighbors=k, order_by_index=obi), k, obi


@pytest.mark.parametrize('name', nngp_names())
def test_nngp_golden(name):
    g = load_golden(name)
    nn, k, obi = make(g)
    sx, sf = nn._2sorted(g['x'], source_f=g['f'])
    assert np.array_equal(sx, g['sorted_x']) and np.array_equal(sf, g['sorted_f'])

    nb_self = nn._2neighbors(sx)
    nb_tar = nn._2neighbors(sx, target_x=g['xs'])
    assert len(nb_self) == len(sx) and len(nb_tar) == len(g['xs'])
    for got, want, pts in [(nb_self, neighbor_lists(g['nb_self']), sx), (nb_tar, neighbor_lists(g['nb_target']), g['xs'])]:
        if bool(g['tie_free']):
            for i, (a, b) in enumerate(zip(got, want)):
                assert np.array_equal(a, b), (i, a, b)
        else:
            for a, b, p in zip(got, want, pts):       # same distances, whatever the order inside a tie
                da, db = np.sum((sx[a] - p)**2, axis=1), np.sum((sx[b] - p)**2, axis=1)
                assert np.array_equal(da, db)
    if not bool(g['tie_free']):
        stable = oracle.nn_neighbors(sx, g['xs'], k, obi, stable=True)
        assert all(np.array_equal(a, b) for a, b in zip(nb_tar, stable))
        stable = oracle.nn_neighbors(sx, None, k, obi, stable=True)
        assert all(np.array_equal(a, b) for a, b in zip(nb_self, stable))

    pv = prior_var(g)
    # cond of a k x k neighbour matrix is at most (k * sum sigma_SE^2 + sigma_white^2) / sigma_white^2; the fixtures carry
    # white noise of sigma/1000 -> ~1e7, the edge of the 1e-10 contract (conftest.mean_tol)
    white = sum(float(s)**2 for pre in ('kernel', 'nugget') for t, s in zip(g[pre + '_types'], g[pre + '_sigmas']) if t == 1)
    tol = mean_tol((k * pv + white) / white)
    # per-point conditioning on the REFERENCE's neighbour sets: isolates the arithmetic from the tie rule
    mean, diag = nn._2diag(sx, sx, sf, neighbor_lists(g['nb_self']))
    assert relmax(mean, g['self_mean']) < tol
    assert np.max(np.abs(diag - g['self_diag'])) < 1e-8 * pv
    mean, diag = nn._2diag(g['xs'], sx, sf, neighbor_lists(g['nb_target']))
    assert relmax(mean, g['cond_mean']) < tol
    assert np.max(np.abs(diag - g['cond_diag'])) < 1e-8 * pv

    if bool(g['tie_free']):      # end to end through the public methods
        mean, cov = nn.condition(g['xs'], g['x'], g['f'])
        assert cov.shape == (len(mean), len(mean)) and np.count_nonzero(cov - np.diag(np.diag(cov))) == 0
        assert relmax(mean, g['cond_mean']) < tol
        assert np.max(np.abs(np.diag(cov) - g['cond_diag'])) < 1e-8 * pv
        m2, d2 = nn.condition_diag(g['xs'], g['x'], g['f'])
        assert np.array_equal(m2, mean) and np.array_equal(d2, np.diag(cov))
        ll = nn.loglikelihood(g['x'], g['f'])
        # the yardstick is the 80-bit Cholesky value stored beside the reference's (make_fixtures.py): the product has
        # to hit THAT to 1e-11, and may be as far from the reference's LU-inverse value as the reference itself is
        exact, ref = float(g['loglike_longdouble']), float(g['loglike'])
        assert abs(ll - exact) < 1e-11 * abs(exact) + 1e-12
        assert abs(ll - ref) <= 1.01 * abs(ref - exact) + 1e-10 * abs(ref)
        assert nn.loglikelihood(sx, sf, neighbors=nb_self) == ll          # pre-sorted entry (uiuc-test-nngp-timing:230)
        assert abs(nn.loglikelihood(sx, sf, neighbors=neighbor_lists(g['nb_self'])) - ll) <= 1e-12 * abs(ll)


@pytest.mark.parametrize('n,m,d,k,obi,seed', [(3000, 700, 2, 10, None, 1), (2500, 300, 3, 20, 1, 2), (1500, 1000, 5, 8, None, 3),
                                              (700, 50, 16, 33, 0, 4), (257, 129, 1, 4, None, 5)])
def test_nngp_seeded_against_oracle(n, m, d, k, obi, seed):
    rng = np.random.default_rng(seed)
    x = rng.random((n, d))
    f = np.sin(3 * x[:, 0]) + np.cos(2 * x[:, -1]) + 0.1 * x.sum(axis=1)
    xs = rng.random((m, d))
    ell = (0.6 * d**0.5 * n**(-1.0 / d) * 3,) * d
    kern = mgpi.SquaredExponentialKernel(1.0, *ell)
    nug = mgpi.WhiteNoiseKernel(1e-2)
    nn = mgpi.NearestNeighborInterpolator(kern, nugget=nug, num_neighbors=k, order_by_index=obi)
    kt, nt = [oracle.term_se(1.0, *ell)], [oracle.term_white(1e-2)]

    sx, sf = nn._2sorted(x, source_f=f)
    osx, osf = oracle.nn_sorted(x, f, obi)
    assert np.array_equal(sx, osx) and np.array_equal(sf, osf)
    assert np.array_equal(nn._2ranks(x), oracle.nn_ranks(x, obi))
    nb = nn._2neighbors(sx, target_x=xs)
    want = oracle.nn_neighbors(osx, xs, k, obi, stable=True)
    assert all(np.array_equal(a, b) for a, b in zip(nb, want))
    sub = slice(0, 400)                       # the reference-set loop of the oracle is O(N^2) in Python
    nbs = nn._2neighbors(sx[sub])
    want = oracle.nn_neighbors(osx[sub], None, k, obi, stable=True)
    assert all(np.array_equal(a, b) for a, b in zip(nbs, want))

    mean, diag = nn.condition_diag(xs[:200], x, f)
    omean, ocov = oracle.nn_condition(kt, nt, xs[:200], x, f, k, obi, stable=True)
    assert relmax(mean, omean) < 1e-10
    assert np.max(np.abs(diag - np.diag(ocov))) < 1e-8


def test_nngp_api_surface_and_errors():
    kern = mgpi.SquaredExponentialKernel(1.0, 0.3, 0.3)
    nn = mgpi.NearestNeighborInterpolator(kern, nugget=mgpi.WhiteNoiseKernel(1e-2))
    assert nn.num_neighbors == mgpi.DEFAULT_NUM_NEIGHBORS == 10 and nn.order_by_index is mgpi.DEFAULT_ORDER_BY_INDEX is None
    assert mgpi.factory(mgpi.Interpolator)['NearestNeighborInterpolator'] is mgpi.NearestNeighborInterpolator
    rng = np.random.default_rng(0)
    x, f = rng.random((50, 2)), rng.random(50)
    with pytest.raises(NotImplementedError):
        nn.compress(x, f)
    with pytest.raises(NotImplementedError):
        nn.predict(x, x, f)
    with pytest.raises(AssertionError):
        nn._2neighbors(x, target_x=rng.random((3, 3)))
    with pytest.raises(ValueError):
        mgpi.NearestNeighborInterpolator(kern, num_neighbors=63)._2neighbors(x)
    # a single point against an explicit neighbour set = the dense conditional on that set
    m1, d1 = nn._sample2diag(x[[0]], x[1:9], f[1:9])
    m2, c2 = mgpi.Interpolator(kern, nugget=nn.nugget).condition(x[[0]], x[1:9], f[1:9])
    assert abs(m1 - m2[0]) < 1e-12 and abs(d1 - c2[0, 0]) < 1e-12
    m0, d0 = nn._sample2diag(x[[0]], x[:0], f[:0])
    assert m0 == 0.0 and d0 == 1.0 + 1e-4
    # duplicated neighbours under an exact-equality nugget: the neighbour matrix is singular -> the `safe` checks fire
    dup = np.vstack([x[:5], x[:5], [[2.0, 2.0]]])
    with pytest.raises(AssertionError):
        mgpi.NearestNeighborInterpolator(kern, num_neighbors=10).loglikelihood(dup, np.arange(11.0))


def test_nngp_rvs_draws_from_the_diagonal_conditional():
    rng = np.random.default_rng(4)
    x = rng.random((400, 2))
    f = np.sin(4 * x[:, 0]) + x[:, 1]
    xs = rng.random((6, 2))
    nn = mgpi.NearestNeighborInterpolator(mgpi.SquaredExponentialKernel(1.0, 0.2, 0.2), nugget=mgpi.WhiteNoiseKernel(5e-2))
    mean, var = nn.condition_diag(xs, x, f)
    draws = nn.rvs(xs, x, f, size=20000, rng=np.random.default_rng(9))
    assert draws.shape == (20000, 6) and nn.sample.__func__ is nn.rvs.__func__
    assert np.all(np.abs(draws.mean(axis=0) - mean) < 5 * np.sqrt(var / 20000))
    assert np.all(np.abs(draws.var(axis=0) / var - 1) < 0.05)
    assert np.max(np.abs(np.corrcoef(draws.T) - np.eye(6))) < 0.04


def test_nngp_hyperparameter_logprob_reuses_neighbors():
    rng = np.random.default_rng(8)
    x = rng.random((600, 2))
    f = np.sin(5 * x[:, 0]) * x[:, 1]
    nn = mgpi.NearestNeighborInterpolator(mgpi.SquaredExponentialKernel(1.0, 0.2, 0.2), nugget=mgpi.WhiteNoiseKernel(1e-2))
    logprob = nn._construct_logprob(x, f)
    assert isinstance(logprob.kwargs['neighbors'], mgpi.NeighborSets)
    a = logprob([0.9, 0.25, 0.15])
    assert list(nn.kernel.params) == [0.9, 0.25, 0.15]
    assert abs(a - nn.loglikelihood(x, f)) < 1e-9 * abs(a)
    assert logprob([-1.0, 0.2, 0.2]) == -np.inf


def test_nngp_large_reference_set_properties():
    """N = 60000 (3.6e9 candidate pairs): size-independent properties - every neighbour precedes its point, lists are
    sorted by distance, nobody is its own neighbour, k neighbours once k predecessors exist; a brute-force numpy check
    on a sample of rows; the loglikelihood is finite"""
    rng = np.random.default_rng(12)
    n, k = 60000, 10
    x = rng.random((n, 2))
    f = np.sin(6 * x[:, 0]) * np.cos(4 * x[:, 1])
    nn = mgpi.NearestNeighborInterpolator(mgpi.SquaredExponentialKernel(1.0, 0.02, 0.02), nugget=mgpi.WhiteNoiseKernel(1e-2),
                                          num_neighbors=k, order_by_index=0)
    sx, sf = nn._2sorted(x, source_f=f)
    nb = nn._2neighbors(sx)
    assert np.array_equal(nb.count, np.minimum(np.arange(n), k))
    rows = np.arange(n)[:, None]
    valid = nb.index >= 0
    assert np.all(nb.index[valid] < np.broadcast_to(rows, nb.index.shape)[valid])
    dist = np.where(valid, np.sum((sx[np.maximum(nb.index, 0)] - sx[:, None, :])**2, axis=2), np.inf)
    with np.errstate(invalid='ignore'):
        assert np.all(np.diff(dist, axis=1)[valid[:, 1:]] >= 0)
    for i in rng.integers(k + 1, n, 25):
        d2 = np.sum((sx[:i] - sx[i])**2, axis=1)
        assert np.array_equal(nb[i], np.argsort(d2, kind='stable')[:k])
    ll = nn.loglikelihood(sx, sf, neighbors=nb)
    assert np.isfinite(ll)
