"""Parity of the HIP path (through the C-ABI, via the Python host) against the golden vectors of the real
reference and against the oracle on seeded inputs.  Needs an MI355X: run with -m gpu.

Stated tolerances (fp64; SURVEY.md section 8c, conftest.mean_tol / var_tol):
  K entries           <= 4 ulp per summand, everywhere (the exp argument is bit-identical to the reference's)
  mean / predict      max-norm-relative <= 1e-10 for cond(K) <= 1e7, 4e-17*cond beyond
  variance / cov      <= 1e-8 * prior variance for cond(K) <= 1e7
  alpha               <= 100 * cond * eps (the reference's LU inverse is itself only this reproducible)
  loglikelihood       rtol 1e-10 (cond-scaled like the mean)
"""
import numpy as np
import pytest

import mgpi_amd as mgpi
from conftest import golden_names, load_golden, matern_names, matern_terms, mean_tol, oracle_terms, prior_var, product_kernel, relmax, var_tol
from oracle import mgpi_oracle as oracle

pytestmark = pytest.mark.gpu

NAMES = golden_names()
EPS = 2.220446049250313e-16


def ulps(a, ref):
    """|a - ref| in units of the last place of ref (np.spacing handles the subnormal range)"""
    return np.abs(np.asarray(a) - np.asarray(ref)) / np.spacing(np.abs(np.asarray(ref, dtype=float)) + 0.0)


def assert_K_close(K, ref, nterms=1):
    """K entries to <= 4 ulp.  The device forms (x1-x2)**2/length**2 correctly rounded (the same
    value as the reference's division) and sums it in the reference's order, so r is bit-identical and the only
    difference left is exp itself: the inlined exp is within 0.78 ulp of the true value, numpy's within 0.66, i.e. at most
    ~1.5 ulp apart per SquaredExponential summand (measured: 1).  Every entry is held to 4 ulp of the LARGEST summand -
    for a sum of terms the small entries inherit the absolute error of the big ones."""
    assert K.shape == ref.shape
    assert np.all(np.isfinite(K))
    worst = float(np.max(ulps(K, ref)))
    assert worst <= 4.0 * nterms, 'max K-entry difference %.1f ulp' % worst


@pytest.mark.parametrize('name', NAMES)
def test_golden(name):
    g = load_golden(name)
    kernel, nugget = product_kernel(g, 'kernel'), product_kernel(g, 'nugget')
    x, f, xs = g['x'], g['f'], g['xs']
    interp = mgpi.Interpolator(kernel, nugget=nugget)
    both = kernel if nugget is None else kernel + nugget

    if 'K' in g:
        assert_K_close(mgpi.Interpolator._x2cov(x, x, both), g['K'])
        assert_K_close(mgpi.Interpolator._x2cov(xs, x, kernel), g['Ks'])
    else:
        K = mgpi.Interpolator._x2cov(x, x, both)
        np.testing.assert_allclose([K.sum(), np.abs(K).sum(), np.trace(K)], g['K_checksum'], rtol=1e-13)

    if bool(g['singular']):
        # duplicated rows under the exact-equality white noise: K is exactly singular and the reference's
        # inv raises LinAlgError; the device Cholesky reports the vanishing pivot
        with pytest.raises(np.linalg.LinAlgError):
            interp.compress(x, f)
        with pytest.raises(AssertionError):
            interp.loglikelihood(x, f)
        return

    cond = float(g['cond_K'])
    alpha = interp.compress(x, f)
    assert alpha.shape == (len(x),) and alpha.flags['C_CONTIGUOUS']
    assert relmax(alpha, g['alpha']) < max(1e-12, 100 * cond * EPS)

    pred = interp.predict(xs, x, alpha)
    assert pred.shape == (len(xs),)
    assert relmax(pred, g['predict_mean']) < mean_tol(cond)
    # the reference's own compressed vector through our predict: isolates the fused K* contraction.  Only the
    # summation order differs, so the bound is the dot product's own: eps * sum_j |K*_ij alpha_j| (which is
    # 1e5 x the result for the ill-conditioned C1 fixture), not a condition-number bound
    Ks_ref = g['Ks'] if 'Ks' in g else oracle.x2cov_chunked(oracle_terms(g, 'kernel'), xs, x)
    dot_bound = 8 * EPS * np.max(np.abs(Ks_ref) @ np.abs(g['alpha'])) / np.max(np.abs(g['predict_mean']))
    assert relmax(interp.predict(xs, x, g['alpha']), g['predict_mean']) < dot_bound + 1e-14

    mean, cov = interp.condition(xs, x, f)
    assert mean.shape == (len(xs),) and cov.shape == (len(xs), len(xs))
    assert relmax(mean, g['cond_mean']) < mean_tol(cond)
    pv = prior_var(g)
    assert np.max(np.abs(np.diag(cov) - g['cond_var'])) / pv < var_tol(cond)
    if 'cond_cov' in g:
        assert np.max(np.abs(cov - g['cond_cov'])) / pv < var_tol(cond)

    mean2, var = interp.condition_diag(xs, x, f)
    assert relmax(mean2, g['cond_mean']) < mean_tol(cond)
    assert np.max(np.abs(var - g['cond_var'])) / pv < var_tol(cond)

    ll = interp.loglikelihood(x, f)
    assert abs(ll - float(g['loglike'])) <= mean_tol(cond) * abs(float(g['loglike']))


@pytest.mark.parametrize('d', [1, 2, 3, 4, 5, 7, 16])
def test_paired_cov_matches_oracle(d):
    """Kernel.cov keeps the reference's row-paired semantics (kernels.py:89-93)"""
    rng = np.random.default_rng(d)
    p = 1000
    x1 = rng.random((p, d))
    x2 = rng.random((p, d))
    x2[::7] = x1[::7]                       # exact coincidences -> white noise fires
    x2[3, 0] = np.nextafter(x1[3, 0], 2.0)  # one ulp apart is NOT equal
    ell = tuple(0.2 + 0.05 * k for k in range(d))
    k = mgpi.SquaredExponentialKernel(1.5, *ell) + mgpi.WhiteNoiseKernel(0.3) + mgpi.SquaredExponentialKernel(0.2, *(2 * np.array(ell)))
    terms = [oracle.term_se(1.5, *ell), oracle.term_white(0.3), oracle.term_se(0.2, *(2 * np.array(ell)))]
    got = k.cov(x1, x2)
    ref = oracle.paired_cov(terms, x1, x2)
    assert got.shape == (p,)
    if d < 8:
        assert float(np.max(ulps(got, ref))) <= 8.0          # two SE summands at <= 4 ulp each (r itself is bit-identical)
    else:
        # numpy sums 8 or more terms pairwise with 8 accumulators; the device sums left to right
        np.testing.assert_allclose(got, ref, rtol=5e-14, atol=0)
    # white noise delta is exact
    w = mgpi.WhiteNoiseKernel(0.3).cov(x1, x2)
    assert np.array_equal(w, oracle.paired_cov([oracle.term_white(0.3)], x1, x2))
    assert w[0] == 0.3**2 and w[1] == 0.0 and w[3] == 0.0


def test_exp_over_its_whole_range():
    """the inlined exp(-r) of the kernel build against numpy's on r in [0, 760]: <= 2 ulp down to the last subnormal,
    exact zeros beyond, and lengths that are not normal numbers (0, inf) take the true-division path with the
    reference's values (inf -> 0, 0/0 -> NaN)"""
    rng = np.random.default_rng(12)
    r = np.concatenate([rng.random(20000) * 760.0, 10.0**rng.uniform(-300, 0, 5000), [0.0, 745.13, 745.14, 746.0, 800.0, 1e6]])
    x1 = np.sqrt(r).reshape(-1, 1)
    x2 = np.zeros_like(x1)
    got = mgpi.SquaredExponentialKernel(1.0, 1.0).cov(x1, x2)
    with np.errstate(under='ignore'):
        ref = oracle.paired_cov([oracle.term_se(1.0, 1.0)], x1, x2)
    assert float(np.max(ulps(got, ref))) <= 2.0
    assert np.array_equal(got[ref == 0.0], ref[ref == 0.0])
    for ell in (0.0, np.inf):
        xa, xb = np.array([[0.0], [1.0], [2.0]]), np.array([[0.0], [1.5], [2.0]])
        with np.errstate(all='ignore'):
            want = oracle.paired_cov([oracle.term_se(1.3, ell)], xa, xb)
        np.testing.assert_array_equal(mgpi.SquaredExponentialKernel(1.3, ell).cov(xa, xb), want)


def test_se_convention_and_diagonal():
    """no factor 1/2 (kernels.py:185); K_ii = sigma^2 (+ sigma_n^2)"""
    se = mgpi.SquaredExponentialKernel(2.0, 0.5)
    assert abs(se.cov([[0.0]], [[0.5]])[0] - 4.0 * np.exp(-1.0)) < 1e-15
    K = mgpi.Interpolator._x2cov([[0.0], [0.5], [0.0]], [[0.0], [0.5], [0.0]], se + mgpi.WhiteNoiseKernel(0.1))
    assert K[0, 0] == 4.0 + 0.1**2 and K[0, 2] == 4.0 + 0.1**2     # duplicated rows get the nugget off-diagonal
    assert abs(K[0, 1] - 4.0 * np.exp(-1.0)) < 1e-15


def test_closed_forms_n1_n2():
    se = mgpi.SquaredExponentialKernel(2.0, 0.5)
    it = mgpi.Interpolator(se, nugget=mgpi.WhiteNoiseKernel(0.1))
    kk = 4.0 + 0.01
    ll = it.loglikelihood([[0.3]], [1.5])
    assert abs(ll - (-0.5 * 1.5**2 / kk - 0.5 * np.log(kk) - 0.5 * np.log(2 * np.pi))) < 1e-14
    mean, cov = it.condition([[0.3]], [[0.3]], [1.5])
    assert abs(mean[0] - 1.5 * 4.0 / kk) < 1e-14 and abs(cov[0, 0] - (4.0 - 16.0 / kk)) < 1e-13
    # N = 2 against the explicit 2x2 inverse
    x = np.array([[0.0], [0.4]])
    f = np.array([1.0, -0.5])
    c = 4.0 * np.exp(-(0.4 / 0.5)**2)
    Kinv = np.array([[kk, -c], [-c, kk]]) / (kk * kk - c * c)
    assert relmax(it.compress(x, f), Kinv @ f) < 1e-14


def test_exact_interpolation_as_nugget_vanishes():
    rng = np.random.default_rng(5)
    x = rng.random((40, 2)) * 3
    f = np.sin(x[:, 0]) + x[:, 1]
    it = mgpi.Interpolator(mgpi.SquaredExponentialKernel(1.0, 0.4, 0.4), nugget=mgpi.WhiteNoiseKernel(1e-7))
    mean, var = it.condition_diag(x, x, f)
    assert relmax(mean, f) < 1e-9 and np.max(np.abs(var)) < 1e-9


def test_predict_of_compress_equals_condition_mean():
    g = load_golden('sceos_3d')
    it = mgpi.Interpolator(product_kernel(g, 'kernel'), nugget=product_kernel(g, 'nugget'))
    a = it.compress(g['x'], g['f'])
    m1 = it.predict(g['xs'], g['x'], a)
    m2, _ = it.condition(g['xs'], g['x'], g['f'])
    assert relmax(m1, m2) < 1e-12


@pytest.mark.parametrize('n,m,d,seed', [(257, 33, 2, 11), (640, 129, 3, 12), (1500, 300, 4, 13), (1100, 17, 1, 14)])
def test_seeded_against_oracle(n, m, d, seed):
    """ragged sizes (not multiples of any tile), against the oracle's LU-inverse statement"""
    rng = np.random.default_rng(seed)
    x = rng.random((n, d))
    xs = rng.random((m, d))
    f = np.sin(4 * x[:, 0]) + (x[:, -1] - 0.5)**2
    ell = tuple(np.full(d, 1.5 * n**(-1.0 / d)))
    it = mgpi.Interpolator(mgpi.SquaredExponentialKernel(1.0, *ell), nugget=mgpi.WhiteNoiseKernel(1e-2))
    kernel, nugget = [oracle.term_se(1.0, *ell)], [oracle.term_white(1e-2)]
    K = oracle.x2cov_chunked(kernel + nugget, x, x)
    cond = np.linalg.cond(K)
    ref_mean, ref_cov = oracle.condition(kernel, nugget, xs, x, f)
    mean, cov = it.condition(xs, x, f)
    assert relmax(mean, ref_mean) < mean_tol(cond)
    assert np.max(np.abs(cov - ref_cov)) < var_tol(cond)
    mean2, var = it.condition_diag(xs, x, f, m_slab=128)      # several query slabs
    assert relmax(mean2, ref_mean) < mean_tol(cond)
    assert np.max(np.abs(var - np.diag(ref_cov))) < var_tol(cond)
    ll = it.loglikelihood(x, f)
    ref_ll = oracle.loglikelihood(kernel, nugget, x, f)
    assert abs(ll - ref_ll) < mean_tol(cond) * abs(ref_ll)


def test_input_layouts_and_errors():
    rng = np.random.default_rng(3)
    cols = [rng.random(50), rng.random(50)]
    xF = np.transpose(cols)                    # F-ordered (N, D) view, what parse_table returns (io.py:123)
    assert not xF.flags['C_CONTIGUOUS']
    xC = np.ascontiguousarray(xF)
    f = rng.random(50)
    k = mgpi.SquaredExponentialKernel(1.0, 0.3, 0.3)
    it = mgpi.Interpolator(k, nugget=mgpi.WhiteNoiseKernel(0.1))
    assert np.array_equal(it.compress(xF, f), it.compress(xC, f))
    assert np.array_equal(it.compress(xC.tolist(), f.tolist()), it.compress(xC, f))
    with pytest.raises(AssertionError):       # interpolators.py:219
        mgpi.Interpolator._x2cov(xC, rng.random((5, 3)), k)
    with pytest.raises(AssertionError):
        it.predict(rng.random((5, 3)), xC, f)
    # not positive definite: loglikelihood asserts like interpolators.py:340; compress raises LinAlgError
    bad = mgpi.Interpolator(mgpi.SquaredExponentialKernel(1.0, 50.0, 50.0))      # nearly rank one, no nugget
    with pytest.raises(AssertionError, match='covariance is not positive definite!'):
        bad.loglikelihood(xC, f)
    with pytest.raises(np.linalg.LinAlgError):
        bad.compress(xC, f)


def test_empty_inputs_raise_like_the_reference():
    """the reference's pair builder indexes row 0 before anything else (interpolators.py:247-249): every entry point
    raises IndexError on an empty source or target set [measured on the reference]"""
    it = mgpi.Interpolator(mgpi.SquaredExponentialKernel(1.0, 0.3, 0.3), nugget=mgpi.WhiteNoiseKernel(1e-2))
    rng = np.random.default_rng(0)
    x, f = rng.random((5, 2)), rng.random(5)
    none_x, none_f = np.zeros((0, 2)), np.zeros(0)
    for call in [lambda: it.condition(none_x, x, f), lambda: it.condition_diag(none_x, x, f),
                 lambda: it.predict(none_x, x, it.compress(x, f)), lambda: it.condition(x, none_x, none_f),
                 lambda: it.compress(none_x, none_f), lambda: it.loglikelihood(none_x, none_f),
                 lambda: mgpi.Interpolator._x2cov(none_x, x, it.kernel), lambda: mgpi.Interpolator._x2cov(x, none_x, it.kernel)]:
        with pytest.raises(IndexError):
            call()


def test_verbose_stage_banners(capsys):
    rng = np.random.default_rng(4)
    x = rng.random((20, 2))
    it = mgpi.Interpolator(mgpi.SquaredExponentialKernel(1.0, 0.3, 0.3), nugget=mgpi.WhiteNoiseKernel(0.1))
    it.condition(x[:5], x, rng.random(20), verbose=True)
    out = capsys.readouterr().out
    for phrase in ['constructing 20 x 20 source-source covariance matrix', 'inverting source-source covariance matrix',
                   'computing conditioned mean', 'computing conditioned covariance', '    time : ']:
        assert phrase in out


def test_update_invalidates_nothing_stale():
    """objects are mutable (update); every call must see the current parameters"""
    rng = np.random.default_rng(8)
    x = rng.random((30, 1))
    f = rng.random(30)
    k = mgpi.SquaredExponentialKernel(1.0, 0.2)
    it = mgpi.Interpolator(k, nugget=mgpi.WhiteNoiseKernel(0.1))
    a1 = it.compress(x, f)
    it.update(sigma=2.0)
    a2 = it.compress(x, f)
    ref = oracle.compress([oracle.term_se(2.0, 0.2)], [oracle.term_white(0.1)], x, f)
    assert relmax(a2, ref) < 1e-11 and relmax(a1, a2) > 1e-3


def test_rvs_statistics_and_shapes():
    """rvs has no golden output (the reference's own rvs raises TypeError): parity unpinned.
    Checked statistically against the conditioned mean/covariance and for determinism under a fixed seed."""
    rng = np.random.default_rng(17)
    x = rng.random((60, 2))
    f = np.sin(3 * x[:, 0]) + x[:, 1]
    xs = rng.random((6, 2))
    it = mgpi.Interpolator(mgpi.SquaredExponentialKernel(1.0, 0.25, 0.25), nugget=mgpi.WhiteNoiseKernel(0.05))
    mean, cov = it.condition(xs, x, f)
    draws = it.rvs(xs, x, f, size=20000, rng=np.random.default_rng(1))
    assert draws.shape == (20000, 6)
    se = np.sqrt(np.diag(cov) / 20000)
    assert np.all(np.abs(draws.mean(axis=0) - mean) < 5 * se)
    emp = np.cov(draws, rowvar=False)
    assert np.max(np.abs(emp - cov)) < 0.05 * np.max(np.diag(cov))
    again = it.sample(xs, x, f, size=20000, rng=np.random.default_rng(1))
    assert np.array_equal(draws, again)
    assert it.rvs(xs, x, f).shape == (1, 6)
    # targets on (almost) noise-free sources: the conditioned covariance is only semi-definite; still samples
    it2 = mgpi.Interpolator(mgpi.SquaredExponentialKernel(1.0, 0.25, 0.25), nugget=mgpi.WhiteNoiseKernel(1e-6))
    d2 = it2.rvs(x[:5], x, f, size=10, rng=np.random.default_rng(2))
    assert d2.shape == (10, 5) and np.all(np.isfinite(d2)) and np.max(np.abs(d2 - f[:5])) < 1e-3


def test_rvs_arithmetic_against_the_restated_intent():
    """deterministic check of rvs [r3]: the same standard-normal deviates z through the oracle's restatement of the
    reference's documented intent, mean + (chol(cov / outer(s, s)) z^T)^T s  (mgpi/interpolators.py:314-324), with mean
    and cov from the ORACLE's condition().  Jitter-free case (nugget large enough for a definite conditioned covariance).
    Tolerance 1e-10 of the largest scale: two Cholesky factors of a matrix with cond ~1e3."""
    rng = np.random.default_rng(29)
    x = rng.random((80, 2))
    f = np.sin(3 * x[:, 0]) + x[:, 1]
    xs = rng.random((9, 2))
    it = mgpi.Interpolator(mgpi.SquaredExponentialKernel(1.0, 0.25, 0.3), nugget=mgpi.WhiteNoiseKernel(0.05))
    kernel, nugget = [oracle.term_se(1.0, 0.25, 0.3)], [oracle.term_white(0.05)]
    mean_ref, cov_ref = oracle.condition(kernel, nugget, xs, x, f)

    class Fixed(object):                                   # hands the sampler a known z
        def __init__(self, z):
            self.z = z

        def standard_normal(self, shape):
            assert tuple(shape) == self.z.shape
            return self.z.copy()
    z = np.random.default_rng(5).standard_normal((7, 9))
    got = it.rvs(xs, x, f, size=7, rng=Fixed(z))
    want = oracle.rvs_cholesky(mean_ref, cov_ref, z)
    assert got.shape == want.shape == (7, 9)
    assert np.max(np.abs(got - want)) < 1e-10 * max(1.0, np.max(np.abs(want)))


def test_kernel_call_is_the_pairwise_matrix():
    g = load_golden('toy_multiterm')
    k = product_kernel(g, 'kernel')
    assert_K_close(k(g['xs'], g['x']), g['Ks'])
    both = k + product_kernel(g, 'nugget')
    assert_K_close(both(g['x']), g['K'])


def test_optimize_kernel_recovers_a_better_likelihood():
    """optimize_kernel (scipy TNC with finite-difference gradients over the device loglikelihood, from a RANDOM start
    like the reference's, interpolators.py:439-487) must not end below the likelihood of its own starting point and
    must leave the kernel at the parameters it returns.  (How far it climbs is the optimiser's business: its 1e-8
    steps see the rounding of the likelihood and -inf outside the positive orthant, so it often stops early.)"""
    rng = np.random.default_rng(23)
    x = rng.random((150, 1)) * 4
    f = np.sin(2 * x[:, 0]) + 0.05 * rng.standard_normal(150)
    k = mgpi.SquaredExponentialKernel(0.5, 0.2)
    it = mgpi.Interpolator(k, nugget=mgpi.WhiteNoiseKernel(0.05))
    np.random.seed(3)
    start = it._construct_initial_params(size=1)[0]          # the same draw optimize_kernel makes below
    it.update(**dict(zip(k._params, start)))
    at_start = it.loglikelihood(x, f)
    it.update(sigma=0.5, length0=0.2)                        # the draw is centred on the current parameters
    np.random.seed(3)
    out = it.optimize_kernel(x, f)
    after = it.loglikelihood(x, f)
    assert np.isfinite(after) and after >= at_start - 1e-6 * abs(at_start)
    assert out.dtype.names == ('sigma', 'length0') and out['sigma'][0] == k.params[0] and out['length0'][0] == k.params[1]


#-------------------------------------------------
# Matern (kernels.py:135-162): the device evaluates K_order itself (Temme / Steed), scipy.special.kv is the checker


@pytest.mark.parametrize('name', matern_names())
def test_matern_golden(name):
    """every entry at non-zero separation against the reference's own output (rtol 2e-12: K_order by series /
    continued fraction to ~1e-13, and the fixtures hold separations 1e-7 ell .. 50 ell); the zero-separation
    diagonal, NaN in the reference, is the limit sigma**2 (documented deviation)"""
    g = load_golden(name)
    kern = matern_terms(g, 'product')
    K = mgpi.Interpolator._x2cov(g['x'], g['x'], kern)
    Ks = mgpi.Interpolator._x2cov(g['xs'], g['x'], kern)
    off = ~np.eye(len(K), dtype=bool)
    np.testing.assert_allclose(K[off], g['K'][off], rtol=2e-12, atol=1e-300)
    np.testing.assert_allclose(Ks, g['Ks'], rtol=2e-12, atol=1e-300)
    extra = sum(float(s)**2 for s in g['extra_sigmas'])
    np.testing.assert_allclose(np.diag(K), float(g['sigma'])**2 + extra, rtol=1e-15)
    np.testing.assert_array_equal(K, K.T)
    # row-paired entry point (Kernel.cov) agrees with the matrix build bit for bit
    i, j = np.arange(len(g['xs'])), (np.arange(len(g['xs'])) * 7) % len(g['x'])
    np.testing.assert_array_equal(kern.cov(g['xs'][i], g['x'][j]), Ks[i, j])


@pytest.mark.parametrize('order', [0.05, 0.3, 0.5, 0.999, 1.0, 1.0001, 1.5, 2.0, 2.5, 3.49, 3.5, 3.51, 6.0, 12.3, 40.0, 120.0])
def test_matern_orders_against_oracle(order):
    """separations from 1e-9 to 300 length scales, both branches of K_order (x < 2 series, x >= 2 continued fraction)
    and the upward recurrence.  AMOS behind scipy.special.kv flushes K to exactly zero from x ~ 650 on, where
    r**order K(r) is still representable and the device returns it: those points only have to be tiny"""
    rng = np.random.default_rng(int(order * 1000))
    r = np.concatenate([10.0**rng.uniform(-9, 2.5, 3000), np.linspace(1.9, 2.1, 41) / np.sqrt(2 * order)])
    x1 = rng.standard_normal((len(r), 3))
    u = rng.standard_normal((len(r), 3))
    u /= np.linalg.norm(u, axis=1)[:, None]
    lengths = (0.7, 1.3, 2.1)
    x2 = x1 + u * r[:, None] * np.array(lengths)
    got = mgpi.MaternKernel(order, 1.7, *lengths).cov(x1, x2)
    want = oracle.paired_cov([oracle.term_matern(order, 1.7, *lengths)], x1, x2)
    ok = np.isfinite(want)                  # kv overflows to inf (times 0 -> NaN) for tiny separations at large order
    assert ok.mean() > 0.25          # order 120: kv itself overflows below x ~ 1
    flushed = ok & (want == 0) & (np.sqrt(2 * order) * r > 600)
    np.testing.assert_allclose(got[ok & ~flushed], want[ok & ~flushed], rtol=5e-12, atol=1e-300)
    assert np.all(got[flushed] < 1e-150)
    assert np.all(np.isfinite(got)) and np.all(got <= 1.7**2 * (1 + 1e-12)) and np.all(got >= 0)
    # where the reference's expression gives up (inf * 0): near pairs follow the small-separation limit sigma**2, the
    # rest is checked in log space with the exponentially scaled kve where that is finite
    x = np.sqrt(2 * order) * r
    near = ~ok & (x < 1e-3)
    assert np.all(np.abs(got[near] - 1.7**2) < 1e-5 * 1.7**2)
    from scipy.special import gammaln, kve
    with np.errstate(all='ignore'):
        logm = (1 - order) * np.log(2) - gammaln(order) + order * np.log(x) + np.log(kve(order, x)) - x
    rest = ~ok & ~near & np.isfinite(logm) & (logm > -600) & (x < 700)    # beyond: exp(-x) underflows, the device returns 0
    np.testing.assert_allclose(got[rest], 1.7**2 * np.exp(logm[rest]), rtol=1e-10)


def test_matern_half_integer_closed_forms():
    """order 1/2: sigma**2 exp(-r); 3/2: (1+r) exp(-r); 5/2: (1 + r + r**2/3) exp(-r)   with r = sqrt(2 order) d"""
    rng = np.random.default_rng(5)
    x1, x2 = rng.random((500, 2)), rng.random((500, 2))
    d = np.sqrt(np.sum(((x1 - x2) / np.array([0.2, 0.4]))**2, axis=1))
    for order, poly in [(0.5, lambda r: 1.0), (1.5, lambda r: 1 + r), (2.5, lambda r: 1 + r + r * r / 3)]:
        r = np.sqrt(2 * order) * d
        got = mgpi.MaternKernel(order, 2.0, 0.2, 0.4).cov(x1, x2)
        np.testing.assert_allclose(got, 4.0 * poly(r) * np.exp(-r), rtol=1e-12)


def test_matern_full_path_like_the_example_ini():
    """etc/example-interpolator.ini: SE(0.1; 10, 20) + WhiteNoise + Matern(2, 0.1; 15, 10) - unusable in the
    reference (NaN diagonal).  With the zero-separation limit it is a valid kernel: check the whole path against the
    oracle run with the same convention (MATERN_ZERO_LIMIT, oracle header)."""
    rng = np.random.default_rng(21)
    x = np.transpose([rng.uniform(0, 150, 400), rng.uniform(30, 100, 400)])
    xs = np.transpose([rng.uniform(0, 150, 90), rng.uniform(30, 100, 90)])
    f = 0.05 * np.tanh((x[:, 1] - 60.0) / 20.0) + 0.01 * np.sin(x[:, 0] / 30.0)
    kern = mgpi.SquaredExponentialKernel(0.1, 10, 20) + mgpi.MaternKernel(2, 0.1, 15, 10)
    interp = mgpi.Interpolator(kern, nugget=mgpi.WhiteNoiseKernel(1e-3))
    kt = [oracle.term_se(0.1, 10, 20), oracle.term_matern(2, 0.1, 15, 10)]
    nt = [oracle.term_white(1e-3)]
    oracle.MATERN_ZERO_LIMIT = True
    try:
        cond = np.linalg.cond(oracle.x2cov(kt + nt, x, x))
        want_mean, want_cov = oracle.condition(kt, nt, xs, x, f)
        want_ll = oracle.loglikelihood(kt, nt, x, f)
    finally:
        oracle.MATERN_ZERO_LIMIT = False
    mean, cov = interp.condition(xs, x, f)
    assert relmax(mean, want_mean) < mean_tol(cond)
    assert np.max(np.abs(cov - want_cov)) < var_tol(cond) * 0.02
    assert abs(interp.loglikelihood(x, f) - want_ll) < mean_tol(cond) * abs(want_ll)


def test_matern_order_limits():
    for bad in [0.0, -1.0, 120.5, np.nan]:
        with pytest.raises((ValueError, RuntimeError, AssertionError)):
            mgpi.MaternKernel(bad, 1.0, 1.0).cov(np.zeros((1, 1)), np.ones((1, 1)))


def test_loglikelihood_batch_matches_single_calls_and_golden():
    """mgpi_loglikelihood_batch: every member equals the single-call value and the
    reference's own loglikelihood (rtol 1e-10, cond-scaled like the mean); a member that is not positive definite
    returns -inf with its pivot and leaves the others untouched; the kernel's parameters are restored"""
    for name in ('toy_n64_d2_nugget', 'toy_multiterm', 'c1_n512_d2', 'uiuc_checkerboard_nugget', 'sceos_3d'):
        assert name in NAMES
        g = load_golden(name)
        kernel, nugget = product_kernel(g, 'kernel'), product_kernel(g, 'nugget')
        it = mgpi.Interpolator(kernel, nugget=nugget)
        base = np.array(kernel.params, dtype=float)
        rows = np.array([base, base * 1.1, base * 0.9, base * np.linspace(0.8, 1.2, len(base))])
        vals, piv = it.loglikelihoods(g['x'], g['f'], rows)
        assert np.array_equal(np.array(kernel.params, dtype=float), base) and np.all(piv == 0)
        tol = max(1e-10, 4e-17 * float(g['cond_K']))
        assert abs(vals[0] - float(g['loglike'])) < tol * abs(float(g['loglike']))
        for row, v in zip(rows, vals):
            kernel.update(*row)
            ref = oracle.loglikelihood(oracle_terms_of(kernel), oracle_terms_of(nugget), g['x'], g['f'])
            assert abs(v - ref) < max(tol, 1e-9) * abs(ref)
        kernel.update(*base)


def test_loglikelihood_keeps_unchanged_data_on_the_device_and_sees_every_change():
    """Interpolator.loglikelihood(x, f) - the reference's entry point, called again and again by a hyper-parameter search
    (interpolators.py:331-343, 382) - reuses the device copies of x and f while their VALUES are unchanged, and uploads again
    when a caller mutates an array in place, passes other arrays of the same shape, or changes the shape"""
    rng = np.random.default_rng(11)
    x = rng.random((200, 2))
    f = np.sin(3 * x[:, 0]) + x[:, 1]
    kernel, nugget = mgpi.SquaredExponentialKernel(1.0, 0.3, 0.4), mgpi.WhiteNoiseKernel(0.05)
    it = mgpi.Interpolator(kernel, nugget=nugget)

    def ref(xx, ff):
        return oracle.loglikelihood(oracle_terms_of(kernel), oracle_terms_of(nugget), xx, ff)

    v0 = it.loglikelihood(x, f)
    held = it._ll_data[3], it._ll_data[4]
    assert abs(v0 - ref(x, f)) < 1e-10 * abs(v0)
    assert it.loglikelihood(x.copy(), f.copy()) == v0 and (it._ll_data[3], it._ll_data[4]) == held      # same values: no upload
    kernel.update(1.2, 0.25, 0.5)                                                                       # new parameters, same data
    v1 = it.loglikelihood(x, f)
    assert (it._ll_data[3], it._ll_data[4]) == held and abs(v1 - ref(x, f)) < 1e-10 * abs(v1) and v1 != v0
    f[17] += 0.5                                                                                        # mutated in place
    v2 = it.loglikelihood(x, f)
    assert it._ll_data[4] is not held[1] and abs(v2 - ref(x, f)) < 1e-10 * abs(v2) and v2 != v1
    x2 = rng.random((200, 2))                                                                           # other points, same shape
    v3 = it.loglikelihood(x2, f)
    assert abs(v3 - ref(x2, f)) < 1e-10 * abs(v3)
    v4 = it.loglikelihood(x2[:150], f[:150])                                                            # other shape
    assert abs(v4 - ref(x2[:150], f[:150])) < 1e-10 * abs(v4)
    fn = f.copy()
    fn[3] = np.nan                                                                                      # NaNs never compare equal
    it.loglikelihoods(x2, fn, [np.array(kernel.params, dtype=float)])
    a = it._ll_data[4]
    it.loglikelihoods(x2, fn, [np.array(kernel.params, dtype=float)])
    assert it._ll_data[4] is not a


def oracle_terms_of(k):
    if k is None:
        return []
    out = []
    for kind, sigma, lengths in [(t[0], t[1], t[2]) for t in k.terms()]:
        out.append(oracle.term_se(sigma, *lengths) if kind == 0 else oracle.term_white(sigma))
    return out


@pytest.mark.parametrize('n,d', [(1, 1), (63, 2), (64, 2), (65, 3), (338, 2), (1311, 2), (2050, 3)])
def test_loglikelihood_batch_sizes_and_failures(n, d):
    rng = np.random.default_rng(n)
    x = rng.random((n, d))
    f = np.sin(3 * x[:, 0]) + 0.1 * rng.standard_normal(n)
    k = mgpi.SquaredExponentialKernel(1.0, *([0.2] * d)) + mgpi.WhiteNoiseKernel(0.05)
    it = mgpi.Interpolator(k)
    base = np.array(k.params, dtype=float)
    good = [base, base * 1.3, base * 0.7]
    res = it.resident(x, f)
    vals, piv = it.loglikelihoods(x, f, good, resident=res)
    for row, v in zip(good, vals):
        ref = oracle.loglikelihood([oracle.term_se(row[0], *row[1:1 + d]), oracle.term_white(row[1 + d])], [], x, f)
        assert abs(v - ref) <= 1e-10 * max(1.0, abs(ref)) * (1 + n / 500.0)
    assert np.all(piv == 0)
    # single call == member of the batch, bit for bit (same kernels)
    assert it.loglikelihood(x, f) == vals[0]
    if n >= 20:
        # one member with no noise and length scales far beyond the data's extent: K is numerically rank deficient for
        # that member only (its pivots fall below the tolerance)
        singular = base.copy()
        singular[1:1 + d] *= 300.0
        singular[-1] = 0.0
        vals2, piv2 = it.loglikelihoods(x, f, [base, singular, base * 1.3], resident=res)
        assert vals2[0] == vals[0] and vals2[2] == vals[1] and vals2[1] == -np.inf
        assert piv2[0] == 0 and piv2[2] == 0 and 1 <= piv2[1] <= n
        k.update(*singular)
        try:
            with pytest.raises(AssertionError):
                it.loglikelihood(x, f)
        finally:
            k.update(*base)


def test_logposterior_vectorised_call():
    """LogPosterior on a (B, num_dim) array (emcee's vectorize=True): one device call, invalid rows -inf"""
    from mgpi_amd.hyper import LogPosterior
    rng = np.random.default_rng(5)
    x = rng.random((200, 2))
    f = np.sin(4 * x[:, 0]) + x[:, 1]
    k = mgpi.SquaredExponentialKernel(1.0, 0.3, 0.3) + mgpi.WhiteNoiseKernel(0.05)
    it = mgpi.Interpolator(k)
    lp = LogPosterior(it, x, f, fixed={'sigma_1': 0.05}, temperature=2.0)
    pts = np.array([[1.0, 0.3, 0.3], [-1.0, 0.3, 0.3], [1.2, 0.25, 0.4], [np.nan, 1, 1]])
    got = lp(pts)
    assert got.shape == (4,) and got[1] == -np.inf and got[3] == -np.inf
    for i in (0, 2):
        assert abs(got[i] - lp(pts[i])) < 1e-12 * abs(got[i])
        ref = oracle.loglikelihood([oracle.term_se(pts[i][0], pts[i][1], pts[i][2]), oracle.term_white(0.05)], [], x, f) / 2.0
        assert abs(got[i] - ref) < 1e-10 * abs(ref)
