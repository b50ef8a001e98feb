"""Tile-level checks of the HIP building blocks through the C-ABI (ctypes): the fp64 MFMA gemm (layout bugs
show up as exact-integer mismatches), the blocked Cholesky against LAPACK, the triangular solves, and the
size-independent properties at BASELINE.json's full single-GPU sizes (C2 N=8192, C3 N=32768)."""
import ctypes

import numpy as np
import pytest

import mgpi_amd as mgpi
from mgpi_amd import _lib
from conftest import relmax
from oracle import mgpi_oracle as oracle

pytestmark = pytest.mark.gpu


@pytest.fixture(scope='module')
def ctx():
    return mgpi.default_context()


def dev(ctx, a, wide=True):
    return _lib.DeviceMatrix.from_host(ctx, a, wide=wide)


@pytest.mark.parametrize('m,n,k', [(128, 128, 16), (256, 128, 64), (128, 384, 512), (640, 640, 128)])
def test_gemm_nt_exact_integers(ctx, m, n, k):
    """C -= A B^T with small integers is exact in fp64: any fragment/lane mapping error shows as a mismatch.
    A and B are asymmetric and C is not symmetric, so a transposed write-back cannot hide."""
    rng = np.random.default_rng(m + n + k)
    A = rng.integers(-8, 9, (m, k)).astype(float)
    B = rng.integers(-8, 9, (n, k)).astype(float)
    C = rng.integers(-100, 100, (m, n)).astype(float)
    dA, dB, dC = dev(ctx, A), dev(ctx, B), dev(ctx, C)      # keep the handles alive across the call
    _lib.check(ctx.lib.mgpi_gemm_nt_dev(ctx.handle, dA.handle, dB.handle, dC.handle, 0))
    assert np.array_equal(dC.to_host(), C - A @ B.T)


def test_gemm_lower_only_touches_lower_tiles(ctx):
    rng = np.random.default_rng(0)
    n, k = 512, 64
    A = rng.integers(-4, 5, (n, k)).astype(float)
    C = rng.integers(-50, 50, (n, n)).astype(float)
    dC = dev(ctx, C)
    dA = dev(ctx, A)
    _lib.check(ctx.lib.mgpi_gemm_nt_dev(ctx.handle, dA.handle, dA.handle, dC.handle, 1))
    got = dC.to_host()
    full = C - A @ A.T
    ti = np.arange(n)[:, None] // 128
    tj = np.arange(n)[None, :] // 128
    assert np.array_equal(got[tj <= ti], full[tj <= ti])        # tiles on/below the diagonal updated
    assert np.array_equal(got[tj > ti], C[tj > ti])             # strictly-upper tiles untouched


@pytest.mark.parametrize('n', [1, 2, 63, 64, 65, 128, 200, 513, 1111])
def test_potrf_against_lapack(ctx, n):
    rng = np.random.default_rng(n)
    x = rng.random((n, 2))
    terms = [oracle.term_se(1.0, 0.2, 0.3), oracle.term_white(0.05)]
    K = oracle.x2cov_chunked(terms, x, x)
    dK = dev(ctx, K)
    logdet = ctypes.c_double(0.0)
    _lib.check(ctx.lib.mgpi_potrf_dev(ctx.handle, dK.handle, ctypes.byref(logdet)))
    L = np.tril(dK.to_host())
    ref = np.linalg.cholesky(K)
    assert relmax(L, ref) < 1e-11
    assert np.max(np.abs(L @ L.T - K)) / np.max(np.abs(K)) < 1e-14
    assert abs(logdet.value - np.linalg.slogdet(K)[1]) < 1e-10 * max(1.0, abs(logdet.value))


def test_potrf_reports_first_bad_pivot(ctx):
    K = np.eye(300)
    K[150, 150] = -1.0
    dK = dev(ctx, K)
    with pytest.raises(mgpi.NotPositiveDefinite) as err:
        _lib.check(ctx.lib.mgpi_potrf_dev(ctx.handle, dK.handle, None))
    assert err.value.pivot == 151
    assert 'covariance is not positive definite!' in str(err.value)
    b = dev(ctx, np.ones(300), wide=False)
    with pytest.raises(mgpi.MgpiError):          # solving with an unfactored matrix is a state error
        _lib.check(ctx.lib.mgpi_potrs_dev(ctx.handle, dK.handle, b.handle))


@pytest.mark.parametrize('n,nrhs', [(100, 1), (700, 1), (700, 5), (1300, 130)])
def test_potrs_and_trsm(ctx, n, nrhs):
    rng = np.random.default_rng(n + nrhs)
    x = rng.random((n, 3))
    K = oracle.x2cov_chunked([oracle.term_se(1.0, 0.2, 0.2, 0.2), oracle.term_white(0.1)], x, x)
    B = rng.standard_normal((n, nrhs))
    dK = dev(ctx, K)
    _lib.check(ctx.lib.mgpi_potrf_dev(ctx.handle, dK.handle, None))
    out = np.empty_like(B)
    _lib.check(ctx.lib.mgpi_potrs(ctx.handle, dK.handle, B.ctypes.data, n, nrhs, out.ctypes.data))
    ref = np.linalg.solve(K, B)
    assert relmax(out, ref) < 1e-10
    if nrhs > 1:
        L = np.linalg.cholesky(K)
        from scipy.linalg import solve_triangular
        for trans in (0, 1):
            dB = dev(ctx, B)
            _lib.check(ctx.lib.mgpi_trsm_dev(ctx.handle, dK.handle, dB.handle, trans))
            assert relmax(dB.to_host(), solve_triangular(L, B, lower=True, trans=trans)) < 1e-10


@pytest.mark.parametrize('n', [1, 65, 700, 1700, 5000, 16500])
def test_posv_is_potrf_then_potrs(ctx, n):
    """mgpi_posv_dev (factor + solve in one call, the forward sweep riding beside the factorisation on a third stream with the
    64 x 64 diagonal tiles read from the scratch strip) gives BIT-IDENTICAL alpha, factor and logdet to mgpi_potrf_dev followed
    by mgpi_potrs_dev - same kernels, same order [r3].  n = 16500 engages the merged trailing update and two-panel
    super-panels, 5000 the look-ahead panels, 1700 the one-panel tail; a matrix that is not positive definite reports its
    pivot from the fused call too; a multi-column right-hand side takes the plain path."""
    rng = np.random.default_rng(n)
    x = rng.random((n, 3))
    f = np.sin(3 * x[:, 0]) + x[:, 1] * x[:, 2]
    terms = (mgpi.SquaredExponentialKernel(1.0, 0.2, 0.2, 0.2) + mgpi.WhiteNoiseKernel(1e-2)).terms()
    desc, count = _lib.make_terms(terms)
    X = _lib.DeviceMatrix.from_host(ctx, x)
    K1, K2 = _lib.DeviceMatrix(ctx, n, n, wide=True), _lib.DeviceMatrix(ctx, n, n, wide=True)
    for K in (K1, K2):
        _lib.check(ctx.lib.mgpi_cov_matrix_dev(ctx.handle, X.handle, None, desc, count, 1, K.handle))
    a1, a2 = _lib.DeviceMatrix.from_host(ctx, f), _lib.DeviceMatrix.from_host(ctx, f)
    ld1, ld2 = ctypes.c_double(0.0), ctypes.c_double(0.0)
    _lib.check(ctx.lib.mgpi_potrf_dev(ctx.handle, K1.handle, ctypes.byref(ld1)))
    _lib.check(ctx.lib.mgpi_potrs_dev(ctx.handle, K1.handle, a1.handle))
    _lib.check(ctx.lib.mgpi_posv_dev(ctx.handle, K2.handle, a2.handle, ctypes.byref(ld2)))
    assert ld1.value == ld2.value
    assert np.array_equal(a1.to_host(), a2.to_host())
    if n <= 5000:
        assert np.array_equal(np.tril(K1.to_host()), np.tril(K2.to_host()))
        Kh = oracle.x2cov_chunked([oracle.term_se(1.0, 0.2, 0.2, 0.2), oracle.term_white(1e-2)], x, x, rows=max(1, 262144 // n))
        assert np.max(np.abs(Kh @ a2.to_host().reshape(-1) - f)) < 1e-11 * max(1.0, np.max(np.abs(f)))
    # several right-hand sides: factor, then the blocked multi-rhs solve
    if n == 700:
        B = rng.standard_normal((n, 5))
        _lib.check(ctx.lib.mgpi_cov_matrix_dev(ctx.handle, X.handle, None, desc, count, 1, K2.handle))
        dB = dev(ctx, B)
        _lib.check(ctx.lib.mgpi_posv_dev(ctx.handle, K2.handle, dB.handle, None))
        assert relmax(Kh @ dB.to_host(), B) < 1e-10
    # not positive definite: the pivot comes back, nothing hangs on the third stream
    if n == 1700:
        bad = np.eye(n)
        bad[1234, 1234] = -1.0
        dK, db = dev(ctx, bad), _lib.DeviceMatrix.from_host(ctx, f)
        with pytest.raises(mgpi.NotPositiveDefinite) as err:
            _lib.check(ctx.lib.mgpi_posv_dev(ctx.handle, dK.handle, db.handle, None))
        assert err.value.pivot == 1235


@pytest.mark.parametrize('n,m', [(300, 130), (1700, 128), (1700, 1000), (5000, 700), (8192, 4096), (16500, 1500)])
def test_riders_give_the_variance_of_the_two_pass_form(n, m, monkeypatch):
    """condition_diag with the queries RIDING through the factorisation (K* below K: solved by the panels' substitutions,
    updated inside the trailing-update launches; mgpi_posv_riders_dev) against the two-pass form (factor, then the blocked
    multi-rhs solve of mgpi_predict_var_dev) and against the oracle's LAPACK statement: same alpha bit for bit (the square part
    sees the same operations), mean identical, variance to 1e-11 of the prior variance [r3].  Sizes: single panel, one-panel
    tail, look-ahead panels, C2 with its 4096 queries, merged trailing update; n and m off the 128 grid."""
    rng = np.random.default_rng(n + m)
    x = rng.random((n, 3))
    f = np.sin(3 * x[:, 0]) + x[:, 1] * x[:, 2]
    xs = rng.random((m, 3))
    xs[:3] = x[:3]                                           # queries on sources
    it = mgpi.Interpolator(mgpi.SquaredExponentialKernel(1.0, 0.2, 0.2, 0.2), nugget=mgpi.WhiteNoiseKernel(1e-2))
    from mgpi_amd.gp import riders_pay
    monkeypatch.delenv('MGPI_RIDERS', raising=False)
    assert riders_pay(8192, 4096) and not riders_pay(32768, 4096) and not riders_pay(8192, 100) and not riders_pay(1000, 2000)
    monkeypatch.setenv('MGPI_RIDERS', '1')
    mean_r, var_r = it.condition_diag(xs, x, f)
    gp = it.fit(x, f, riders_x=xs)
    alpha_r = gp.alpha.to_host().reshape(-1)
    var_again = gp.variance(xs)                              # the factor inside the riders matrix serves the two-pass entry too
    monkeypatch.setenv('MGPI_RIDERS', '0')
    mean_2, var_2 = it.condition_diag(xs, x, f)
    alpha_2 = it.compress(x, f)
    assert np.array_equal(alpha_r, alpha_2) and np.array_equal(mean_r, mean_2)
    assert np.max(np.abs(var_r - var_2)) < 1e-11 and np.max(np.abs(var_again - var_2)) < 1e-13
    assert np.all(var_r > -1e-12) and np.all(var_r[:3] < 1.1e-4)
    if n <= 5000:
        kernel, nugget = [oracle.term_se(1.0, 0.2, 0.2, 0.2)], [oracle.term_white(1e-2)]
        L = oracle.cholesky_lower(oracle.x2cov_chunked(kernel + nugget, x, x, rows=max(1, 262144 // n)))
        assert np.max(np.abs(var_r - oracle.diag_variance_cholesky(kernel, L, xs, x))) < 1e-10


def test_c_abi_host_entry_points(ctx):
    """the host-pointer entry points a ctypes binding at the reference's call sites would use"""
    rng = np.random.default_rng(21)
    x1, x2 = rng.random((70, 3)), rng.random((45, 3))
    desc, n = _lib.make_terms([(0, 1.2, (0.3, 0.4, 0.5)), (1, 0.1, ())])
    out = np.empty((70, 45))
    _lib.check(ctx.lib.mgpi_cov_matrix(ctx.handle, x1.ctypes.data, 70, x2.ctypes.data, 45, 3, desc, n, out.ctypes.data))
    ref = oracle.x2cov([oracle.term_se(1.2, 0.3, 0.4, 0.5), oracle.term_white(0.1)], x1, x2)
    np.testing.assert_allclose(out, ref, rtol=1e-15)
    sym = np.empty((70, 70))
    _lib.check(ctx.lib.mgpi_cov_matrix(ctx.handle, x1.ctypes.data, 70, None, 0, 3, desc, n, sym.ctypes.data))
    np.testing.assert_allclose(sym, oracle.x2cov([oracle.term_se(1.2, 0.3, 0.4, 0.5), oracle.term_white(0.1)], x1, x1), rtol=1e-15)
    # bad arguments come back as negative statuses with a message, never a crash
    assert ctx.lib.mgpi_cov_matrix(ctx.handle, x1.ctypes.data, 70, x2.ctypes.data, 45, 99, desc, n, out.ctypes.data) < 0
    assert 'd=99' in _lib.last_error()


def _c_config(n_side, d, seed):
    """SURVEY.md section 8d recipe: jittered grid in [0,1]^d, f = sin/cos product, SE sigma=1, ell=2*spacing,
    white sigma_n=1e-2"""
    rng = np.random.default_rng(seed)
    axes = [(np.arange(s) + 0.5) / s for s in n_side]
    x = np.transpose([a.ravel() for a in np.meshgrid(*axes, indexing='ij')])
    x = x + rng.uniform(-0.1, 0.1, x.shape) / np.array(n_side)
    f = np.sin(3 * x[:, 0]) * np.cos(2 * x[:, 1]) + 0.5 * np.sin(4 * x[:, 2])
    ell = tuple(2.0 / np.array(n_side))
    return x, f, ell


@pytest.mark.parametrize('n_side,seed', [((32, 16, 16), 2), ((32, 32, 32), 3)])
def test_full_size_properties(n_side, seed):
    """C2 (N=8192) and C3 (N=32768): the reference cannot run C3 at all, so parity at
    full size is through size-independent properties:
      residual   || (K+nugget) alpha - f ||_inf / ||f||_inf        (K alpha via the fused contraction)
      variance   0 <= var <= prior, and var at source points ~ sigma_n^2 scale
      linearity  alpha(f1 + 2 f2) == alpha(f1) + 2 alpha(f2)
      logdet     identical when the sources are permuted"""
    x, f, ell = _c_config(n_side, 3, seed)
    n = len(x)
    rng = np.random.default_rng(seed)
    k = mgpi.SquaredExponentialKernel(1.0, *ell)
    nug = mgpi.WhiteNoiseKernel(1e-2)
    it = mgpi.Interpolator(k, nugget=nug)
    gp = it.fit(x, f)
    alpha = gp.alpha.to_host().reshape(-1)
    # K alpha: predict with kernel+nugget at the sources themselves
    Kalpha = mgpi.Interpolator(k + nug).predict(x, x, alpha)
    assert np.max(np.abs(Kalpha - f)) / np.max(np.abs(f)) < 1e-10
    xs = rng.random((512, 3))
    xs[:8] = x[:8]
    var = gp.variance(xs)
    assert np.all(var > -1e-9) and np.all(var <= 1.0 + 1e-12)
    assert np.all(var[:8] < 1.1e-4 + 1e-9)            # at a source the variance is below the noise variance
    f2 = np.cos(5 * x[:, 1])
    a2 = it.fit(x, f2).alpha.to_host().reshape(-1)
    a12 = it.fit(x, f + 2 * f2).alpha.to_host().reshape(-1)
    assert relmax(a12, alpha + 2 * a2) < 1e-9
    perm = rng.permutation(n)
    gp_p = it.fit(x[perm], f[perm])
    assert abs(gp_p.logdet - gp.logdet) < 1e-9 * abs(gp.logdet)
    assert relmax(gp_p.alpha.to_host().reshape(-1), alpha[perm]) < 1e-8


@pytest.mark.parametrize('P,p,Q,q,bi0,bj0,mb,nbk', [(1, 0, 1, 0, 1, 1, 5, 5), (2, 1, 4, 2, 0, 0, 6, 3), (2, 0, 4, 3, 2, 1, 4, 3),
                                                   (2, 1, 2, 1, 3, 3, 5, 5), (3, 2, 2, 0, 1, 2, 7, 9)])
def test_gemm_staircase_matches_block_cyclic_definition(ctx, P, p, Q, q, bi0, bj0, mb, nbk):
    """the one-launch local trailing update of the multi-GPU driver: a 128-tile of the local matrix is updated
    iff it lies on/below the GLOBAL diagonal of the block-cyclic matrix (exact integer data)"""
    nb = 256
    rng = np.random.default_rng(P * 100 + Q * 10 + p + q)
    m, n, k = mb * nb, nbk * nb, nb
    A = rng.integers(-4, 5, (m, k)).astype(float)
    B = rng.integers(-4, 5, (n, k)).astype(float)
    C = rng.integers(-50, 50, (m, n)).astype(float)
    dA, dB, dC = dev(ctx, A), dev(ctx, B), dev(ctx, C)
    _lib.check(ctx.lib.mgpi_gemm_nt_stair_dev(ctx.handle, dA.handle, dB.handle, dC.handle, P, p, Q, q, bi0, bj0, nb))
    got = dC.to_host()
    full = C - A @ B.T
    ti = np.arange(m) // 128
    tj = np.arange(n) // 128
    tpb = nb // 128
    gr = (p + P * (bi0 + ti // tpb)) * tpb + ti % tpb          # global tile row of each local row
    gc = (q + Q * (bj0 + tj // tpb)) * tpb + tj % tpb
    live = gr[:, None] >= gc[None, :]
    assert np.array_equal(got[live], full[live])
    assert np.array_equal(got[~live], C[~live])


def test_profile_counts_the_flops_of_the_schedule(ctx):
    """the roofline's numerator: the library's own count of the lower-syrk launches' algorithmic flops (profile slot 0)
    equals the closed form of the shipped schedule in bench.py - at a size where the two-level schedule engages"""
    import bench
    n = 17000
    rng = np.random.default_rng(1)
    x = rng.random((n, 3))
    it = mgpi.Interpolator(mgpi.SquaredExponentialKernel(1.0, 0.06, 0.06, 0.06), nugget=mgpi.WhiteNoiseKernel(0.1), context=ctx)
    ctx.set_profile(True)
    it.fit(x)
    flops = ctx.get_profile_flops(0)
    ms, launches = ctx.get_profile(0)
    ctx.set_profile(False)
    assert flops == bench.syrk_schedule_flops(n) and flops != bench.syrk_launch_flops(n)
    assert launches > 0 and ms > 0
    # a subset of the slots (what bench.py keeps on inside its timed region): the others record nothing
    ctx.set_profile(True, slots=(0,))
    it.fit(x)
    assert ctx.get_profile_flops(0) == flops and ctx.get_profile(0)[1] == launches
    assert ctx.get_profile(2) == (0.0, 0) and ctx.get_profile(1) == (0.0, 0)
    ctx.set_profile(False)


@pytest.mark.parametrize('w,batch,row_step,col_step', [(64, 1, 0, 0), (128, 3, 256, 128), (512, 2, 512, 1024), (256, 5, 256, 256)])
def test_trtri_neg_strided_batch(ctx, w, batch, row_step, col_step):
    """-(block)^-1 for lower-triangular blocks at a constant stride (the diagonal blocks a rank owns in a block-cyclic
    factor): against numpy, and the strict upper triangle of the result is exactly zero"""
    rng = np.random.default_rng(w + batch)
    r0, c0 = 64, 128
    rows, cols = r0 + (batch - 1) * row_step + w, c0 + (batch - 1) * col_step + w
    A = rng.standard_normal((rows, cols))
    for j in range(batch):
        blk = np.tril(rng.standard_normal((w, w))) * 0.1 + np.diag(1.0 + rng.random(w))
        A[r0 + j * row_step:r0 + j * row_step + w, c0 + j * col_step:c0 + j * col_step + w] = blk + np.triu(rng.standard_normal((w, w)), 1)
    dA = dev(ctx, A)
    out = _lib.DeviceMatrix(ctx, batch * w, w, wide=True)
    _lib.check(ctx.lib.mgpi_trtri_neg_dev(ctx.handle, dA.handle, r0, c0, row_step, col_step, w, batch, out.handle))
    got = out.to_host()
    for j in range(batch):
        blk = np.tril(A[r0 + j * row_step:r0 + j * row_step + w, c0 + j * col_step:c0 + j * col_step + w])
        want = -np.linalg.inv(blk)
        g = got[j * w:(j + 1) * w]
        assert np.max(np.abs(g - want)) < 1e-12 * np.max(np.abs(want))
        assert np.count_nonzero(np.triu(g, 1)) == 0
    with pytest.raises(mgpi.MgpiError):
        _lib.check(ctx.lib.mgpi_trtri_neg_dev(ctx.handle, dA.handle, r0, c0, row_step, col_step, w, batch + 50, out.handle))


@pytest.mark.parametrize('nb', [128, 384, 640, 1024, 2048])
def test_other_outer_block_sizes(nb):
    """contexts with another outer block than the 512 every measurement uses: nb = 384 / 640 are not 64 * 2^p (the diagonal-block
    inverses fall back to the block-substitution routine), 1024 is the widest block of the fused backward sweep, 2048 takes the
    two-launch sweep.  alpha against numpy's solve, the residual, and the variance against the default context's [r3]"""
    rng = np.random.default_rng(5)
    n = 3000
    x = rng.random((n, 3))
    f = np.sin(3 * x[:, 0]) + x[:, 1]
    k, nug = mgpi.SquaredExponentialKernel(1.0, 0.2, 0.2, 0.2), mgpi.WhiteNoiseKernel(1e-2)
    K = mgpi.Interpolator._x2cov(x, x, k + nug)
    ref = np.linalg.solve(K, f)
    q = x[:200] + 0.01
    mean0, var0 = mgpi.Interpolator(k, nugget=nug).condition_diag(q, x, f)
    it = mgpi.Interpolator(k, nugget=nug, context=mgpi.Context(nb=nb))
    a = it.compress(x, f)
    assert np.max(np.abs(a - ref)) < 1e-9 * np.max(np.abs(ref))
    assert np.max(np.abs(K @ a - f)) < 1e-12 * max(1.0, np.max(np.abs(f)))
    mean, var = it.condition_diag(q, x, f)
    assert np.max(np.abs(mean - mean0)) < 1e-11 * np.max(np.abs(mean0))
    assert np.max(np.abs(var - var0)) < 1e-11


@pytest.mark.parametrize('n', [700, 5000, 9000])
def test_backward_sweep_by_inverted_blocks_against_substitution(ctx, n):
    """[r3] the backward sweep of a single-right-hand-side solve takes the inverted diagonal blocks (trsv_bwd_dinv_kernel) when the
    factorisation's status read-back says the factor's diagonal spans at most 1e4, the substitution kernels otherwise - among
    others for a factor whose status this library never read.  A second handle over the SAME device memory (mgpi_mat_wrap: range
    unknown) therefore solves by substitution: the two alphas agree to 1e-13 of |alpha| and both leave a residual below 1e-11."""
    rng = np.random.default_rng(n)
    x = rng.random((n, 3))
    f = np.sin(3 * x[:, 0]) + x[:, 1] * x[:, 2]
    terms = (mgpi.SquaredExponentialKernel(1.0, 0.2, 0.2, 0.2) + mgpi.WhiteNoiseKernel(1e-2)).terms()
    desc, count = _lib.make_terms(terms)
    X = _lib.DeviceMatrix.from_host(ctx, x)
    K = _lib.DeviceMatrix(ctx, n, n, wide=True)
    _lib.check(ctx.lib.mgpi_cov_matrix_dev(ctx.handle, X.handle, None, desc, count, 1, K.handle))
    _lib.check(ctx.lib.mgpi_potrf_dev(ctx.handle, K.handle, None))
    rows, cols, ld = K.shape3()
    fpad = np.zeros(ld)                                     # the wrapped handle is the PADDED square (unit diagonal beyond n)
    fpad[:n] = f
    a1, a2 = _lib.DeviceMatrix.from_host(ctx, f), _lib.DeviceMatrix.from_host(ctx, fpad)
    _lib.check(ctx.lib.mgpi_potrs_dev(ctx.handle, K.handle, a1.handle))
    h = ctypes.c_void_p()
    _lib.check(ctx.lib.mgpi_mat_wrap(ctx.handle, K.devptr, ld, ld, ld, 1, ctypes.byref(h)))
    try:
        _lib.check(ctx.lib.mgpi_potrs_dev(ctx.handle, h, a2.handle))
    finally:
        ctx.lib.mgpi_mat_free(h)
    b1, b2 = a1.to_host().reshape(-1), a2.to_host().reshape(-1)
    assert np.all(b2[n:] == 0.0)
    b2 = b2[:n]
    assert np.all(np.isfinite(b1)) and np.all(np.isfinite(b2))
    assert np.max(np.abs(b1 - b2)) < 1e-13 * np.max(np.abs(b2))
    if n <= 5000:
        Kh = oracle.x2cov_chunked([oracle.term_se(1.0, 0.2, 0.2, 0.2), oracle.term_white(1e-2)], x, x, rows=max(1, 262144 // n))
        for b in (b1, b2):
            assert np.max(np.abs(Kh @ b - f)) < 1e-11 * max(1.0, np.max(np.abs(f)))


@pytest.mark.parametrize('env', [{'MGPI_NO_LOOKAHEAD': '1'}, {'MGPI_POOL_MB': '0'},
                                 {'MGPI_SUPER_PANELS': '1', 'MGPI_SUPER_MIN': '512'}, {'MGPI_SUPER_PANELS': '2', 'MGPI_SUPER_MIN': '512'},
                                 {'MGPI_VAR_DINV': '0'}, {'MGPI_TRSM_LOOKAHEAD': '1', 'MGPI_TRSM_SUPER_MIN': '512'},
                                 {'MGPI_TRSM_SUPER': '1', 'MGPI_TRSM_SUPER_MIN': '512'}, {'MGPI_TRSM_SUPER': '4', 'MGPI_TRSM_SUPER_MIN': '512'},
                                 {'MGPI_PANEL_STEP_ROWS': '0'},
                                 {'MGPI_PANEL_STEP_ROWS': '1000000', 'MGPI_SUPER_MIN': '512'},
                                 {'MGPI_POTRF_MERGED': '0'}, {'MGPI_TAIL_ROWS': '0'}, {'MGPI_TAIL_ROWS': '4096', 'MGPI_SUPER_MIN': '512'},
                                 {'MGPI_PANEL_INNER': '512'}, {'MGPI_PANEL_INNER': '128', 'MGPI_PANEL_INNER_ROWS': '0'},
                                 {'MGPI_PANEL_INNER': '256', 'MGPI_PANEL_INNER_ROWS': '0', 'MGPI_MERGED_MIN': '0'},
                                 {'MGPI_MERGED_MIN': '0', 'MGPI_SUPER_MIN': '512'},
                                 {'MGPI_COV_FAST': '0'}, {'MGPI_PREDICT_MFMA': '1'}, {'MGPI_GEMM_SKINNY_TILES': '0'}, {'MGPI_BCAST': 'rccl'},
                                 {'AMD_SERIALIZE_KERNEL': '3', 'MGPI_MERGED_MIN': '0', 'MGPI_SUPER_MIN': '512'},
                                 {'HIP_LAUNCH_BLOCKING': '1', 'MGPI_MERGED_MIN': '0', 'MGPI_SUPER_MIN': '512'},
                                 {'GPU_MAX_HW_QUEUES': '1', 'MGPI_MERGED_MIN': '0', 'MGPI_SUPER_MIN': '512'},
                                 {'MGPI_LL_CHUNK_BYTES': '12000000'}, {'MGPI_TRSV_DINV': '0'}, {}],
                         ids=lambda e: ','.join('%s=%s' % kv for kv in e.items()) or 'defaults')
def test_developer_switch_variants(env):
    """the schedule thresholds and the few alternative forms that ship behind an environment switch (DESIGN.md section c)
    compute the same results: no look-ahead, no buffer pool, 1 / 2 panels per trailing update (with a threshold small enough
    for the two-level schedule to engage), the panel step as one launch everywhere / split everywhere, the separate
    column update instead of the merged trailing update, tail and sub-panel thresholds, the general kernel-term loop, the
    MFMA form of predict-mean, 128-wide tiles for skinny updates; environments that serialise kernels (AMD_SERIALIZE_KERNEL,
    HIP_LAUNCH_BLOCKING, one hardware queue) with a threshold that WOULD engage the merged trailing update - the library must
    notice and take the separate column update instead of stalling in its bounded wait; a log-likelihood batch split into
    chunks; the backward sweep of the solve by substitution instead of inverted diagonal blocks.  (The measured losers of rounds 1-2 - ping-pong and
    register-staged gemm, round-1 panel kernels, left-looking steps, matrix-pipe factorisation - are gone, round 3.)"""
    import os
    import subprocess
    import sys
    worker = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'variant_worker.py')
    out = subprocess.run([sys.executable, worker], env=dict(os.environ, **env), capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and 'VARIANT-OK' in out.stdout, out.stdout[-1500:] + out.stderr[-3000:]


@pytest.mark.parametrize('m,w', [(64, 64), (640, 512), (1536, 256), (192, 128)])
def test_trsm_right_block(ctx, m, w):
    """A <- A L^-T with a factored diagonal block (the panel solve of the block-cyclic driver): one launch per 64 columns
    in accumulator layout, against scipy; the strict upper triangle of L is never read"""
    from scipy.linalg import solve_triangular
    rng = np.random.default_rng(m + w)
    L = np.tril(rng.standard_normal((w, w))) * 0.2 + np.diag(1.0 + rng.random(w))
    A = rng.standard_normal((m, w))
    Lg = L + np.triu(np.full((w, w), np.nan), 1)
    dL, dA = dev(ctx, Lg), dev(ctx, A)
    _lib.check(ctx.lib.mgpi_mat_mark_factored(dL.handle, 1))
    _lib.check(ctx.lib.mgpi_trsm_right_dev(ctx.handle, dL.handle, dA.handle))
    want = solve_triangular(L, A.T, lower=True).T
    assert relmax(dA.to_host(), want) < 1e-12
