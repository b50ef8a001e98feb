"""One process per developer switch (tests/test_gpu_kernels.py::test_developer_switch_variants): the environment
variable is read once when the library first launches a gemm, so every variant needs a fresh process.  Checks the
128x128 gemm forms with integer-valued operands (exact in fp64), the NN form, and a factorisation + solve."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import mgpi_amd as mgpi
from mgpi_amd import _lib

ctx = mgpi.Context(device=0)
lib, h = ctx.lib, ctx.handle
rng = np.random.default_rng(0)


def dev(a):
    return _lib.DeviceMatrix.from_host(ctx, a, wide=True)


for (m, n, k) in [(256, 384, 512), (640, 640, 48), (128, 128, 16)]:
    A = rng.integers(-3, 4, size=(m, k)).astype(float)
    B = rng.integers(-3, 4, size=(n, k)).astype(float)
    C0 = rng.integers(-5, 6, size=(m, n)).astype(float)
    dA, dB, dC = dev(A), dev(B), dev(C0)
    _lib.check(lib.mgpi_gemm_nt_dev(h, dA.handle, dB.handle, dC.handle, 0))
    assert np.array_equal(dC.to_host(), C0 - A @ B.T), ('NT', m, n, k)
    dBn, dC2 = dev(np.ascontiguousarray(B.T)), dev(C0)
    _lib.check(lib.mgpi_gemm_nn_dev(h, dA.handle, dBn.handle, dC2.handle))
    assert np.array_equal(dC2.to_host(), C0 - A @ B.T), ('NN', m, n, k)
    if m == n:
        dC3 = dev(C0)
        _lib.check(lib.mgpi_gemm_nt_dev(h, dA.handle, dA.handle, dC3.handle, 1))
        got, want = dC3.to_host(), C0 - A @ A.T
        for ti in range(m // 128):
            for tj in range(n // 128):
                blk = (slice(128 * ti, 128 * ti + 128), slice(128 * tj, 128 * tj + 128))
                assert np.array_equal(got[blk], want[blk] if tj <= ti else C0[blk]), ('lower', ti, tj)

# the factor itself against LAPACK, and the index of the first failing pivot (whatever form the 64 x 64 sweeps take)
for nn in (65, 200, 513):
    xx = rng.random((nn, 2))
    Kk = mgpi.Interpolator._x2cov(xx, xx, mgpi.SquaredExponentialKernel(1.0, 0.2, 0.3) + mgpi.WhiteNoiseKernel(0.05))
    dK = dev(Kk)
    _lib.check(lib.mgpi_potrf_dev(h, dK.handle, None))
    Lg, Lr = np.tril(dK.to_host()), np.linalg.cholesky(Kk)
    assert np.max(np.abs(Lg - Lr)) / np.max(np.abs(Lr)) < 1e-11, ('potrf vs lapack', nn)
    assert np.max(np.abs(Lg @ Lg.T - Kk)) / np.max(np.abs(Kk)) < 1e-14, ('L L^T', nn)
Kb = np.eye(300)
Kb[150, 150] = -1.0
dKb = dev(Kb)
try:
    _lib.check(lib.mgpi_potrf_dev(h, dKb.handle, None))
    raise SystemExit('an indefinite matrix was accepted')
except mgpi.NotPositiveDefinite as err:
    assert err.pivot == 151, err.pivot

n = 1700
x = rng.random((n, 3))
f = np.sin(3 * x[:, 0]) + x[:, 1] * x[:, 2]
it = mgpi.Interpolator(mgpi.SquaredExponentialKernel(1.0, 0.2, 0.2, 0.2), nugget=mgpi.WhiteNoiseKernel(1e-2), context=ctx)
K = mgpi.Interpolator._x2cov(x, x, it.kernel + it.nugget)
alpha = it.compress(x, f)
assert np.max(np.abs(K @ alpha - f)) < 1e-10 * np.max(np.abs(f))
mean, var = it.condition_diag(x[:300] + 0.01, x, f)
Ks300 = mgpi.Interpolator._x2cov(x[:300] + 0.01, x, it.kernel)
assert np.max(np.abs(mean - Ks300 @ alpha)) < 1e-10 * np.max(np.abs(mean))
ll = it.loglikelihood(x, f)
s, logdet = np.linalg.slogdet(K)
assert abs(ll - (-0.5 * f @ alpha - 0.5 * logdet - 0.5 * n * np.log(2 * np.pi))) < 1e-9 * abs(ll)
assert np.all(var > 0) and np.all(var < 1.0)
# a batch of kernel-parameter sets against one-at-a-time calls (MGPI_LL_CHUNK_BYTES makes the library split the batch)
rows = np.array([[1.0, 0.2, 0.2, 0.2], [1.3, 0.25, 0.2, 0.3], [0.7, 0.15, 0.3, 0.2], [1.1, 0.3, 0.3, 0.3], [0.9, 0.2, 0.25, 0.2]])
vals, piv = it.loglikelihoods(x[:700], f[:700], rows)
for row, v in zip(rows, vals):
    it.update(*row)
    assert it.loglikelihood(x[:700], f[:700]) == v, ('batched loglikelihood', row)
assert not piv.any()
it.update(*rows[0])
if os.environ.get('MGPI_SUPER_MIN') or os.environ.get('MGPI_TRSM_SUPER_MIN'):
    # large enough for the two-level schedule to engage with the threshold the test sets; residual check only
    n = 4600
    x = rng.random((n, 3))
    f = np.sin(3 * x[:, 0]) + x[:, 1] * x[:, 2]
    it = mgpi.Interpolator(mgpi.SquaredExponentialKernel(1.0, 0.15, 0.15, 0.15), nugget=mgpi.WhiteNoiseKernel(1e-2), context=ctx)
    K = mgpi.Interpolator._x2cov(x, x, it.kernel + it.nugget)
    alpha = it.compress(x, f)
    assert np.max(np.abs(K @ alpha - f)) < 1e-10 * np.max(np.abs(f))
    s, logdet = np.linalg.slogdet(K)
    ll = it.loglikelihood(x, f)
    assert abs(ll - (-0.5 * f @ alpha - 0.5 * logdet - 0.5 * n * np.log(2 * np.pi))) < 1e-9 * abs(ll)
    xs = rng.random((700, 3))
    mean, var = it.condition_diag(xs, x, f)
    Ks = mgpi.Interpolator._x2cov(xs, x, it.kernel)
    Lc = np.linalg.cholesky(K)
    from scipy.linalg import solve_triangular
    V = solve_triangular(Lc, Ks.T, lower=True)
    assert np.max(np.abs(mean - Ks @ alpha)) < 1e-10 * np.max(np.abs(mean))
    assert np.max(np.abs(var - (1.0 - np.sum(V * V, axis=0)))) < 1e-9
ctx.trim()
print('VARIANT-OK', flush=True)
