import sys, os, numpy as np
sys.path.insert(0, os.environ.get('GRAFT_REPO_ROOT', '/root/repo'))
import mgpi_amd as mgpi
rng = np.random.default_rng(2026)
worst = 0.0
for trial in range(60):
    n = int(rng.integers(1, 7000)); d = int(rng.integers(1, 5)); m = int(rng.integers(1, 700))
    x = rng.random((n, d)); f = np.sin(3 * x[:, 0]) + x[:, -1]
    ell = tuple(rng.uniform(0.15, 0.5, d)); sn = float(10 ** rng.uniform(-3, -1))
    k = mgpi.SquaredExponentialKernel(float(rng.uniform(0.5, 2)), *ell); w = mgpi.WhiteNoiseKernel(sn)
    it = mgpi.Interpolator(k, nugget=w)
    K = mgpi.Interpolator._x2cov(x, x, k + w)
    try:
        a = it.compress(x, f)
    except np.linalg.LinAlgError as e:
        print('trial %d n=%d d=%d sn=%.1e: %s' % (trial, n, d, sn, str(e)[:60])); continue
    back = np.max(np.abs(K @ a - f)) / (np.max(np.abs(K).sum(axis=1)) * np.max(np.abs(a)) + np.max(np.abs(f)))
    xs = rng.random((m, d))
    mean, var = it.condition_diag(xs, x, f)
    Ks = mgpi.Interpolator._x2cov(xs, x, k)
    merr = np.max(np.abs(mean - Ks @ a)) / max(1e-300, np.max(np.abs(mean)))
    L = np.linalg.cholesky(K); V = np.linalg.solve(L, Ks.T) if n < 3000 else None
    verr = np.max(np.abs(var - (k.params[0] ** 2 - (V * V).sum(axis=0)))) if V is not None else 0.0
    worst = max(worst, back)
    flag = '' if (back < 1e-15 and merr < 1e-9 and verr < 1e-9) else '  <-- CHECK'
    print('trial %2d n=%4d d=%d m=%3d sn=%.1e: backward %.1e mean-consistency %.1e var %.1e%s' % (trial, n, d, m, sn, back, merr, verr, flag), flush=True)
print('worst backward error %.2e' % worst)
