#!/usr/bin/env python3
"""backward error of the single-rhs solve with inverted diagonal blocks against plain substitution, as the conditioning grows
(developer probe behind MGPI_TRSV_DINV_RATIO): python3 tools/dinv_accuracy_probe.py"""
import os, sys, subprocess
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

if len(sys.argv) > 1:      # child: one nugget, whatever form the environment selects
    import mgpi_amd as mgpi
    nug = float(sys.argv[1])
    rng = np.random.default_rng(7)
    n = 4000
    x = rng.random((n, 3))
    f = np.sin(3 * x[:, 0]) + x[:, 1] * x[:, 2]
    k, w = mgpi.SquaredExponentialKernel(1.0, 0.3, 0.3, 0.3), mgpi.WhiteNoiseKernel(nug)
    K = mgpi.Interpolator._x2cov(x, x, k + w)
    ev = np.linalg.eigvalsh(K)
    L = np.linalg.cholesky(K)
    d = np.diag(L)
    ctx = mgpi.Context()
    ctx.set_pivot_tol(0.0)
    a = mgpi.Interpolator(k, nugget=w, context=ctx).compress(x, f)
    back = np.max(np.abs(K @ a - f)) / (np.max(np.abs(K).sum(axis=1)) * np.max(np.abs(a)) + np.max(np.abs(f)))
    ref = np.linalg.solve(K, f)
    print('nugget %.0e: cond K %.1e, max/min L_ii %.1e | backward error %.2e, alpha vs numpy %.2e' % (
        nug, ev[-1] / ev[0], d.max() / d.min(), back, np.max(np.abs(a - ref)) / np.max(np.abs(ref))), flush=True)
    sys.exit(0)
for nug in (1e-2, 1e-3, 3e-4, 1e-4, 3e-5):
    for form, env in (('inverse if the gate allows', {}), ('inverse forced', {'MGPI_TRSV_DINV_RATIO': '1e300'}), ('substitution', {'MGPI_TRSV_DINV': '0'})):
        out = subprocess.run([sys.executable, os.path.abspath(__file__), repr(nug)], env=dict(os.environ, **env), capture_output=True, text=True)
        print('%-28s %s' % (form, (out.stdout.strip() or out.stderr.strip()[-300:])), flush=True)
