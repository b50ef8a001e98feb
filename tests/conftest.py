import glob
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def _all_golden():
    return sorted(os.path.splitext(os.path.basename(p))[0] for p in glob.glob(os.path.join(GOLDEN, '*.npz')))


def golden_names():
    """the full-path fixtures (kernel matrices + compress/predict/condition/loglikelihood outputs)"""
    return [n for n in _all_golden() if not n.startswith(('matern_', 'nngp_'))]


def nngp_names():
    """NearestNeighborInterpolator fixtures (sorted set, neighbour lists, per-point means/variances, loglikelihood)"""
    return [n for n in _all_golden() if n.startswith('nngp_')]


def neighbor_lists(padded):
    """-1 padded (M, k) index array -> list of index arrays, the reference's return type"""
    return [row[row >= 0] for row in np.asarray(padded)]


def matern_names():
    """kernel-matrix-only fixtures of the reference's MaternKernel (its K(x,x) has a NaN diagonal)"""
    return [n for n in _all_golden() if n.startswith('matern_')]


def matern_terms(g, flavour):
    """a matern_* fixture -> oracle term tuples (flavour 'oracle') or an mgpi_amd kernel object ('product')"""
    order, sigma, lengths = float(g['order']), float(g['sigma']), [float(v) for v in g['lengths']]
    if flavour == 'oracle':
        from oracle import mgpi_oracle as oracle
        return [oracle.term_matern(order, sigma, *lengths)] + oracle_terms(g, 'extra')
    import mgpi_amd as mgpi
    first = mgpi.MaternKernel(order, sigma, *lengths)
    rest = product_kernel(g, 'extra')
    return first if rest is None else first + rest


def load_golden(name):
    """inputs + reference outputs written by tests/golden/make_fixtures.py (numpy.load, no pickle)"""
    with np.load(os.path.join(GOLDEN, name + '.npz'), allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def oracle_terms(g, prefix):
    """term spec of a fixture -> oracle term tuples"""
    from oracle import mgpi_oracle as oracle
    out = []
    for kind, sigma, lengths in zip(g[prefix + '_types'], g[prefix + '_sigmas'], g[prefix + '_lengths']):
        out.append(oracle.term_se(sigma, *lengths) if kind == 0 else oracle.term_white(sigma))
    return out


def product_kernel(g, prefix):
    """term spec of a fixture -> mgpi_amd kernel object (None when there are no terms)"""
    import mgpi_amd as mgpi
    objs = []
    for kind, sigma, lengths in zip(g[prefix + '_types'], g[prefix + '_sigmas'], g[prefix + '_lengths']):
        objs.append(mgpi.SquaredExponentialKernel(sigma, *lengths) if kind == 0 else mgpi.WhiteNoiseKernel(sigma))
    if not objs:
        return None
    return objs[0] if len(objs) == 1 else mgpi.CombinedKernel(*objs)


def relmax(a, b):
    """max-norm relative difference, the norm SURVEY.md section 8c states tolerances in"""
    a = np.asarray(a, dtype=float)
    b = np.asarray(b, dtype=float)
    scale = np.max(np.abs(b))
    return float(np.max(np.abs(a - b)) / (scale if scale > 0 else 1.0))


def mean_tol(cond):
    """north_star tolerance for means: 1e-10 max-norm-relative while cond(K) <= 1e7;
    beyond that the reference's own LU inverse is only reproducible to ~cond*eps, so the bound grows with it"""
    return max(1e-10, 4e-17 * float(cond))


def var_tol(cond):
    """variance / covariance tolerance: 1e-8 while cond(K) <= 1e7, relative to the
    PRIOR variance k(x*,x*) - the quantity the quadratic form is subtracted from, which sets the size of the
    cancellation error (the posterior variance itself can be 1e4 times smaller, e.g. the C1 fixture)"""
    return max(1e-8, 4e-15 * float(cond))


def prior_var(g):
    """k(x*, x*) of a fixture's kernel: sum of sigma^2 over its summands (SE at zero distance + white)"""
    return float(np.sum(np.asarray(g['kernel_sigmas'], dtype=float)**2))
