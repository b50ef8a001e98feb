#!/usr/bin/env python3
"""Generate golden vectors by running the REAL reference (read-only at /root/reference).

Run in the build container only:   python3 tests/golden/make_fixtures.py
The reference cannot travel to the GPU box, so its outputs are committed as small .npz
files next to this script.  Each file holds inputs (x, f, xs, kernel/nugget term spec) and
the reference's outputs (K, alpha, predict mean, condition mean/cov or diag, loglikelihood).

Term spec encoding (shared with oracle/mgpi_oracle.py and include/mgpi_hip.h):
  *_types  int array   0 = SquaredExponentialKernel, 1 = WhiteNoiseKernel
  *_sigmas float array
  *_lengths (nterms, D) float array (unused rows zero)
Scenarios follow SURVEY.md section 8c.
"""
import os
import sys
import time

import numpy as np

REFERENCE = os.environ.get('MGPI_REFERENCE', '/root/reference')
sys.path.insert(0, REFERENCE)
import mgpi  # noqa: E402  (the unmodified reference)

HERE = os.path.dirname(os.path.abspath(__file__))

SE, WHITE = 0, 1


def build(terms):
    """terms: list of ('se', sigma, lengths) / ('white', sigma) -> reference kernel object"""
    if not terms:
        return None
    objs = []
    for t in terms:
        if t[0] == 'se':
            objs.append(mgpi.SquaredExponentialKernel(t[1], *t[2]))
        else:
            objs.append(mgpi.WhiteNoiseKernel(t[1]))
    return objs[0] if len(objs) == 1 else mgpi.CombinedKernel(*objs)


def encode(terms, ndim, prefix):
    types = np.array([SE if t[0] == 'se' else WHITE for t in terms], dtype=np.int64)
    sigmas = np.array([t[1] for t in terms], dtype=float)
    lengths = np.zeros((len(terms), ndim))
    for i, t in enumerate(terms):
        if t[0] == 'se':
            lengths[i] = t[2]
    return {prefix + '_types': types, prefix + '_sigmas': sigmas, prefix + '_lengths': lengths}


def run(name, x, f, xs, kernel, nugget, full_cov=True, store_K=True):
    t0 = time.time()
    x = np.ascontiguousarray(x, dtype=float)
    xs = np.ascontiguousarray(xs, dtype=float)
    f = np.ascontiguousarray(f, dtype=float)
    ndim = x.shape[1]
    kobj, nobj = build(kernel), build(nugget)
    interp = mgpi.Interpolator(kobj, nugget=nobj)

    out = dict(x=x, f=f, xs=xs)
    out.update(encode(kernel, ndim, 'kernel'))
    out.update(encode(nugget, ndim, 'nugget'))

    both = kobj if nobj is None else kobj + nobj
    K = mgpi.Interpolator._x2cov(x, x, both)
    Ks = mgpi.Interpolator._x2cov(xs, x, kobj)
    if store_K:
        out['K'] = K
        out['Ks'] = Ks
    out['K_checksum'] = np.array([K.sum(), np.abs(K).sum(), np.trace(K)])
    out['cond_K'] = np.array(np.linalg.cond(K))

    try:
        alpha = interp.compress(x, f)
    except np.linalg.LinAlgError:
        # duplicated source rows make K exactly singular under the reference's exact-equality
        # white noise (kernels.py:131): np.linalg.inv raises.  Only K / Ks are golden then.
        out['singular'] = np.array(True)
        alpha = None
    if alpha is not None:
        out['singular'] = np.array(False)
        out['alpha'] = alpha
        out['predict_mean'] = interp.predict(xs, x, alpha)

        mean, cov = interp.condition(xs, x, f)
        out['cond_mean'] = mean
        if full_cov:
            out['cond_cov'] = cov
        out['cond_var'] = np.diag(cov).copy()
        out['loglike'] = np.array(interp.loglikelihood(x, f))

    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print('%-28s N=%-5d M=%-5d D=%d cond(K)=%.2e  %.1fs  %.0f kB' %
          (name, len(x), len(xs), ndim, out['cond_K'], time.time() - t0, os.path.getsize(path) / 1e3))


def run_matern(name, x, xs, order, sigma, lengths, extra=()):
    """Matern kernel matrices from the reference: K(x,x) (NaN on the diagonal - the reference's 0*inf, SURVEY.md
    section 0) and K(xs,x) with no coincident points.  `extra`: further ('se'|'white', ...) summands."""
    kobj = mgpi.MaternKernel(order, sigma, *lengths)
    objs = [kobj] + [build([t]) for t in extra]
    kern = objs[0] if len(objs) == 1 else mgpi.CombinedKernel(*objs)
    with np.errstate(invalid='ignore'):
        K = mgpi.Interpolator._x2cov(x, x, kern)
        Ks = mgpi.Interpolator._x2cov(xs, x, kern)
    out = dict(x=x, xs=xs, K=K, Ks=Ks, order=np.array(order), sigma=np.array(sigma), lengths=np.array(lengths, dtype=float))
    out.update(encode(list(extra), x.shape[1], 'extra'))
    path = os.path.join(HERE, name + '.npz')
    np.savez_compressed(path, **out)
    print('%-28s N=%-5d M=%-5d D=%d order=%g  NaN entries in K: %d' % (name, len(x), len(xs), x.shape[1], order, np.isnan(K).sum()))


def pad_neighbors(neighbors, k):
    idx = -np.ones((len(neighbors), k), dtype=np.int32)
    for i, nb in enumerate(neighbors):
        idx[i, :len(nb)] = nb
    return idx


def nngp_loglike_longdouble(kernel, nugget, sx, sf, neighbors):
    """the NNGP loglikelihood (interpolators.py:833-846) on the reference's own neighbour sets, with every per-point
    conditional worked out by a Cholesky in 80-bit extended precision.  NOT a reference output: a yardstick for it.
    The reference's LU-inverse loses ~cond * eps in each conditional variance and the loglikelihood divides by those
    variances, so its own value carries an error of 1e-9..1e-8 relative (measured against this number), four orders
    above what a fp64 Cholesky delivers; the parity test needs to know which of the two it is looking at."""
    LD = np.longdouble

    def cov(terms, a, b):
        tot = LD(0)
        for t in terms:
            if t[0] == 'se':
                tot += LD(t[1])**2 * np.exp(-sum(((LD(u) - LD(v)) / LD(l))**2 for u, v, l in zip(a, b, t[2])))
            else:
                tot += LD(t[1])**2 * (1 if all(u == v for u, v in zip(a, b)) else 0)
        return tot

    both = list(kernel) + list(nugget)
    ll = LD(0)
    for i in range(len(sx)):
        idx = neighbors[i]
        c = len(idx)
        if c == 0:
            mean, diag = LD(0), cov(both, sx[i], sx[i])
        else:
            K = np.array([[cov(both, sx[a], sx[b]) for b in idx] for a in idx], dtype=LD)
            ks = np.array([cov(kernel, sx[i], sx[b]) for b in idx], dtype=LD)
            L = np.zeros((c, c), dtype=LD)
            v, w = np.zeros(c, dtype=LD), np.zeros(c, dtype=LD)
            for j in range(c):
                L[j, j] = np.sqrt(K[j, j] - np.sum(L[j, :j]**2))
                for r in range(j + 1, c):
                    L[r, j] = (K[r, j] - np.sum(L[r, :j] * L[j, :j])) / L[j, j]
                v[j] = (ks[j] - np.sum(L[j, :j] * v[:j])) / L[j, j]
                w[j] = (LD(sf[idx[j]]) - np.sum(L[j, :j] * w[:j])) / L[j, j]
            mean, diag = np.sum(v * w), cov(kernel, sx[i], sx[i]) - np.sum(v * v)
        ll += -(mean - LD(sf[i]))**2 / diag / 2 - np.log(diag) / 2
    return float(ll - LD(len(sx)) * np.log(2 * LD(np.pi)) / 2)


def run_nngp(name, x, f, xs, kernel, nugget, k, obi, tie_free=True):
    """NearestNeighborInterpolator (interpolators.py:612-871): sorted training set, neighbour lists (reference-set
    mode and target mode, -1 padded), per-point means/variances, loglikelihood, condition"""
    x = np.ascontiguousarray(x, dtype=float)
    xs = np.ascontiguousarray(xs, dtype=float)
    f = np.ascontiguousarray(f, dtype=float)
    nn = mgpi.NearestNeighborInterpolator(build(kernel), nugget=build(nugget), num_neighbors=k, order_by_index=obi)
    sx, sf = nn._2sorted(x, source_f=f)
    nb_self = nn._2neighbors(sx)
    nb_tar = nn._2neighbors(sx, target_x=xs)
    self_mean, self_diag = nn._2diag(sx, sx, sf, nb_self)
    mean, cov = nn.condition(xs, x, f)
    out = dict(x=x, f=f, xs=xs, k=np.array(k), obi=np.array(-1 if obi is None else obi), tie_free=np.array(tie_free),
               sorted_x=sx, sorted_f=sf, nb_self=pad_neighbors(nb_self, k), nb_target=pad_neighbors(nb_tar, k),
               self_mean=self_mean, self_diag=self_diag, cond_mean=mean, cond_diag=np.diag(cov).copy(),
               loglike=np.array(nn.loglikelihood(x, f)),
               loglike_longdouble=np.array(nngp_loglike_longdouble(kernel, nugget, sx, sf, nb_self)))
    out.update(encode(kernel, x.shape[1], 'kernel'))
    out.update(encode(nugget, x.shape[1], 'nugget'))
    np.savez_compressed(os.path.join(HERE, name + '.npz'), **out)
    print('%-28s N=%-5d M=%-5d D=%d k=%d obi=%s empty target sets: %d   loglike %.9f (80-bit Cholesky: %.9f)' %
          (name, len(x), len(xs), x.shape[1], k, obi, sum(len(n) == 0 for n in nb_tar), out['loglike'], out['loglike_longdouble']))


def nngp_cases():
    rng = np.random.default_rng(31)
    x = rng.random((400, 2))
    f = np.sin(4 * x[:, 0]) * np.cos(3 * x[:, 1]) + 0.3 * x[:, 1]
    xs = rng.random((80, 2))
    xs[0] = x[17]                       # coincides with a training point: that point is excluded from its neighbours
    xs[1] = [-1.0, -1.0]                # ranked below every training point: no neighbours at all
    xs[2] = [2.0, 2.0]                  # ranked above everything
    se, wn = ('se', 0.8, (0.25, 0.2)), ('white', 0.8e-3)
    run_nngp('nngp_rand2d_k10', x, f, xs, [se], [wn], 10, None)
    run_nngp('nngp_rand2d_obi0_k7', x, f, xs, [se, wn], [], 7, 0)
    run_nngp('nngp_rand2d_obi1_k14', x, f, xs, [se], [wn], 14, 1)
    run_nngp('nngp_rand2d_k15', x, f, xs, [se], [wn], 15, None)
    run_nngp('nngp_rand2d_k30', x, f, xs, [se], [('white', 3e-3)], 30, None)
    run_nngp('nngp_rand2d_k40', x, f, xs, [se], [('white', 3e-3)], 40, None)
    x3 = rng.random((300, 3))
    f3 = np.sin(3 * x3[:, 0]) + x3[:, 1] * x3[:, 2]
    run_nngp('nngp_rand3d_k5', x3, f3, rng.random((40, 3)), [('se', 1.0, (0.4, 0.4, 0.4))], [('white', 1e-2)], 5, None)
    run_nngp('nngp_fewer_than_k', x[:6], f[:6], xs[:9], [se], [wn], 10, None)
    x9 = rng.random((150, 9))           # D >= 8: numpy's pairwise summation order in the rank and the distance
    run_nngp('nngp_rand9d_k6', x9, np.sin(x9.sum(axis=1)), rng.random((20, 9)), [('se', 1.0, (1.5,) * 9)], [('white', 1e-2)], 6, None)
    # regular grid, checkerboard-thinned like test/uiuc/uiuc-test-nngp-timing:117-150 - distances tie everywhere
    g = np.linspace(0.0, 1.0, 16)
    G0, G1 = np.meshgrid(g, g, indexing='ij')
    keep = ~((np.arange(16)[:, None] % 2 == 1) & (np.arange(16)[None, :] % 2 == 1))
    xg = np.transpose([G0[keep], G1[keep]])
    fg = np.tanh(3 * (xg[:, 1] - 0.5)) + 0.1 * np.sin(5 * xg[:, 0])
    sig = 1.5 * np.std(fg)
    ell = (2 * (g[1] - g[0]),) * 2
    run_nngp('nngp_grid2d_ties_k10', xg, fg, np.transpose([G0[~keep], G1[~keep]]), [('se', sig, ell), ('white', sig / 1000)], [], 10, 0,
             tie_free=False)


def smooth(x):
    """a smooth scalar field on [0,1]^D-ish inputs"""
    x = np.asarray(x, dtype=float)
    w = np.arange(1, x.shape[1] + 1)
    return np.sin(2.0 * x @ w / len(w)) + 0.5 * np.cos(3.0 * x[:, 0]) * np.prod(np.cos(x[:, 1:]), axis=1)


def main():
    # (1) toys: N in {1,2,3,17,64}, D in {1,2,3,4}; duplicated source rows; targets on sources
    rng = np.random.default_rng(1)
    for n, d in [(1, 1), (2, 2), (3, 3), (17, 4), (64, 2), (64, 3)]:
        x = rng.random((n, d))
        f = smooth(x)
        xs = np.vstack([rng.random((5, d)), x[:min(n, 3)]])          # last rows coincide with sources
        ell = tuple(0.3 + 0.1 * i for i in range(d))
        # white noise INSIDE the kernel (how every reference test script uses it)
        run('toy_n%d_d%d_whitein' % (n, d), x, f, xs,
            [('se', 1.3, ell), ('white', 1e-2)], [])
        # white noise as nugget
        run('toy_n%d_d%d_nugget' % (n, d), x, f, xs,
            [('se', 1.3, ell)], [('white', 1e-2)])

    # duplicated source rows: the exact-equality nugget puts sigma_n^2 OFF the diagonal too
    x = rng.random((12, 2))
    x[7] = x[2]
    x[11] = x[2]
    f = smooth(x)
    f[7] += 1e-3
    xs = np.vstack([rng.random((4, 2)), x[2:3]])
    run('toy_duplicates', x, f, xs, [('se', 0.9, (0.35, 0.25))], [('white', 0.05)])

    # two SE terms + two white terms (CombinedKernel flattening, kernels.py:54-59)
    x = rng.random((40, 3))
    run('toy_multiterm', x, smooth(x), rng.random((9, 3)),
        [('se', 1.0, (0.3, 0.4, 0.5)), ('se', 0.2, (0.05, 0.07, 0.06)), ('white', 3e-3)],
        [('white', 1e-2)])

    # no white noise at all, well separated points (K well conditioned on its own)
    x = rng.random((30, 2)) * 4.0
    run('toy_nonugget', x, smooth(x), rng.random((6, 2)) * 4.0, [('se', 1.0, (0.5, 0.5))], [])

    nngp_cases()

    # Matern (kernels.py:135-162): etc/example-interpolator.ini's Kernel3 (order 2, sigma 0.1, lengths 15 10) on a
    # (muB, T)-like scatter, half-integer and non-integer orders, tiny and huge separations
    rngm = np.random.default_rng(11)
    xm = np.transpose([rngm.uniform(0, 150, 60), rngm.uniform(30, 100, 60)])
    xsm = np.transpose([rngm.uniform(0, 150, 25), rngm.uniform(30, 100, 25)])
    xsm[0] = xm[0] + [1e-7, 0.0]                       # nearly coincident
    xsm[1] = xm[1] + [400.0, 300.0]                    # far: K_o underflows towards 0
    run_matern('matern_o2_ini', xm, xsm, 2, 0.1, (15., 10.))
    run_matern('matern_o0p5', xm, xsm, 0.5, 1.3, (20., 12.))
    run_matern('matern_o1p5', xm, xsm, 1.5, 0.7, (20., 12.))
    run_matern('matern_o2p7_plus_se', xm, xsm, 2.7, 0.7, (25., 15.), extra=[('se', 0.1, (10., 20.)), ('white', 1e-3)])
    x3 = rngm.random((40, 3))
    run_matern('matern_o1_d3', x3, rngm.random((12, 3)), 1, 2.0, (0.3, 0.4, 0.5))

    # (2) uiuc-test style: 2-D grid, checkerboard split, sigma = 1.5 std f, ell = 2*spacing,
    #     white sigma/1000 inside the CombinedKernel (test/uiuc/uiuc-test:114-189); mean subtracted (:203)
    g0 = np.linspace(0.0, 150.0, 24)
    g1 = np.linspace(30.0, 100.0, 24)
    G0, G1 = np.meshgrid(g0, g1, indexing='ij')
    full = np.transpose([G0.ravel(), G1.ravel()])
    ff = 0.3 + 0.1 * np.tanh((full[:, 1] - 60.0) / 15.0) + 0.02 * np.sin(full[:, 0] / 40.0)
    parity = (np.add.outer(np.arange(24), np.arange(24)) % 2).ravel().astype(bool)
    x, f, xs = full[parity], ff[parity], full[~parity][::3]
    f = f - np.mean(f)
    sig = 1.5 * np.std(f)
    ell = (2 * (g0[1] - g0[0]), 2 * (g1[1] - g1[0]))
    run('uiuc_checkerboard_whitein', x, f, xs, [('se', sig, ell), ('white', sig / 1000.)], [])
    run('uiuc_checkerboard_nugget', x, f, xs, [('se', sig, ell)], [('white', sig / 1000.)])

    # (3) sc-eos-test style: 3-D, sigma = 2 std f, ell = 2*spacing, nugget sigma/100
    #     (test/stellarcollapse-eos/sc-eos-test:132-148), regular target grid (:172-176)
    a = [np.linspace(0.05, 0.5, 6), np.linspace(4.0, 14.0, 9), np.linspace(17.0, 22.0, 8)]
    A = np.meshgrid(*a, indexing='ij')
    x = np.transpose([_.ravel() for _ in A])
    f = 0.6 * (x[:, 2] - 17.0) - 0.05 * (x[:, 1] - 9.0) + 0.5 * np.sin(4 * x[:, 0])
    f = f - np.mean(f)
    sig = 2 * np.std(f)
    ell = tuple(2 * (_[1] - _[0]) for _ in a)
    b = [np.linspace(_[0], _[-1], 5) for _ in a]
    B = np.meshgrid(*b, indexing='ij')
    xs = np.transpose([_.ravel() for _ in B])
    run('sceos_3d', x, f, xs, [('se', sig, ell)], [('white', sig / 100.)])

    # (4)/(5) C1: etc/example-interpolator.ini values (SE sigma=0.1, ell=(10,20); nugget 1e-4; the
    #     Matern term dropped) on a synthetic (muB, T) table: N=512, D=2
    rng = np.random.default_rng(0)
    g0 = np.linspace(0.0, 150.0, 32)
    g1 = np.linspace(30.0, 100.0, 32)
    G0, G1 = np.meshgrid(g0, g1, indexing='ij')
    full = np.transpose([G0.ravel(), G1.ravel()])
    ff = 0.05 * np.tanh((full[:, 1] - 60.0) / 20.0) + 0.01 * np.sin(full[:, 0] / 30.0) + 1e-4 * rng.standard_normal(len(full))
    x, f = full[::2], ff[::2]                     # io.py:113-120 row-stride downsample = 2 -> 512 rows
    xs = np.transpose([rng.uniform(0, 150, 256), rng.uniform(30, 100, 256)])
    run('c1_n512_d2', x, f, xs, [('se', 0.1, (10., 20.))], [('white', 1e-4)])

    # (6) larger direct-oracle cases: full K not stored
    for n, seed in [(2048, 6), (4096, 7)]:
        rng = np.random.default_rng(seed)
        side = np.array([16, 16, n // 256])
        axes = [(np.arange(s) + 0.5) / s for s in side]
        A = np.meshgrid(*axes, indexing='ij')
        x = np.transpose([_.ravel() for _ in A])
        x = x + rng.uniform(-0.1, 0.1, x.shape) / side
        f = np.sin(3 * x[:, 0]) * np.cos(2 * x[:, 1]) + 0.5 * np.sin(4 * x[:, 2])
        xs = rng.random((256, 3))
        ell = tuple(2.0 / side)
        run('grid_n%d_d3' % n, x, f, xs, [('se', 1.0, ell)], [('white', 1e-2)],
            full_cov=False, store_K=False)


if __name__ == '__main__':
    main()
