"""NearestNeighborInterpolator: the reference's NNGP (mgpi/interpolators.py:612-871; Datta, Banerjee, Finley &
Gelfand 2016) with its two loops - the neighbour search (:654-744) and the per-point conditioning (:751-792) - as HIP
kernels (csrc/nngp.hip).  Same class name, constructor, helper methods and return types as the reference, so that
``factory(Interpolator)`` finds it by name and the reference's NNGP scripts run unchanged.

Host side (numpy, O(N log N)): the ranks and the sort of the training set - the same numpy calls as the reference,
so the order is identical, ties included.  Device side: everything quadratic or per-point.

One deliberate choice: where two candidate neighbours are at exactly the same distance the one with the lower index
in the sorted training set is preferred; the reference's ``np.argsort`` (:726) leaves that order unspecified.  The
two agree whenever the distances are distinct.
"""
import collections.abc
import ctypes
import time

import numpy as np

from . import _lib
from .gp import Interpolator, _flatten
from .hyper import DEFAULT_TEMPERATURE

__all__ = ['NearestNeighborInterpolator', 'NeighborSets', 'DEFAULT_NUM_NEIGHBORS', 'DEFAULT_ORDER_BY_INDEX']

DEFAULT_NUM_NEIGHBORS = 10       # interpolators.py:40
DEFAULT_ORDER_BY_INDEX = None    # interpolators.py:41


class NeighborSets(collections.abc.Sequence):
    """the neighbour sets of M points: behaves like the reference's list of index arrays (``len``, indexing,
    iteration, closest neighbour first) but keeps the packed (M, k) table the kernels read"""

    def __init__(self, index, count):
        self.index = np.ascontiguousarray(index, dtype=np.int32)
        self.count = np.ascontiguousarray(count, dtype=np.int32)
        assert self.index.ndim == 2 and self.count.shape == (len(self.index),)

    def __len__(self):
        return len(self.count)

    def __getitem__(self, i):
        if isinstance(i, slice):
            return NeighborSets(self.index[i], self.count[i])
        return self.index[i, :self.count[i]]

    @classmethod
    def pack(cls, neighbors, width=None):
        """any sequence of index sequences -> NeighborSets (what a caller of the reference may hand in, :833)"""
        if isinstance(neighbors, cls):
            return neighbors
        count = np.array([len(nb) for nb in neighbors], dtype=np.int32)
        width = max(1, int(count.max()) if len(count) else 1, width or 1)
        index = -np.ones((len(count), width), dtype=np.int32)
        for i, nb in enumerate(neighbors):
            index[i, :count[i]] = nb
        return cls(index, count)


class NearestNeighborInterpolator(Interpolator):
    """Nearest-Neighbor Gaussian Process: every point is conditioned on its ``num_neighbors`` closest predecessors in
    an ordering of the training set, which makes the precision matrix sparse and the cost linear in N"""

    def __init__(self, kernel, nugget=None, num_neighbors=DEFAULT_NUM_NEIGHBORS, order_by_index=DEFAULT_ORDER_BY_INDEX):
        self.num_neighbors = num_neighbors
        self.order_by_index = order_by_index
        Interpolator.__init__(self, kernel, nugget=nugget)

    #--- ordering (interpolators.py:633-652): host, the reference's own numpy calls

    def _2rank(self, x):
        if self.order_by_index is None:
            return np.sum(x)
        return x[self.order_by_index]

    def _2ranks(self, x):
        x = np.asarray(x)
        if self.order_by_index is not None:
            return np.array(x[:, self.order_by_index])
        # row by row np.sum: a contiguous copy makes the axis-1 reduction use the same (pairwise) order per row
        return np.add.reduce(np.ascontiguousarray(x), axis=1)

    def _2sorted(self, source_x, source_f=None):
        order = np.argsort(self._2ranks(source_x))
        if source_f is not None:
            source_f = source_f[order]
        return source_x[order], source_f

    #--- neighbour search (interpolators.py:654-744)

    def _check_width(self):
        k = int(self.num_neighbors)
        if not 1 <= k <= _lib.NN_MAX_NEIGHBORS:
            raise ValueError('num_neighbors=%d outside [1, %d]' % (k, _lib.NN_MAX_NEIGHBORS))
        return k

    def _2neighbors(self, source_x, target_x=None, verbose=False, Verbose=False):
        """neighbour sets of target_x (default: of source_x itself, each sample against its predecessors) within
        source_x, which is assumed sorted; returns NeighborSets (a sequence of index arrays, closest first)"""
        verbose |= Verbose
        k = self._check_width()
        ctx = self.context
        sx = _lib.as_f64(source_x, ndim=2)
        srank = np.ascontiguousarray(self._2ranks(sx), dtype=float)
        if target_x is None:
            if verbose:
                print('setting target_x = source_x --> finding neighbors within reference set')
            tx = trank = None
            m = len(sx)
            order = None
        else:
            tx = _lib.as_f64(target_x, ndim=2)
            assert tx.shape[1] == sx.shape[1], 'conflicting number of dimensions for x1 and x2'
            trank = np.ascontiguousarray(self._2ranks(tx), dtype=float)
            # hand the targets over in rank order: the threads of a block then scan the same stretch of the reference
            # set and stop together; the rows are put back in the caller's order below
            order = np.argsort(trank, kind='stable')
            tx, trank = np.ascontiguousarray(tx[order]), np.ascontiguousarray(trank[order])
            m = len(tx)
        if verbose:
            print('%d samples in reference set:' % len(sx))
        index = np.empty((m, k), dtype=np.int32)
        count = np.empty(m, dtype=np.int32)
        ptr = lambda a: None if a is None else a.ctypes.data_as(ctypes.c_void_p)
        _lib.check(ctx.lib.mgpi_nn_neighbors(ctx.handle, ptr(sx), ptr(srank), len(sx), sx.shape[1], ptr(tx), ptr(trank), m, k,
                                             int(self.order_by_index is None), ptr(index), ptr(count)))
        if order is not None:
            back = np.empty_like(order)
            back[order] = np.arange(m)
            index, count = index[back], count[back]
        return NeighborSets(index, count)

    #--- per-point conditioning (interpolators.py:751-792)

    def _2diag(self, target_x, source_x, source_f, neighbors, verbose=False, safe=True):
        """(2, M) array: predicted means and variances of target_x, each conditioned on its own neighbour set"""
        ctx = self.context
        tx = _lib.as_f64(target_x, ndim=2)
        sx = _lib.as_f64(source_x, ndim=2)
        sf = _lib.as_f64(source_f).reshape(-1)
        assert tx.shape[1] == sx.shape[1], 'conflicting number of dimensions for x1 and x2'
        assert len(sf) == len(sx), 'source_f has %d entries for %d source points' % (len(sf), len(sx))
        sets = NeighborSets.pack(neighbors)
        assert len(sets) == len(tx), '%d neighbor sets for %d target points' % (len(sets), len(tx))
        width = sets.index.shape[1]
        if width > _lib.NN_MAX_NEIGHBORS:
            raise ValueError('neighbor sets with up to %d members; at most %d are supported' % (width, _lib.NN_MAX_NEIGHBORS))
        terms = _flatten(self.kernel, self.nugget)
        desc, count = _lib.make_terms(terms)
        nkernel = len(self.kernel.terms())
        out = np.empty((2, len(tx)))
        ptr = lambda a: a.ctypes.data_as(ctypes.c_void_p)
        _lib.check(ctx.lib.mgpi_nn_condition(ctx.handle, ptr(tx), len(tx), ptr(sx), ptr(sf), len(sx), sx.shape[1],
                                             ptr(sets.index), ptr(sets.count), width, desc, count, nkernel,
                                             ptr(out[0]), ptr(out[1])))
        if safe:    # the reference's sanity checks (:766-772), on the whole batch
            mean, diag = out
            assert not np.any(mean != mean), 'mean is nan\nkernel=%s\nfirst target=%s' % (self.kernel, tx[np.argmax(mean != mean)])
            assert not np.any(diag != diag), 'diag is nan\nkernel=%s\nfirst target=%s' % (self.kernel, tx[np.argmax(diag != diag)])
            assert np.all(diag > 0), 'marginal variance is negative!\ndiag=%s\nkernel=%s\nx=%s' % \
                (diag[np.argmin(diag)], self.kernel, tx[np.argmin(diag)])
        return out

    def _sample2diag(self, x, ref_x, ref_f, verbose=False, safe=True):
        """(mean, variance) at the single point x (1, D) given the neighbours (ref_x, ref_f)"""
        ref_x = np.asarray(ref_x, dtype=float).reshape(-1, np.shape(x)[-1])
        if len(ref_x) > _lib.NN_MAX_NEIGHBORS:       # larger than any neighbour set: the dense path
            m, c = Interpolator.condition(self, x, ref_x, ref_f, verbose=verbose)
            mean, diag = m[0], c[0, 0]
        elif len(ref_x) == 0:
            mean, diag = self._2diag(x, np.zeros((1, np.shape(x)[-1])), np.zeros(1), [[]], safe=False)[:, 0]
        else:
            mean, diag = self._2diag(x, ref_x, ref_f, [np.arange(len(ref_x))], safe=False)[:, 0]
        if safe:
            assert mean == mean, 'mean is nan\nkernel=%s\nx=%s\nref_x=%s\nref_f=%s' % (self.kernel, x, ref_x, ref_f)
            assert diag == diag, 'diag is nan\nkernel=%s\nx=%s\nref_x=%s\nref_f=%s' % (self.kernel, x, ref_x, ref_f)
            assert diag > 0, 'marginal variance is negative!\ndiag=%s\nkernel=%s\nx=%s\nref_x=%s\nref_f=%s' % \
                (diag, self.kernel, x, ref_x, ref_f)
        return mean, diag

    #--- likelihood and hyper-parameter search (interpolators.py:796-846)

    def _construct_logprob(self, source_x, source_f, logprior=None, fixed=None, temperature=DEFAULT_TEMPERATURE,
                           verbose=False, **kwargs):
        """sort and find the neighbour sets once (they do not depend on the kernel parameters)"""
        source_x, source_f = self._2sorted(np.asarray(source_x), source_f=np.asarray(source_f))
        kwargs['neighbors'] = self._2neighbors(source_x, verbose=verbose)
        return Interpolator._construct_logprob(self, source_x, source_f, logprior=logprior, fixed=fixed,
                                               temperature=temperature, verbose=verbose, **kwargs)

    def loglikelihood(self, source_x, source_f, neighbors=None, verbose=False):
        """sum of the independent 1-D Gaussians of the NNGP factorisation"""
        source_x, source_f = np.asarray(source_x), np.asarray(source_f)
        if neighbors is None:      # otherwise the data are taken to be sorted already, with these neighbour sets
            source_x, source_f = self._2sorted(source_x, source_f=source_f)
            neighbors = self._2neighbors(source_x, verbose=verbose)
        mean, diag = self._2diag(source_x, source_x, source_f, neighbors, verbose=verbose)
        return -0.5 * np.sum((mean - source_f)**2 / diag) - 0.5 * np.sum(np.log(diag)) - 0.5 * len(source_x) * np.log(2 * np.pi)

    def resident(self, source_x, source_f):
        return None              # the NNGP kernels take host arrays (neighbour sets are host-side index lists, :744, :833)

    def loglikelihoods(self, source_x, source_f, kernel_params, resident=None, neighbors=None, verbose=False):
        """(values (B,), flags (B,)) for every row of kernel_params - the vectorised form LogPosterior calls; the NNGP
        likelihood is one search-free launch per row (the neighbour sets do not depend on the kernel parameters), so the
        rows simply run one after the other.  A row that produces a NaN / non-positive conditional variance (the
        reference's `safe` asserts, :766-772) gets -inf and flag 1."""
        rows = np.atleast_2d(np.asarray(kernel_params, dtype=float))
        keep = np.array(self.kernel.params, dtype=float)
        out, flags = np.full(len(rows), -np.inf), np.zeros(len(rows), dtype=np.int32)
        try:
            for i, row in enumerate(rows):
                self.kernel.update(*row)
                try:
                    out[i] = self.loglikelihood(source_x, source_f, neighbors=neighbors, verbose=verbose)
                except AssertionError:
                    flags[i] = 1
        finally:
            self.kernel.update(*keep)
        return out, flags

    #--- prediction (interpolators.py:850-884)

    def condition_diag(self, target_x, source_x, source_f, verbose=False, Verbose=False):
        """(mean (M,), variance (M,)): ``condition`` without the M x M diagonal matrix (additive)"""
        verbose |= Verbose
        if verbose:
            print('finding neighbors for %d target_x within %d source_x samples' % (len(target_x), len(source_x)))
            t0 = time.time()
        source_x, source_f = self._2sorted(np.asarray(source_x), source_f=np.asarray(source_f))
        neighbors = self._2neighbors(source_x, target_x=target_x, verbose=Verbose)
        if verbose:
            print('    time : %.6f sec' % (time.time() - t0))
            print('computing predicted means, variances independently')
            t0 = time.time()
        mean, diag = self._2diag(target_x, source_x, source_f, neighbors, verbose=Verbose)
        if verbose:
            print('    time : %.6f sec' % (time.time() - t0))
        return mean, diag

    def condition(self, target_x, source_x, source_f, verbose=False, Verbose=False):
        """(mean (M,), covariance (M, M)); given the training set the targets are independent, so the covariance is
        diagonal - returned as a dense matrix like the reference's (:884)"""
        mean, diag = self.condition_diag(target_x, source_x, source_f, verbose=verbose, Verbose=Verbose)
        return mean, np.diag(diag)

    def rvs(self, target_x, source_x, source_f, size=1, rng=None):
        """`size` realisations of f(target_x) | source_f.  The reference's rvs (interpolators.py:306-324) draws from
        whatever ``self.condition`` returns; for the NNGP that covariance is diagonal, so the draw is mean + z * sqrt(var)
        with independent standard normals z (numpy's, like Interpolator.rvs; `rng` may be a numpy Generator)."""
        mean, diag = self.condition_diag(target_x, source_x, source_f)
        normal = rng.standard_normal if rng is not None else np.random.standard_normal
        return mean + normal((int(size), len(mean))) * np.sqrt(diag)

    sample = rvs

    # compress / predict: upstream's gap is KEPT, on purpose.
    #
    # The reference leaves both as NotImplementedError with a sketch (mgpi/interpolators.py:875-938): compress would be
    # inv(K_nngp) @ f and predict K_nngp(x*, x) @ compressed.  In the NNGP that pair has no content of its own.  With the
    # samples in rank order the NNGP precision is K_nngp^-1 = (I - A)^T D^-1 (I - A) (A = the per-sample neighbour weights
    # b_i, D = the conditional variances - exactly what _2diag evaluates sample by sample), and a target outside the
    # reference set is conditionally independent of everything but its neighbour set N(x*), so its covariance row is
    # K_nngp(x*, x) = b*^T K_nngp[N(x*), :].  Then
    #     predict(compress(f)) = b*^T K_nngp[N(x*), :] K_nngp^-1 f = b*^T f[N(x*)]
    # which is condition()'s mean, term by term: the "compressed" vector would have to be expanded back to f (two sparse
    # triangular solves) before a single prediction could use it, and the N x N object upstream's sketch inverts is never
    # sparse on the covariance side.  condition / condition_diag are the prediction entry points of this class; a saved
    # model is (sorted source_x, source_f, kernel parameters).  The dense Interpolator.compress / predict remain available
    # through the base class for callers that want the exact process on a subset.
    def compress(self, source_x, source_f, verbose=False, Verbose=False):
        raise NotImplementedError('NearestNeighborInterpolator.compress is left unimplemented, as in the reference '
                                  '(mgpi/interpolators.py:888-899): in the NNGP predict(compress(f)) reduces to '
                                  "condition()'s mean (see the comment above this method); use condition / condition_diag")

    def predict(self, target_x, source_x, compressed, verbose=False, Verbose=False):
        raise NotImplementedError('NearestNeighborInterpolator.predict is left unimplemented, as in the reference '
                                  '(mgpi/interpolators.py:921-926): K_nngp(x*, x) K_nngp^-1 f = b*^T f[N(x*)] is '
                                  "condition()'s mean; use condition / condition_diag")
