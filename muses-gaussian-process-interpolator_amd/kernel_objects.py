"""Host-side kernel objects whose arithmetic runs on the GPU.

API mirror of the reference's kernel classes (mgpi/kernels.py): the same class names
(so ``type = SquaredExponentialKernel`` in an INI still resolves through
``factory(Kernel)``, mgpi/utils.py:20-26 / mgpi/io.py:408), the same constructor argument
order, and the same public surface: ``params`` (float ndarray), ``_params`` (names),
``params_dict``, ``params_array``, ``update(...)``, ``k1 + k2`` and the ROW-PAIRED
``cov(x1, x2)`` of mgpi/kernels.py:89-93.

What is different is everything underneath: a kernel here is a *description*.  ``terms()``
flattens any kernel into the ordered list of summands ``(type, sigma, lengths)`` that becomes the
``mgpi_kernel_desc`` array of include/mgpi_hip.h, and ``cov`` ships that description plus the
coordinates to ``mgpi_cov_paired``.  No covariance arithmetic happens in Python.
"""
import warnings

import numpy as np

from . import _lib

__all__ = ['Kernel', 'NDKernel', 'WhiteNoiseKernel', 'MaternKernel', 'SquaredExponentialKernel', 'CombinedKernel']


def _assignments(names, args, params):
    """Decode the three calling conventions of ``update`` (mgpi/kernels.py:63-85) into a list of
    (name, value): a full positional vector, a single dict, or keyword arguments."""
    if args and params:
        raise ValueError('cannot update with both args and params at the same time!')
    if not args:
        return list(params.items())
    if len(args) == 1 and isinstance(args[0], dict) and len(names) != 1:
        return list(args[0].items())
    if len(args) == len(names):
        if len(args) == 1 and isinstance(args[0], dict):
            return list(args[0].items())
        return list(zip(names, args))
    raise ValueError('could not interpret args=%s' % (args,))


class Kernel(object):
    """base of all kernels: a named float parameter vector + a device description"""

    _params = ()          # parameter names, in constructor order

    def __init__(self, *params):
        assert len(params) == len(self._params), 'must specify all parameters!\n\tparams=%s' % (self._params,)
        self.params = np.array(params, dtype=float)

    # -- description used by the device ---------------------------------------------------------

    def terms(self):
        """[(MGPI_TERM_*, sigma, lengths), ...] in summation order"""
        raise NotImplementedError('%s has no device implementation' % type(self).__name__)

    # -- the reference's covariance operator -----------------------------------------------------

    def cov(self, x1, x2):
        """row-paired covariance: x1, x2 both (Nsamp, Ndim) -> (Nsamp,)   [mgpi/kernels.py:89-93]"""
        a = _lib.as_f64(x1, ndim=2)
        b = _lib.as_f64(x2, ndim=2)
        if a.shape != b.shape:
            raise ValueError('paired covariance needs x1 and x2 of equal shape, got %s and %s' % (a.shape, b.shape))
        desc, count = _lib.make_terms(self.terms())
        ctx = _lib.default_context()
        out = np.empty(a.shape[0], dtype=np.float64)
        _lib.check(ctx.lib.mgpi_cov_paired(ctx.handle, a.ctypes.data, b.ctypes.data, a.shape[0], a.shape[1],
                                           desc, count, out.ctypes.data))
        return out

    def __call__(self, x1, x2=None):
        """additive (BASELINE.json north_star asks for it; the reference has no __call__): the PAIRWISE matrix
        K[i, j] = cov(x1[i], x2[j]), shape (n1, n2) - what Interpolator._x2cov builds; x2=None means x2 = x1"""
        a = _lib.as_f64(x1, ndim=2)
        b = a if x2 is None else _lib.as_f64(x2, ndim=2)
        assert b.shape[1] == a.shape[1], 'conflicting number of dimensions for x1 and x2'
        desc, count = _lib.make_terms(self.terms())
        ctx = _lib.default_context()
        out = np.empty((len(a), len(b)), dtype=np.float64)
        _lib.check(ctx.lib.mgpi_cov_matrix(ctx.handle, a.ctypes.data, len(a), None if x2 is None else b.ctypes.data,
                                           len(b), a.shape[1], desc, count, out.ctypes.data))
        return out

    # -- parameter views -------------------------------------------------------------------------

    @property
    def params_dict(self):
        return {name: value for name, value in zip(self._params, self.params)}

    @property
    def params_array(self):
        record = np.empty(1, dtype=[(name, float) for name in self._params])
        for name, value in self.params_dict.items():
            record[name] = value
        return record

    def update(self, *args, **params):
        """update(v0, v1, ...) | update({'name': v}) | update(name=v)"""
        for name, value in _assignments(self._params, args, params):
            try:
                slot = self._params.index(name)
            except ValueError:
                warnings.warn('Warning! cannot update %s in object type %s' % (name, type(self).__name__))
                continue
            self.params[slot] = value

    # -- composition -----------------------------------------------------------------------------

    def _members(self):
        return (self,)

    def __add__(self, other):
        """k1 + k2 is always a FLAT CombinedKernel (mgpi/kernels.py:54-59)"""
        return CombinedKernel(*(self._members() + other._members()))

    def __repr__(self):
        body = ', '.join('%s=%.6e' % pair for pair in self.params_dict.items())
        return '%s(%s)' % (type(self).__name__, body)

    __str__ = __repr__


class NDKernel(Kernel):
    """kernels with one length scale per input dimension: parameter names length0..length{D-1}
    (mgpi/kernels.py:97-118)"""

    _leading = ()         # names of the parameters that precede the lengths

    def __init__(self, *values):
        lead = len(self._leading)
        lengths = values[lead:]
        assert len(lengths), 'must specify at least one length'
        self._num_dim = len(lengths)
        self._params = tuple(self._leading) + tuple('length%d' % k for k in range(self._num_dim))
        Kernel.__init__(self, *values)

    @property
    def num_dim(self):
        return self._num_dim


class WhiteNoiseKernel(Kernel):
    """sigma**2 * delta(x1, x2) with EXACT floating-point coordinate equality (mgpi/kernels.py:122-131)"""

    _params = ('sigma',)

    def terms(self):
        return [(_lib.TERM_WHITE, float(self.params[0]), ())]


class MaternKernel(NDKernel):
    """Matern(order, sigma, *lengths) (mgpi/kernels.py:135-162):
        r = sqrt(2 order sum_d ((x1_d - x2_d)/length_d)**2);  cov = sigma**2 2**(1-order)/Gamma(order) r**order K_order(r)
    evaluated on the GPU for any real order in (0, 120] (K_order by Temme's series / Steed's continued fraction,
    csrc/cov_device.h, in the role of scipy.special.kv; half-integer orders up to 10.5 by their closed form).  Deliberate deviation: at zero separation the reference
    computes 0**order * K(0) = 0 * inf = NaN, which makes every source-source matrix unusable;
    here r == 0 gives the limit sigma**2."""

    _leading = ('order', 'sigma')

    def __init__(self, order, sigma, *lengths):
        NDKernel.__init__(self, order, sigma, *lengths)

    def terms(self):
        return [(_lib.TERM_MATERN, float(self.params[1]), tuple(float(v) for v in self.params[2:]), float(self.params[0]))]


class SquaredExponentialKernel(MaternKernel):
    """sigma**2 * exp(-sum_d ((x1_d - x2_d)/length_d)**2): no factor 1/2 (mgpi/kernels.py:166-185).
    Kept a subclass of MaternKernel, as in the reference, for factory() discovery."""

    _leading = ('sigma',)

    def __init__(self, sigma, *lengths):
        NDKernel.__init__(self, sigma, *lengths)

    def terms(self):                     # overrides MaternKernel.terms: params are (sigma, lengths...), no order
        return [(_lib.TERM_SE, float(self.params[0]), tuple(float(v) for v in self.params[1:]))]


class CombinedKernel(Kernel):
    """a sum of kernels (mgpi/kernels.py:191-293).  Parameter names carry the member index as a
    suffix (``sigma_0``, ``length1_2``); ``params`` is the concatenation of the members' vectors."""

    def __init__(self, *kernels):
        assert len(kernels) >= 2, 'must supply at least 2 kernels'
        dims = set(k.num_dim for k in kernels if isinstance(k, NDKernel))
        assert len(dims) <= 1, 'conflict in dimensionality of kernels!'
        self._num_dim = dims.pop() if dims else None
        self._num_kernels = len(kernels)
        self.kernels = tuple(kernels)
        self._params = tuple(self._combinedkernel_name(name, idx)
                             for idx, member in enumerate(self.kernels) for name in member._params)

    @property
    def params(self):
        return np.concatenate([member.params for member in self.kernels])

    @property
    def num_dim(self):
        return self._num_dim

    @staticmethod
    def _combinedkernel_name(name, index):
        return '%s_%s' % (name, index)

    @staticmethod
    def _kernel_name(name):
        stem, _, tail = name.rpartition('_')
        if not stem or not tail.isdigit():
            raise RuntimeError('cannot map "%s" to parameter name and kernel index!' % name)
        return stem, int(tail)

    def update(self, *args, **params):
        for qualified, value in _assignments(self._params, args, params):
            name, idx = self._kernel_name(qualified)
            self.kernels[idx].update(**{name: value})

    def terms(self):
        return [term for member in self.kernels for term in member.terms()]

    def _members(self):
        return self.kernels

    def __repr__(self):
        lines = [type(self).__name__]
        lines.extend('    kernel %-2d : %s' % pair for pair in enumerate(self.kernels))
        return '\n'.join(lines)

    __str__ = __repr__
