"""ctypes binding of libmgpi_hip.so (include/mgpi_hip.h) - the only way the Python host reaches the GPU.

There is no CPU fallback: if the shared library is missing, or no HIP device is visible, every
numeric entry point raises.  The library is built in-tree by ``__graft_entry__.build()`` /
``make -C muses-gaussian-process-interpolator_amd/csrc``.
"""
import ctypes
import os
import sys
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, 'libmgpi_hip.so')

MAX_D = 16
MAX_TERMS = 8
NN_MAX_NEIGHBORS = 62
TERM_SE = 0
TERM_WHITE = 1
TERM_MATERN = 2
COMM_ID_BYTES = 128        # MGPI_COMM_ID_BYTES

c_double_p = ctypes.POINTER(ctypes.c_double)
c_int64 = ctypes.c_int64


class KernelDesc(ctypes.Structure):
    """struct mgpi_kernel_desc"""
    _fields_ = [('type', ctypes.c_int32),
                ('reserved', ctypes.c_int32),
                ('sigma', ctypes.c_double),
                ('order', ctypes.c_double),
                ('length', ctypes.c_double * MAX_D)]


class MgpiError(RuntimeError):
    """a negative status from libmgpi_hip.so"""


class NotPositiveDefinite(np.linalg.LinAlgError):
    """potrf met a non-positive pivot (status > 0)"""

    def __init__(self, pivot, message):
        np.linalg.LinAlgError.__init__(self, message)
        self.pivot = pivot


# every exported symbol of include/mgpi_hip.h: name -> (restype, argtypes)
_vp = ctypes.c_void_p
_SIGNATURES = {
    'mgpi_last_error': (ctypes.c_char_p, []),
    'mgpi_device_count': (ctypes.c_int, [ctypes.POINTER(ctypes.c_int)]),
    'mgpi_ctx_create': (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.POINTER(_vp)]),
    'mgpi_ctx_destroy': (ctypes.c_int, [_vp]),
    'mgpi_ctx_sync': (ctypes.c_int, [_vp]),
    'mgpi_ctx_trim': (ctypes.c_int, [_vp]),
    'mgpi_ctx_stream': (_vp, [_vp]),
    'mgpi_ctx_tic': (ctypes.c_int, [_vp]),
    'mgpi_ctx_toc': (ctypes.c_int, [_vp, c_double_p]),
    'mgpi_mat_alloc': (ctypes.c_int, [_vp, c_int64, c_int64, ctypes.POINTER(_vp)]),
    'mgpi_mat_alloc_wide': (ctypes.c_int, [_vp, c_int64, c_int64, ctypes.POINTER(_vp)]),
    'mgpi_mat_free': (ctypes.c_int, [_vp]),
    'mgpi_mat_upload': (ctypes.c_int, [_vp, _vp, c_int64]),
    'mgpi_mat_download': (ctypes.c_int, [_vp, _vp, c_int64]),
    'mgpi_mat_shape': (ctypes.c_int, [_vp, ctypes.POINTER(c_int64), ctypes.POINTER(c_int64), ctypes.POINTER(c_int64)]),
    'mgpi_mat_devptr': (_vp, [_vp]),
    'mgpi_mat_mark_factored': (ctypes.c_int, [_vp, ctypes.c_int]),
    'mgpi_mat_copy': (ctypes.c_int, [_vp, _vp, _vp]),
    'mgpi_mat_symmetrize': (ctypes.c_int, [_vp, _vp]),
    'mgpi_mat_tril': (ctypes.c_int, [_vp, _vp]),
    'mgpi_mat_diag': (ctypes.c_int, [_vp, _vp, _vp]),
    'mgpi_mat_unscale': (ctypes.c_int, [_vp, _vp, _vp]),
    'mgpi_mat_add_diag': (ctypes.c_int, [_vp, _vp, ctypes.c_double]),
    'mgpi_ctx_set_pivot_tol': (ctypes.c_int, [_vp, ctypes.c_double]),
    'mgpi_mat_view': (ctypes.c_int, [_vp, c_int64, c_int64, c_int64, c_int64, ctypes.POINTER(_vp)]),
    'mgpi_cov_paired': (ctypes.c_int, [_vp, _vp, _vp, c_int64, ctypes.c_int, ctypes.POINTER(KernelDesc), ctypes.c_int, _vp]),
    'mgpi_nn_neighbors': (ctypes.c_int, [_vp, _vp, _vp, c_int64, ctypes.c_int, _vp, _vp, c_int64, ctypes.c_int, ctypes.c_int, _vp, _vp]),
    'mgpi_nn_condition': (ctypes.c_int, [_vp, _vp, c_int64, _vp, _vp, c_int64, ctypes.c_int, _vp, _vp, ctypes.c_int,
                                         ctypes.POINTER(KernelDesc), ctypes.c_int, ctypes.c_int, _vp, _vp]),
    'mgpi_cov_matrix': (ctypes.c_int, [_vp, _vp, c_int64, _vp, c_int64, ctypes.c_int, ctypes.POINTER(KernelDesc), ctypes.c_int, _vp]),
    'mgpi_cov_matrix_dev': (ctypes.c_int, [_vp, _vp, _vp, ctypes.POINTER(KernelDesc), ctypes.c_int, ctypes.c_int, _vp]),
    'mgpi_potrf_dev': (ctypes.c_int, [_vp, _vp, c_double_p]),
    'mgpi_potrs_dev': (ctypes.c_int, [_vp, _vp, _vp]),
    'mgpi_posv_dev': (ctypes.c_int, [_vp, _vp, _vp, c_double_p]),
    'mgpi_cov_riders_dev': (ctypes.c_int, [_vp, _vp, _vp, ctypes.POINTER(KernelDesc), ctypes.c_int, ctypes.POINTER(KernelDesc), ctypes.c_int, _vp]),
    'mgpi_posv_riders_dev': (ctypes.c_int, [_vp, _vp, _vp, c_double_p]),
    'mgpi_riders_var_dev': (ctypes.c_int, [_vp, _vp, _vp, ctypes.POINTER(KernelDesc), ctypes.c_int, _vp]),
    'mgpi_mat_square_view': (ctypes.c_int, [_vp, ctypes.POINTER(_vp)]),
    'mgpi_potrs': (ctypes.c_int, [_vp, _vp, _vp, c_int64, ctypes.c_int, _vp]),
    'mgpi_trsm_dev': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int]),
    'mgpi_loglikelihood_batch': (ctypes.c_int, [_vp, _vp, _vp, ctypes.POINTER(KernelDesc), ctypes.c_int, ctypes.c_int, _vp, _vp]),
    'mgpi_predict_mean_dev': (ctypes.c_int, [_vp, _vp, _vp, ctypes.POINTER(KernelDesc), ctypes.c_int, _vp, _vp]),
    'mgpi_predict_var_dev': (ctypes.c_int, [_vp, _vp, _vp, ctypes.POINTER(KernelDesc), ctypes.c_int, _vp, c_int64, _vp]),
    'mgpi_condition_cov_dev': (ctypes.c_int, [_vp, _vp, _vp, ctypes.POINTER(KernelDesc), ctypes.c_int, _vp, _vp]),
    'mgpi_gemm_nt_dev': (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int]),
    'mgpi_mat_wrap': (ctypes.c_int, [_vp, _vp, c_int64, c_int64, c_int64, ctypes.c_int, ctypes.POINTER(_vp)]),
    'mgpi_cov_block_dev': (ctypes.c_int, [_vp, _vp, c_int64, _vp, c_int64, ctypes.POINTER(KernelDesc), ctypes.c_int, _vp]),
    'mgpi_gemm_nn_dev': (ctypes.c_int, [_vp, _vp, _vp, _vp]),
    'mgpi_gemm_nn_neg_dev': (ctypes.c_int, [_vp, _vp, _vp, _vp]),
    'mgpi_gemm_nt_stair_dev': (ctypes.c_int, [_vp, _vp, _vp, _vp] + [ctypes.c_int] * 7),
    'mgpi_trtri_neg_dev': (ctypes.c_int, [_vp, _vp, c_int64, c_int64, c_int64, c_int64, c_int64, ctypes.c_int, _vp]),
    'mgpi_trsm_right_dev': (ctypes.c_int, [_vp, _vp, _vp]),
    'mgpi_trsv_dev': (ctypes.c_int, [_vp, _vp, _vp, ctypes.c_int]),
    'mgpi_gemv_dev': (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int]),
    'mgpi_ctx_prefer_high_priority': (ctypes.c_int, [_vp]),
    'mgpi_gemm_nt_stair_merged_dev': (ctypes.c_int, [_vp, _vp, _vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                     ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.POINTER(c_int64)]),
    'mgpi_ctx_wait_count': (ctypes.c_int, [_vp, _vp, c_int64]),
    'mgpi_ctx_count_reset': (ctypes.c_int, [_vp]),
    'mgpi_ctx_streams_overlap': (ctypes.c_int, [_vp, _vp, ctypes.POINTER(ctypes.c_int)]),
    'mgpi_potrf_enqueue_dev': (ctypes.c_int, [_vp, _vp, c_int64, _vp]),
    'mgpi_potrf_panel_enqueue_dev': (ctypes.c_int, [_vp, _vp, c_int64, _vp]),
    'mgpi_ctx_status_reset': (ctypes.c_int, [_vp]),
    'mgpi_ctx_status_read': (ctypes.c_int, [_vp, ctypes.POINTER(c_int64), c_double_p]),
    'mgpi_ctx_wait': (ctypes.c_int, [_vp, _vp]),
    'mgpi_mat_fill_diag': (ctypes.c_int, [_vp, _vp, ctypes.c_double]),
    'mgpi_axpy_dev': (ctypes.c_int, [_vp, ctypes.c_double, _vp, _vp]),
    'mgpi_copy_blocks_dev': (ctypes.c_int, [_vp, _vp, c_int64, c_int64, _vp, c_int64, c_int64, c_int64, c_int64]),
    'mgpi_colsq_dev': (ctypes.c_int, [_vp, _vp, _vp]),
    'mgpi_comm_unique_id': (ctypes.c_int, [_vp]),
    'mgpi_comm_create': (ctypes.c_int, [_vp, _vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_vp)]),
    'mgpi_comm_split': (ctypes.c_int, [_vp, ctypes.c_int, ctypes.c_int, ctypes.POINTER(_vp)]),
    'mgpi_comm_destroy': (ctypes.c_int, [_vp]),
    'mgpi_comm_info': (ctypes.c_int, [_vp, ctypes.POINTER(ctypes.c_int), ctypes.POINTER(ctypes.c_int)]),
    'mgpi_comm_bcast': (ctypes.c_int, [_vp, _vp, _vp, c_int64, ctypes.c_int]),
    'mgpi_comm_reduce_sum': (ctypes.c_int, [_vp, _vp, _vp, c_int64, ctypes.c_int]),
    'mgpi_comm_allreduce': (ctypes.c_int, [_vp, _vp, _vp, c_int64, ctypes.c_int]),
    'mgpi_comm_allgather': (ctypes.c_int, [_vp, _vp, _vp, _vp, c_int64]),
    'mgpi_comm_barrier': (ctypes.c_int, [_vp, _vp]),
    'mgpi_probe_mfma_f64': (ctypes.c_int, [_vp, ctypes.c_int, c_double_p]),
    'mgpi_probe_hbm': (ctypes.c_int, [_vp, c_int64, ctypes.c_int, c_double_p, c_double_p]),
    'mgpi_ctx_set_lookahead': (ctypes.c_int, [_vp, ctypes.c_int]),
    'mgpi_ctx_set_profile': (ctypes.c_int, [_vp, ctypes.c_int]),
    'mgpi_ctx_set_profile_slots': (ctypes.c_int, [_vp, ctypes.c_uint]),
    'mgpi_ctx_get_profile': (ctypes.c_int, [_vp, ctypes.c_int, c_double_p, ctypes.POINTER(c_int64)]),
    'mgpi_ctx_get_profile_flops': (ctypes.c_int, [_vp, ctypes.c_int, c_double_p]),
}

EXPORTED_SYMBOLS = tuple(sorted(_SIGNATURES))

_lib = None
_lib_lock = threading.Lock()


def load_library():
    """dlopen libmgpi_hip.so and declare every prototype; raises if the library was not built"""
    global _lib
    with _lib_lock:
        if _lib is None:
            if not os.path.exists(LIB_PATH):
                raise MgpiError('%s not found: build it with `python -c "import __graft_entry__ as g; g.build()"` '
                                '(there is no CPU fallback)' % LIB_PATH)
            lib = ctypes.CDLL(LIB_PATH)
            for name, (restype, argtypes) in _SIGNATURES.items():
                fn = getattr(lib, name)   # AttributeError if the symbol is missing
                fn.restype = restype
                fn.argtypes = argtypes
            _lib = lib
    return _lib


def last_error():
    msg = load_library().mgpi_last_error()
    return msg.decode('utf-8', 'replace') if msg else ''


def check(status):
    """translate an mgpi status into the exception the reference would have raised"""
    if status == 0:
        return
    msg = last_error()
    if status > 0:
        raise NotPositiveDefinite(status, msg or 'covariance is not positive definite!')
    if 'conflicting number of dimensions' in msg:
        raise AssertionError(msg)          # mgpi/interpolators.py:219
    raise MgpiError('libmgpi_hip status %d: %s' % (status, msg))


def device_count():
    n = ctypes.c_int(0)
    check(load_library().mgpi_device_count(ctypes.byref(n)))
    return n.value


def make_terms(term_list):
    """[(type, sigma, lengths[, order]), ...] -> (KernelDesc array, count)"""
    n = len(term_list)
    if not (1 <= n <= MAX_TERMS):
        raise ValueError('a kernel must flatten to between 1 and %d summands (got %d)' % (MAX_TERMS, n))
    arr = (KernelDesc * n)()
    for i, term in enumerate(term_list):
        kind, sigma, lengths = term[0], term[1], term[2]
        arr[i].type = int(kind)
        arr[i].sigma = float(sigma)
        arr[i].order = float(term[3]) if len(term) > 3 else 0.0
        if len(lengths) > MAX_D:
            raise ValueError('at most %d dimensions are supported on the device path' % MAX_D)
        for k, val in enumerate(lengths):
            arr[i].length[k] = float(val)
    return arr, n


def make_terms_batch(term_lists):
    """[[term, ...] per problem] (every problem the same number of summands) -> (KernelDesc array, nterms, batch)"""
    nterms = len(term_lists[0])
    if any(len(tl) != nterms for tl in term_lists):
        raise ValueError('every member of a batch must have the same number of kernel summands')
    if not (1 <= nterms <= MAX_TERMS):
        raise ValueError('a kernel must flatten to between 1 and %d summands (got %d)' % (MAX_TERMS, nterms))
    arr = (KernelDesc * (nterms * len(term_lists)))()
    for b, tl in enumerate(term_lists):
        one, _ = make_terms(tl)
        for i in range(nterms):
            arr[b * nterms + i] = one[i]
    return arr, nterms, len(term_lists)


def as_f64(a, ndim=None):
    """C-contiguous float64 view/copy.  parse_table hands out F-ordered (N, D) views
    (mgpi/io.py:123) - those are copied once here."""
    a = np.ascontiguousarray(a, dtype=np.float64)
    if ndim is not None and a.ndim != ndim:
        raise AssertionError('expected a %d-dimensional array, got shape %s' % (ndim, a.shape))
    return a


class DeviceMatrix(object):
    """owning handle of an mgpi_mat"""

    def __init__(self, ctx, rows, cols, handle=None, parent=None, wide=False):
        self.ctx = ctx
        self.rows = int(rows)
        self.cols = int(cols)
        self._parent = parent      # keeps the owner of a view alive
        if handle is None:
            h = _vp()
            alloc = ctx.lib.mgpi_mat_alloc_wide if wide else ctx.lib.mgpi_mat_alloc
            check(alloc(ctx.handle, self.rows, self.cols, ctypes.byref(h)))
            handle = h
        self.handle = handle

    @classmethod
    def from_host(cls, ctx, array, wide=False):
        a = as_f64(array)
        if a.ndim == 1:
            a = a.reshape((-1, 1))
        m = cls(ctx, a.shape[0], a.shape[1], wide=wide)
        check(ctx.lib.mgpi_mat_upload(m.handle, a.ctypes.data, a.shape[1]))
        return m

    def upload(self, array):
        a = as_f64(array).reshape((self.rows, self.cols))
        check(self.ctx.lib.mgpi_mat_upload(self.handle, a.ctypes.data, self.cols))

    def to_host(self):
        out = np.empty((self.rows, self.cols), dtype=np.float64)
        check(self.ctx.lib.mgpi_mat_download(self.handle, out.ctypes.data, self.cols))
        return out

    def view(self, r0, c0, rows, cols):
        h = _vp()
        check(self.ctx.lib.mgpi_mat_view(self.handle, r0, c0, rows, cols, ctypes.byref(h)))
        return DeviceMatrix(self.ctx, rows, cols, handle=h, parent=self)

    @property
    def devptr(self):
        return self.ctx.lib.mgpi_mat_devptr(self.handle)

    def shape3(self):
        """(rows, cols, row stride) as the library sees this handle"""
        r, c, ld = c_int64(0), c_int64(0), c_int64(0)
        check(self.ctx.lib.mgpi_mat_shape(self.handle, ctypes.byref(r), ctypes.byref(c), ctypes.byref(ld)))
        return r.value, c.value, ld.value

    def mark_factored(self, flag=True):
        check(self.ctx.lib.mgpi_mat_mark_factored(self.handle, int(bool(flag))))

    def free(self):
        if self.handle is not None and self.ctx.handle is not None:
            self.ctx.lib.mgpi_mat_free(self.handle)
        self.handle = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class Context(object):
    """one GPU: mgpi_ctx"""

    def __init__(self, device=None, nb=None):
        self.lib = load_library()
        if device is None:
            device = int(os.environ.get('MGPI_DEVICE', os.environ.get('LOCAL_RANK', '0')))
            n = device_count()
            if n > 0:
                device %= n
        if nb is None:
            nb = int(os.environ.get('MGPI_NB', '0'))
        h = _vp()
        check(self.lib.mgpi_ctx_create(int(device), int(nb), ctypes.byref(h)))
        self.handle = h
        self.device = int(device)

    def sync(self):
        check(self.lib.mgpi_ctx_sync(self.handle))

    def trim(self):
        """hand the pooled device memory of freed matrices back to the driver"""
        check(self.lib.mgpi_ctx_trim(self.handle))

    def tic(self):
        check(self.lib.mgpi_ctx_tic(self.handle))

    def toc(self):
        ms = ctypes.c_double(0.0)
        check(self.lib.mgpi_ctx_toc(self.handle, ctypes.byref(ms)))
        return ms.value

    def set_pivot_tol(self, tol):
        """pivot k must exceed tol * K_kk; default 100 eps, 0 = LAPACK's pivot > 0 (include/mgpi_hip.h)"""
        check(self.lib.mgpi_ctx_set_pivot_tol(self.handle, float(tol)))

    def probe_mfma_f64(self, iters=200000):
        """TFLOP/s of back-to-back v_mfma_f64_16x16x4 (random operands, two 4-wave workgroups per CU): what this GPU sustains"""
        out = ctypes.c_double(0.0)
        check(self.lib.mgpi_probe_mfma_f64(self.handle, int(iters), ctypes.byref(out)))
        return float(out.value)

    def set_lookahead(self, on):
        check(self.lib.mgpi_ctx_set_lookahead(self.handle, int(bool(on))))

    def set_profile(self, on, slots=None):
        """HIP-event pairs around the library's launches; slots: an iterable of slot numbers to restrict them to"""
        if on and slots is not None:
            check(self.lib.mgpi_ctx_set_profile_slots(self.handle, sum(1 << int(s) for s in set(slots))))
        else:
            check(self.lib.mgpi_ctx_set_profile(self.handle, int(bool(on))))

    def get_profile(self, which):
        ms = ctypes.c_double(0.0)
        n = c_int64(0)
        check(self.lib.mgpi_ctx_get_profile(self.handle, int(which), ctypes.byref(ms), ctypes.byref(n)))
        return ms.value, n.value

    def get_profile_flops(self, which):
        flops = ctypes.c_double(0.0)
        check(self.lib.mgpi_ctx_get_profile_flops(self.handle, int(which), ctypes.byref(flops)))
        return flops.value

    def close(self):
        if self.handle is not None:
            self.lib.mgpi_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_default_ctx = None


def default_context():
    """process-wide context on MGPI_DEVICE / LOCAL_RANK (default GPU 0), created on first use"""
    global _default_ctx
    if _default_ctx is None:
        _default_ctx = Context()
    return _default_ctx


def set_default_context(ctx):
    global _default_ctx
    _default_ctx = ctx
