#!/usr/bin/env python3
"""developer probe (round 3): a small factorisation of the library beside a persistent gemm-like load that leaves 0 / 32 / 64 CUs
empty (tools/reserved_cu_probe.hip).  Answers: is the panel chain at its stand-alone speed when it has CUs of its own?"""
import ctypes
import os
import sys
import time
import numpy as np
HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import mgpi_amd as mgpi
from mgpi_amd import _lib

probe = ctypes.CDLL(os.path.join(HERE, 'libreserved_probe.so'))
ctx = mgpi.default_context()
rng = np.random.default_rng(0)


def make(n):
    x = rng.random((n, 3))
    k = mgpi.SquaredExponentialKernel(1.0, 0.2, 0.2, 0.2) + mgpi.WhiteNoiseKernel(1e-2)
    desc, count = _lib.make_terms(k.terms())
    X = _lib.DeviceMatrix.from_host(ctx, x)
    K = _lib.DeviceMatrix(ctx, n, n, wide=True)
    return X, K, desc, count


def potrf_ms(X, K, desc, count):
    _lib.check(ctx.lib.mgpi_cov_matrix_dev(ctx.handle, X.handle, None, desc, count, 1, K.handle))
    ctx.sync()
    ctx.tic()
    _lib.check(ctx.lib.mgpi_potrf_dev(ctx.handle, K.handle, None))
    return ctx.toc()


for n in (1024, 2048, 4096):
    X, K, desc, count = make(n)
    alone = min(potrf_ms(X, K, desc, count) for _ in range(5))
    line = 'potrf N=%d alone %.3f ms' % (n, alone)
    for reserve in (0, 1, 2, 4):
        rc = probe.probe_load_start(3000000, reserve, 512)
        assert rc == 0, rc
        potrf_ms(X, K, desc, count)               # also makes sure the load has been submitted and is running
        got = [potrf_ms(X, K, desc, count) for _ in range(6)]
        still = probe.probe_load_running()
        ex, st = ctypes.c_uint(0), ctypes.c_uint(0)
        assert probe.probe_load_wait(ctypes.byref(ex), ctypes.byref(st)) == 0
        line += ' | %d CUs per SE reserved (%d workgroups left, %d stayed): %s ms%s' % (
            reserve, ex.value, st.value, ' '.join('%.3f' % v for v in got[1:]) or '-', '' if still else ' [load ended early]')
    print(line, flush=True)

# what a dependent launch costs, and what clock a dependent fp64 chain sees, alone and beside the load (2 CUs per SE left empty)
probe.probe_tiny_chain.restype = ctypes.c_double
probe.probe_load_start(1, 0, 1)          # creates the probe's buffers
ex, st = ctypes.c_uint(0), ctypes.c_uint(0)
probe.probe_load_wait(ctypes.byref(ex), ctypes.byref(st))
for label, reserve in (('alone', None), ('beside the load, 2 CUs per SE empty', 2), ('beside the load, 4 CUs per SE empty', 4)):
    if reserve is not None:
        assert probe.probe_load_start(3000000, reserve, 512) == 0
        probe.probe_tiny_chain(50, 0)
    tiny = min(probe.probe_tiny_chain(400, 0) for _ in range(3)) / 400 * 1e3
    fma = min(probe.probe_tiny_chain(50, 20000) for _ in range(3)) / 50 * 1e3
    if reserve is not None:
        running = probe.probe_load_running()
        probe.probe_load_wait(ctypes.byref(ex), ctypes.byref(st))
    else:
        running = 1
    print('%s: %.2f us per trivial dependent launch; %.2f us per launch of 20000 dependent fp64 fmas (%.1f ns per fma)%s'
          % (label, tiny, fma, fma / 20000 * 1e3, '' if running else ' [load ended early]'), flush=True)
