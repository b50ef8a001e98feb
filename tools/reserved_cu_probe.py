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
