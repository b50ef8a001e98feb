#!/usr/bin/env python3
"""developer probe (diagnostic build, -DPANEL_TRACE -> tools/libmgpi_trace.so): per-workgroup phase stamps of the 64-column step
kernels of a small factorisation, alone and beside a big trailing-update gemm on another context.  argv: n_small n_big k_big"""
import ctypes, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import mgpi_amd as mgpi
from mgpi_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, 'tools', 'libmgpi_trace.so')

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
nbig = int(sys.argv[2]) if len(sys.argv) > 2 else 16384
kbig = int(sys.argv[3]) if len(sys.argv) > 3 else 1024
a, b = mgpi.Context(device=0), mgpi.Context(device=0)
lib = a.lib
rng = np.random.default_rng(0)
A = _lib.DeviceMatrix.from_host(a, rng.standard_normal((nbig, kbig)), wide=True)
C = _lib.DeviceMatrix(a, nbig, nbig, wide=True)
x = rng.random((ns, 3))
k = mgpi.SquaredExponentialKernel(1.0, 0.3, 0.3, 0.3) + mgpi.WhiteNoiseKernel(0.1)
desc = _lib.make_terms(k.terms())
X = _lib.DeviceMatrix.from_host(b, x)
L = _lib.DeviceMatrix(b, ns, ns, wide=True)
buf = (ctypes.c_ulonglong * (1 << 18))()
cnt = ctypes.c_uint(0)
lib.mgpi_debug_read_trace.argtypes = [ctypes.c_void_p, ctypes.c_void_p]


def small():
    _lib.check(b.lib.mgpi_cov_matrix_dev(b.handle, X.handle, None, desc[0], desc[1], 1, L.handle))
    b.sync()
    lib.mgpi_debug_read_trace(buf, ctypes.byref(cnt))          # clear
    t0 = time.perf_counter()
    _lib.check(b.lib.mgpi_potrf_dev(b.handle, L.handle, None))
    ms = (time.perf_counter() - t0) * 1e3
    lib.mgpi_debug_read_trace(buf, ctypes.byref(cnt))
    t = np.frombuffer(buf, dtype=np.uint64)[:8 * min(cnt.value, (1 << 18) // 8)].reshape(-1, 8).astype(np.int64)
    return ms, t


def stamps():
    out = (ctypes.c_longlong * 8)()
    lib.mgpi_debug_read_stamps(out)
    return list(out)[:6]


def report(tag, ms, t):
    print('%s: potrf %.3f ms, %d workgroup records' % (tag, ms, len(t)))
    t0 = t[:, 2].min()
    prev_end = None
    for c0 in sorted(set(t[:, 0])):
        g = t[t[:, 0] == c0]
        role = g[:, 1] >> 32
        start, end = g[:, 2].min(), g[:, 7].max()
        d = g[role == 0]
        r = g[role == 1]
        u = g[role == 2]
        line = 'c0=%5d  start %7.1f us  span %6.1f  gap-from-prev %5.1f |' % (c0, (start - t0) / 100.0, (end - start) / 100.0,
                                                                            (start - prev_end) / 100.0 if prev_end else 0.0)
        if len(d):
            line += ' diag wg: wait %5.1f load+upd %5.1f factor %5.1f |' % ((d[0, 2] - start) / 100.0, (d[0, 3] - d[0, 2]) / 100.0, (d[0, 4] - d[0, 3]) / 100.0)
        if len(r):
            line += ' %3d row wgs: start spread %5.1f, load+upd %5.1f factor %5.1f trsm %5.1f (means), last end %6.1f |' % (
                len(r), (r[:, 2].max() - start) / 100.0, np.mean(r[:, 3] - r[:, 2]) / 100.0, np.mean(r[:, 4] - r[:, 3]) / 100.0,
                np.mean(r[:, 5] - r[:, 4]) / 100.0, (r[:, 7].max() - start) / 100.0)
        if len(u):
            line += ' %4d update wgs: start spread %5.1f, mean life %4.1f, last end %6.1f' % (len(u), (u[:, 2].max() - start) / 100.0,
                                                                                             np.mean(u[:, 7] - u[:, 2]) / 100.0, (u[:, 7].max() - start) / 100.0)
        print('  ' + line)
        prev_end = end


for _ in range(3):
    small()
report('alone', *small())
print('  round 32 of one factorisation (shader cycles): barrier+raw write %d, pivot block+LDLT %d, rest %d = [own-row solve %d, operand solves+MFMA %d, rsqrt+stores %d]' % tuple(stamps()))
for _ in range(4):
    _lib.check(a.lib.mgpi_gemm_nt_dev(a.handle, A.handle, A.handle, C.handle, 1))
time.sleep(0.002)
ms, t = small()
a.sync()
report('beside a lower-syrk n=%d K=%d' % (nbig, kbig), ms, t)
print('  round 32 of one factorisation (shader cycles): barrier+raw write %d, pivot block+LDLT %d, rest %d = [own-row solve %d, operand solves+MFMA %d, rsqrt+stores %d]' % tuple(stamps()))
