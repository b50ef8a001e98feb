"""CPU-side checks: the host mirror of the reference API, and that the C-ABI library loads and exports
every symbol include/mgpi_hip.h declares.  No compute calls (there is no GPU here)."""
import os
import re

import numpy as np
import pytest

import mgpi_amd as mgpi
from conftest import ROOT


def header_functions():
    text = open(os.path.join(ROOT, 'include', 'mgpi_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(mgpi_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    lib = mgpi._lib.load_library()
    declared = header_functions()
    assert len(declared) >= 30
    for name in declared:
        assert hasattr(lib, name), 'libmgpi_hip.so does not export %s' % name
    # and the ctypes prototype table covers exactly the header
    assert sorted(mgpi._lib.EXPORTED_SYMBOLS) == declared


def test_kernel_desc_layout_matches_header():
    import ctypes
    d = mgpi._lib.KernelDesc
    assert ctypes.sizeof(d) == 4 + 4 + 8 + 8 + 8 * 16
    assert d.sigma.offset == 8 and d.order.offset == 16 and d.length.offset == 24


def test_no_gpu_fails_loudly():
    """the product has no CPU fallback: without a device the context cannot be created"""
    if mgpi._lib.device_count() > 0:
        pytest.skip('a GPU is present')
    with pytest.raises(mgpi.MgpiError):
        mgpi.Context(device=0)
    k = mgpi.SquaredExponentialKernel(1.0, 0.5)
    with pytest.raises(mgpi.MgpiError):
        mgpi.Interpolator(k).compress(np.zeros((4, 1)), np.zeros(4))


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, 'muses-gaussian-process-interpolator_amd')
    for dirpath, _, files in os.walk(pkg):
        for name in files:
            if name.endswith(('.py', '.hip', '.h')):
                text = open(os.path.join(dirpath, name)).read()
                assert 'mgpi_oracle' not in text and 'from oracle' not in text and 'import oracle' not in text, name


def test_product_is_free_of_pytorch():
    """north_star: "no PyTorch".  Device memory, streams and RCCL all go through the C-ABI; nothing under the package -
    nor bench.py - imports or names torch (the gloo CPU tests inject their own transport from tests/)."""
    pkg = os.path.join(ROOT, 'muses-gaussian-process-interpolator_amd')
    for dirpath, _, files in os.walk(pkg):
        for name in files:
            if name.endswith(('.py', '.hip', '.h')):
                assert 'torch' not in open(os.path.join(dirpath, name)).read().lower(), name
    bench_src = open(os.path.join(ROOT, 'bench.py')).read()
    assert 'import torch' not in bench_src and 'torch.cuda' not in bench_src


# ---- parameter containers: behaviour of mgpi/kernels.py:24-118,191-285 -----------------------------------

def test_parameter_names_and_values():
    se = mgpi.SquaredExponentialKernel(0.1, 10, 20)
    assert se._params == ('sigma', 'length0', 'length1')
    assert se.num_dim == 2
    assert se.params.dtype == float and list(se.params) == [0.1, 10.0, 20.0]
    assert se.params_dict == {'sigma': 0.1, 'length0': 10.0, 'length1': 20.0}
    assert se.params_array.dtype.names == ('sigma', 'length0', 'length1')
    assert se.params_array['length1'][0] == 20.0
    wn = mgpi.WhiteNoiseKernel(1e-4)
    assert wn._params == ('sigma',)
    ma = mgpi.MaternKernel(2, 0.1, 15, 10)
    assert ma._params == ('order', 'sigma', 'length0', 'length1') and ma.num_dim == 2
    assert str(wn) == 'WhiteNoiseKernel(sigma=1.000000e-04)'
    with pytest.raises(AssertionError):
        mgpi.WhiteNoiseKernel()
    with pytest.raises(AssertionError):
        mgpi.SquaredExponentialKernel(1.0)


def test_update_forms():
    se = mgpi.SquaredExponentialKernel(1.0, 2.0, 3.0)
    se.update(4, 5, 6)
    assert list(se.params) == [4, 5, 6]
    se.update(sigma=7)
    se.update({'length1': 9})
    assert list(se.params) == [7, 5, 9]
    with pytest.warns(UserWarning):
        se.update(nonsense=1)
    with pytest.raises(ValueError):
        se.update(1, 2)
    with pytest.raises(ValueError):
        se.update(1, 2, 3, sigma=1)


def test_combined_kernel_flattening_and_names():
    a = mgpi.SquaredExponentialKernel(1.0, 0.3, 0.4)
    b = mgpi.WhiteNoiseKernel(0.01)
    c = mgpi.SquaredExponentialKernel(0.5, 1.0, 2.0)
    ab = a + b
    assert isinstance(ab, mgpi.CombinedKernel) and ab.kernels == (a, b)
    abc = ab + c
    assert abc.kernels == (a, b, c)                       # flat, never nested (kernels.py:54-59)
    assert (c + ab).kernels == (c, a, b)
    assert abc._params == ('sigma_0', 'length0_0', 'length1_0', 'sigma_1', 'sigma_2', 'length0_2', 'length1_2')
    assert list(abc.params) == [1.0, 0.3, 0.4, 0.01, 0.5, 1.0, 2.0]
    abc.update(sigma_1=0.02, length1_2=5.0)
    assert b.params[0] == 0.02 and c.params[2] == 5.0    # members are shared, not copied
    abc.update(*range(7))
    assert list(abc.params) == list(range(7))
    assert abc.terms() == [(0, 0.0, (1.0, 2.0)), (1, 3.0, ()), (0, 4.0, (5.0, 6.0))]
    with pytest.raises(AssertionError):
        mgpi.CombinedKernel(a)
    with pytest.raises(AssertionError):
        mgpi.CombinedKernel(a, mgpi.SquaredExponentialKernel(1.0, 1.0))
    with pytest.raises(RuntimeError):
        abc.update(sigma=1.0)
    assert 'kernel 0  : SquaredExponentialKernel' in str(abc)


def test_factory_finds_classes_by_name():
    """how the reference's INI loader resolves `type = ...` (utils.py:20-26, io.py:408,537)"""
    kernels = mgpi.factory(mgpi.Kernel)
    for name in ['Kernel', 'NDKernel', 'WhiteNoiseKernel', 'MaternKernel', 'SquaredExponentialKernel', 'CombinedKernel']:
        assert kernels[name].__name__ == name
    assert mgpi.factory(mgpi.Interpolator)['Interpolator'] is mgpi.Interpolator
    assert issubclass(mgpi.SquaredExponentialKernel, mgpi.MaternKernel)


def test_interpolator_holds_kernel_and_nugget():
    k = mgpi.SquaredExponentialKernel(1.0, 1.0)
    it = mgpi.Interpolator(k)
    it.update(sigma=3.0)
    assert k.params[0] == 3.0
    with pytest.raises(RuntimeError):
        it.update_nugget(sigma=1.0)
    it = mgpi.Interpolator(k, nugget=mgpi.WhiteNoiseKernel(0.1))
    it.update_nugget(sigma=0.2)
    assert it.nugget.params[0] == 0.2


def test_matern_terms_carry_the_order():
    """MaternKernel(order, sigma, *lengths) -> one device term descriptor; the SE subclass drops the order"""
    (kind, sigma, lengths, order), = mgpi.MaternKernel(2.5, 0.1, 15, 10).terms()
    assert (kind, sigma, lengths, order) == (mgpi._lib.TERM_MATERN, 0.1, (15.0, 10.0), 2.5)
    arr, n = mgpi._lib.make_terms(mgpi.MaternKernel(2.5, 0.1, 15, 10).terms())
    assert n == 1 and arr[0].type == 2 and arr[0].order == 2.5 and arr[0].length[1] == 10.0
    (kind, sigma, lengths), = mgpi.SquaredExponentialKernel(0.1, 15, 10).terms()
    assert kind == mgpi._lib.TERM_SE
    mixed = mgpi.SquaredExponentialKernel(0.1, 10, 20) + mgpi.WhiteNoiseKernel(1e-4) + mgpi.MaternKernel(2, 0.1, 15, 10)
    assert [t[0] for t in mixed.terms()] == [0, 1, 2]      # etc/example-interpolator.ini's three kernels


def test_make_terms_limits():
    with pytest.raises(ValueError):
        mgpi._lib.make_terms([])
    with pytest.raises(ValueError):
        mgpi._lib.make_terms([(0, 1.0, tuple(range(1, 18)))])
    arr, n = mgpi._lib.make_terms([(0, 2.0, (1.0, 3.0)), (1, 0.5, ())])
    assert n == 2 and arr[0].sigma == 2.0 and arr[0].length[1] == 3.0 and arr[1].type == 1


def test_logposterior_guards_without_touching_the_gpu():
    """non-positive / NaN parameters and prior-excluded points return -inf before any device work
    (interpolators.py:371-379); fixed parameters are applied and left out of the free set"""
    from mgpi_amd.hyper import LogPosterior, initial_points
    k = mgpi.SquaredExponentialKernel(1.0, 0.5, 0.7)
    it = mgpi.Interpolator(k)
    lp = LogPosterior(it, np.zeros((3, 2)), np.zeros(3), fixed={'length1': 0.9})
    assert lp.free == ['sigma', 'length0'] and k.params[2] == 0.9
    assert lp([-1.0, 1.0]) == -np.inf and lp([np.nan, 1.0]) == -np.inf
    lp2 = LogPosterior(it, np.zeros((3, 2)), np.zeros(3), logprior=lambda p: -np.inf)
    assert lp2([1.0, 1.0, 1.0]) == -np.inf
    pts = initial_points(it, size=5, rng=np.random.default_rng(0))
    assert pts.shape == (5, 3) and np.all(pts > 0)
    pts = initial_points(it, fixed={'sigma': 1.0}, size=4, rng=np.random.default_rng(0))
    assert pts.shape == (4, 2)
    assert hasattr(it, 'optimize_kernel') and hasattr(it, 'sample_kernel')


def test_utils_mirror_the_reference():
    """construct_logprior / subdivide_data / seed (mgpi/utils.py:11-16,30-73), pure numpy"""
    lp = mgpi.construct_logprior(('sigma', 'length0', 'length1'), {'sigma': (0.1, 2.0), 'length1': (1.0, 3.0)}, fixed={'length0': 0.5})
    assert lp([1.0, 2.0]) == 0.0 and lp([3.0, 2.0]) == -np.inf and lp([1.0, 0.5]) == -np.inf
    with pytest.raises(AssertionError):
        mgpi.construct_logprior(('sigma',), {'nope': (0, 1)}, fixed={})
    mgpi.seed(7)
    x, f = np.arange(20.0).reshape(10, 2), np.arange(10.0)
    (xa, fa), (xb, fb) = mgpi.subdivide_data(x, f, frac=0.3)
    assert len(xa) == 3 and len(xb) == 7 and sorted(np.concatenate([fa, fb])) == list(f)
    assert np.array_equal(xa[:, 0] / 2, fa)
    mgpi.seed(7)
    (xa2, _), _ = mgpi.subdivide_data(x, f, frac=0.3)
    assert np.array_equal(xa, xa2)


def test_sample_kernel_returns_the_reference_layout(monkeypatch):
    """sample_kernel returns sampler.get_chain() / get_log_prob() UNflattened, (num_samples, num_walkers,
    num_dim) and (num_samples, num_walkers), and forwards progress=Verbose (interpolators.py:587-609).  emcee is not
    importable here, so a stub sampler stands in; the log-probability is never evaluated (no GPU)."""
    import sys
    import types
    calls = []

    class StubSampler(object):
        def __init__(self, nwalkers, ndim, fn, vectorize=False):
            self.nwalkers, self.ndim, self.vectorize, self.steps = nwalkers, ndim, vectorize, 0

        def run_mcmc(self, state, nsteps, progress=False):
            calls.append((nsteps, progress))
            self.steps = nsteps
            return state

        def reset(self):
            self.steps = 0

        def get_chain(self, flat=False):
            assert not flat
            return np.zeros((self.steps, self.nwalkers, self.ndim))

        def get_log_prob(self, flat=False):
            assert not flat
            return np.zeros((self.steps, self.nwalkers))

    monkeypatch.setitem(sys.modules, 'emcee', types.SimpleNamespace(EnsembleSampler=StubSampler))
    it = mgpi.Interpolator(mgpi.SquaredExponentialKernel(1.0, 0.5, 0.7))
    np.random.seed(0)
    chain, logp, sampler = it.sample_kernel(np.zeros((4, 2)), np.zeros(4), fixed={'sigma': 1.0}, num_burnin=3, num_samples=5, Verbose=True)
    assert chain.shape == (5, 4, 2) and logp.shape == (5, 4)          # num_walkers defaults to 2 * num_dim = 4
    assert calls == [(3, True), (5, True)] and sampler.vectorize


def test_rccl_load_failure_is_reported_not_fatal(tmp_path):
    """MGPI_RCCL_LIB names the library to use; one that cannot be loaded gives MGPI_ERR_COMM (-5) with dlopen's reason
    from every mgpi_comm_* entry that needs RCCL - no crash, no fallback to another library [r3].  Needs no GPU:
    mgpi_comm_unique_id only loads librccl."""
    import subprocess
    import sys
    code = ('import sys; sys.path.insert(0, %r)\n'
            'import mgpi_amd as mgpi\n'
            'from mgpi_amd.distributed import RcclTransport\n'
            'for attempt in range(2):\n'
            '    try:\n'
            '        RcclTransport.new_unique_id()\n'
            '        raise SystemExit("a unique id came out of a library that does not exist")\n'
            '    except mgpi.MgpiError as err:\n'
            '        assert "status -5" in str(err) and ("dlopen" in str(err) or "could not be loaded earlier" in str(err)), str(err)\n'
            'print("COMM-ERR-OK")\n' % os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    out = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, MGPI_RCCL_LIB=str(tmp_path / 'no_such_librccl.so')),
                         capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and 'COMM-ERR-OK' in out.stdout, out.stdout[-1000:] + out.stderr[-2000:]


def test_riders_rule(monkeypatch):
    """where condition_diag lets the queries ride through the factorisation (gp.py::riders_pay): the chain-bound regime only"""
    from mgpi_amd.gp import riders_pay
    monkeypatch.delenv('MGPI_RIDERS', raising=False)
    monkeypatch.delenv('MGPI_RIDERS_MAX_N', raising=False)
    assert riders_pay(8192, 4096) and riders_pay(16384, 128) and riders_pay(300, 300)
    assert not riders_pay(32768, 4096) and not riders_pay(8192, 127) and not riders_pay(1000, 1001)
    monkeypatch.setenv('MGPI_RIDERS', '0')
    assert not riders_pay(8192, 4096)
    monkeypatch.setenv('MGPI_RIDERS', '1')
    assert riders_pay(32768, 4096) and not riders_pay(100, 1000)
