"""BASELINE.json config 0 (C1): the INI plumbing - parse_table on an ascii table with prior cuts and a row-stride
downsample, parse_interpolator by class name - on a synthetic stand-in for the reference's missing data file
(etc/uiuc/equation_of_state.csv.gz is listed in .MISSING_LARGE_BLOBS).  Option names, prior bounds
and kernel values are those of etc/example-ascii-data.ini and etc/example-interpolator.ini.  The golden C1 fixture
drops that file's Matern term (it is NaN at zero distance in the reference); the last test
runs all three kernels with the product's zero-separation limit."""
import os

import numpy as np
import pytest

import mgpi_amd as mgpi
from conftest import load_golden, relmax

DATA_INI = """[UIUC]
type = ascii
path = table.csv
x_columns = muB_MeV T_MeV
f_column = speed_of_sound
prior T_MeV = 0 100
prior muB_MeV = 0 150
downsample = 2
"""

INTERP_INI = """[Interpolator]
type = Interpolator
kernel = Kernel2
nugget = Kernel1

[Kernel1]
type = WhiteNoiseKernel
sigma = 1e-4

[Kernel2]
type = SquaredExponentialKernel
sigma = 1e-1
length0 = 10
length1 = 20
"""


def write_case(tmp_path):
    """the C1 fixture's 32 x 32 grid inside the prior box plus rows outside it that the priors must cut"""
    g = load_golden('c1_n512_d2')
    g0 = np.linspace(0.0, 150.0, 32)
    g1 = np.linspace(30.0, 100.0, 32)
    G0, G1 = np.meshgrid(g0, g1, indexing='ij')
    full = np.transpose([G0.ravel(), G1.ravel()])
    ff = np.zeros(len(full))
    ff[::2] = g['f']                                   # the fixture kept every second row
    rows = np.column_stack([full, ff])
    outside = np.array([[200.0, 50.0, 9.0], [10.0, 150.0, 9.0], [-5.0, 50.0, 9.0]])
    table = np.vstack([rows, outside])
    np.savetxt(tmp_path / 'table.csv', table, delimiter=',', header='muB_MeV,T_MeV,speed_of_sound', comments='')
    (tmp_path / 'data.ini').write_text(DATA_INI)
    (tmp_path / 'interp.ini').write_text(INTERP_INI)
    return g


def test_parse_table_priors_then_downsample(tmp_path):
    g = write_case(tmp_path)
    (x, f), (xcols, fcol) = mgpi.parse_table(str(tmp_path / 'data.ini'))
    assert xcols == ['muB_MeV', 'T_MeV'] and fcol == 'speed_of_sound'
    assert x.shape == (512, 2) and f.shape == (512,)
    assert not x.flags['C_CONTIGUOUS']                # F-ordered view like the reference's (io.py:123)
    assert np.array_equal(x, g['x']) and np.array_equal(f, g['f'])
    with pytest.raises(AssertionError):
        mgpi.parse_table(str(tmp_path / 'data.ini'), section='Nope')


def test_parse_interpolator_by_class_name(tmp_path):
    write_case(tmp_path)
    it = mgpi.parse_interpolator(str(tmp_path / 'interp.ini'))
    assert type(it) is mgpi.Interpolator
    assert type(it.kernel) is mgpi.SquaredExponentialKernel and list(it.kernel.params) == [0.1, 10.0, 20.0]
    assert type(it.nugget) is mgpi.WhiteNoiseKernel and it.nugget.params[0] == 1e-4
    # several kernel sections -> CombinedKernel in file order; the nugget really is optional here
    (tmp_path / 'two.ini').write_text(INTERP_INI.replace('kernel = Kernel2\nnugget = Kernel1', 'kernel = Kernel2 Kernel1'))
    it2 = mgpi.parse_interpolator(str(tmp_path / 'two.ini'))
    assert isinstance(it2.kernel, mgpi.CombinedKernel) and it2.nugget is None
    assert [type(k).__name__ for k in it2.kernel.kernels] == ['SquaredExponentialKernel', 'WhiteNoiseKernel']


def test_ascii_round_trip(tmp_path):
    data = np.zeros(4, dtype=[('a', float), ('b', float)])
    data['a'] = [1, 2, 3, 4]
    data['b'] = [0.5, 0.25, 0.125, 0.0625]
    mgpi.save_ascii_data(str(tmp_path / 't.csv'), data)
    back = mgpi.load_ascii_data(str(tmp_path / 't.csv'))
    assert back.dtype.names == ('a', 'b') and np.array_equal(back['b'], data['b'])


@pytest.mark.gpu
def test_c1_end_to_end_from_ini_files(tmp_path):
    """parse_table -> parse_interpolator -> condition / compress / predict against the reference's golden C1 outputs"""
    from conftest import mean_tol, prior_var, var_tol
    g = write_case(tmp_path)
    (x, f), _ = mgpi.parse_table(str(tmp_path / 'data.ini'))
    it = mgpi.parse_interpolator(str(tmp_path / 'interp.ini'))
    mean, cov = it.condition(g['xs'], x, f)
    cond = float(g['cond_K'])
    assert relmax(mean, g['cond_mean']) < mean_tol(cond)
    assert np.max(np.abs(cov - g['cond_cov'])) / prior_var(g) < var_tol(cond)
    assert relmax(it.predict(g['xs'], x, it.compress(x, f)), g['predict_mean']) < mean_tol(cond)


MATERN_SECTION = """
[Kernel3]
type = MaternKernel
order = 2
sigma = 1e-1
length0 = 15
length1 = 10
"""


def test_parse_matern_section(tmp_path):
    write_case(tmp_path)
    (tmp_path / 'three.ini').write_text(INTERP_INI.replace('kernel = Kernel2', 'kernel = Kernel2 Kernel3') + MATERN_SECTION)
    it = mgpi.parse_interpolator(str(tmp_path / 'three.ini'))
    assert [type(k).__name__ for k in it.kernel.kernels] == ['SquaredExponentialKernel', 'MaternKernel']
    assert list(it.kernel.kernels[1].params) == [2.0, 0.1, 15.0, 10.0]
    assert [t[0] for t in it.kernel.terms()] == [0, 2] and it.kernel.terms()[1][3] == 2.0


@pytest.mark.gpu
def test_all_three_example_kernels_end_to_end(tmp_path):
    """SE + Matern(2) kernel, WhiteNoise nugget, from INI text to a conditioned process; oracle with MATERN_ZERO_LIMIT"""
    from conftest import mean_tol, var_tol
    from oracle import mgpi_oracle as oracle
    g = write_case(tmp_path)
    (tmp_path / 'three.ini').write_text(INTERP_INI.replace('kernel = Kernel2', 'kernel = Kernel2 Kernel3') + MATERN_SECTION)
    (x, f), _ = mgpi.parse_table(str(tmp_path / 'data.ini'))
    it = mgpi.parse_interpolator(str(tmp_path / 'three.ini'))
    xs = g['xs'][:64]
    mean, cov = it.condition(xs, x, f)
    kt = [oracle.term_se(0.1, 10, 20), oracle.term_matern(2, 0.1, 15, 10)]
    nt = [oracle.term_white(1e-4)]
    oracle.MATERN_ZERO_LIMIT = True
    try:
        cond = np.linalg.cond(oracle.x2cov(kt + nt, x, x))
        want_mean, want_cov = oracle.condition(kt, nt, xs, x, f)
    finally:
        oracle.MATERN_ZERO_LIMIT = False
    assert relmax(mean, want_mean) < mean_tol(cond)
    assert np.max(np.abs(cov - want_cov)) / 0.02 < var_tol(cond)


def test_containers_round_trip_with_the_reference_field_names(tmp_path):
    """save/load_compressed_data and save/load_conditioned_data as `.npz` mirrors of mgpi/io.py:294-387: the reference's
    field names, shapes and return order; no pickle; the HDF spelling raises ImportError without h5py, like upstream"""
    rng = np.random.default_rng(0)
    source_x, params, compressed = rng.random((7, 2)), rng.random((3, 4)), rng.random((3, 7))
    path = str(tmp_path / 'model.npz')
    mgpi.save_compressed_data(path, source_x, compressed, params, xcols=['muB_MeV', 'T_MeV'], fcol='speed_of_sound')
    with np.load(path, allow_pickle=False) as raw:
        assert sorted(raw.files) == ['compressed', 'fcol', 'params', 'source_x', 'xcols']
    sx, comp, par, xcols, fcol = mgpi.load_compressed_data(path)
    assert np.array_equal(sx, source_x) and np.array_equal(comp, compressed) and np.array_equal(par, params)
    assert xcols == ['muB_MeV', 'T_MeV'] and fcol == 'speed_of_sound'
    mgpi.save_compressed_data(path, source_x, compressed, params)
    assert mgpi.load_compressed_data(path)[3:] == (['x0', 'x1'], 'f')                 # io.py:329-330 defaults
    target_x, mean, cov = rng.random((5, 2)), rng.random(5), rng.random((5, 5))
    out = str(tmp_path / 'out.npz')
    mgpi.save_conditioned_data(out, target_x, mean, xcols=['a', 'b'], fcol='y')
    m, c, tx, xc, fc = mgpi.load_conditioned_data(out)
    assert c is None and np.array_equal(m, mean) and np.array_equal(tx, target_x) and (xc, fc) == (['a', 'b'], 'y')
    mgpi.save_conditioned_data(out, target_x, mean, cov=cov)
    assert np.array_equal(mgpi.load_conditioned_data(out)[1], cov)
    try:
        import h5py  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError):
            mgpi.save_compressed_data(str(tmp_path / 'model.hdf5'), source_x, compressed, params)
        with pytest.raises(ImportError):
            mgpi.load_conditioned_data(str(tmp_path / 'out.h5'))
    np.savetxt(tmp_path / 'targets.csv', rng.random((4, 3)), delimiter=',', header='T_MeV,junk,muB_MeV', comments='')
    t = mgpi.load_target_data(str(tmp_path / 'targets.csv'), ['muB_MeV', 'T_MeV'])
    assert t.shape == (4, 2)
    with pytest.raises(AssertionError):
        mgpi.load_target_data(str(tmp_path / 'targets.csv'), ['muB_MeV', 'missing'])


@pytest.mark.gpu
def test_compress_then_predict_flows_against_the_oracle(tmp_path):
    """bin/mgpi-compress:60-82 -> container -> bin/mgpi-predict:58-94 as tools/mgpi_flows.py: P kernel-parameter rows,
    one compressed vector each (x resident), saved with the reference's field names; predict averages the P means.
    Every row is checked against the oracle's compress / predict."""
    import subprocess
    import sys
    from conftest import ROOT
    from oracle import mgpi_oracle as oracle
    write_case(tmp_path)
    rows = np.array([[0.1, 10.0, 20.0], [0.12, 9.0, 25.0], [0.08, 14.0, 18.0]])
    np.savetxt(tmp_path / 'rows.csv', rows, delimiter=',', header='sigma,length0,length1', comments='')
    rng = np.random.default_rng(3)
    targets = np.column_stack([rng.uniform(0, 150, 40), rng.uniform(30, 100, 40)])
    np.savetxt(tmp_path / 'targets.csv', targets, delimiter=',', header='muB_MeV,T_MeV', comments='')
    tool = os.path.join(ROOT, 'tools', 'mgpi_flows.py')
    run = lambda *a: subprocess.run([sys.executable, tool] + list(a), cwd=str(tmp_path), capture_output=True, text=True, timeout=600)
    out = run('compress', 'data.ini', 'interp.ini', 'model.npz', '--kernel-params', 'rows.csv', '-v')
    assert out.returncode == 0, out.stderr[-2000:]
    assert 'generating compressed data for 3 parameter sets' in out.stdout
    source_x, compressed, params, xcols, fcol = mgpi.load_compressed_data(str(tmp_path / 'model.npz'))
    assert compressed.shape == (3, 512) and np.array_equal(params, rows) and xcols == ['muB_MeV', 'T_MeV'] and fcol == 'speed_of_sound'
    (sx, sf), _ = mgpi.parse_table(str(tmp_path / 'data.ini'))
    assert np.array_equal(sx, source_x)
    nug = [oracle.term_white(1e-4)]
    means = []
    for row, alpha in zip(rows, compressed):
        kern = [oracle.term_se(row[0], row[1], row[2])]
        ref = oracle.compress(kern, nug, sx, sf)
        assert relmax(alpha, ref) < 1e-6                      # cond(K) ~ 2.5e7 for this table (conftest.mean_tol applies to means)
        means.append(oracle.predict(kern, targets, sx, ref))
    out = run('predict', 'model.npz', 'targets.csv', 'interp.ini', 'pred.npz', '-v')
    assert out.returncode == 0, out.stderr[-2000:]
    mean, cov, tx, xc, fc = mgpi.load_conditioned_data(str(tmp_path / 'pred.npz'))
    assert cov is None and np.array_equal(tx, targets) and (xc, fc) == (xcols, fcol)
    assert relmax(mean, np.mean(means, axis=0)) < 1e-8
    out = run('condition', 'data.ini', 'targets.csv', 'interp.ini', 'cond.npz', '--diag')
    assert out.returncode == 0, out.stderr[-2000:]
    m2, var, _, _, _ = mgpi.load_conditioned_data(str(tmp_path / 'cond.npz'))
    rm, rc = oracle.condition([oracle.term_se(0.1, 10.0, 20.0)], nug, targets, sx, sf)
    assert relmax(m2, rm) < 1e-8 and np.max(np.abs(var - np.diag(rc))) < 1e-8 * 0.01
