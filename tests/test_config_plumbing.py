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
