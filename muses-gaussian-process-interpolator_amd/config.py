"""INI-driven construction of tables and interpolators: the plumbing BASELINE.json's config 0 runs
(``etc/example-ascii-data.ini`` + ``etc/example-interpolator.ini`` -> ``parse_table`` -> ``parse_interpolator`` ->
``condition``).  Same entry points, protected option names and return shapes as mgpi/io.py:24-127,391-540; the
objects it builds are THIS package's kernels and interpolators (found by class name through ``factory``, as the
reference does at io.py:408,537).

Table section                      Interpolator file
    [Name]                             [Interpolator]
    type = ascii                       type = Interpolator
    path = table.csv                   kernel = Sec1 Sec2        (several sections -> CombinedKernel)
    x_columns = colA colB              nugget = Sec3             (optional - the reference's io.py:510 makes it
    f_column = colF                                               mandatory by accident; here it really is optional)
    prior colA = min max               [Sec1]
    downsample = 2                     type = SquaredExponentialKernel
                                       sigma = 1e-1      (remaining options = constructor arguments, in file order)

Only the ascii branch (``np.genfromtxt(names=True)``, io.py:177-183) and an additive ``npz`` branch are implemented:
the HDF branches need h5py, which is not installable here, and raise ImportError like
the reference does without h5py.
"""
import os
import warnings
from configparser import ConfigParser

import numpy as np

from .gp import Interpolator
from .kernel_objects import CombinedKernel, Kernel
from .utils import factory

__all__ = ['parse_table', 'parse_kernel_section', 'parse_interpolator', 'load_data', 'load_ascii_data', 'save_ascii_data',
           'load_target_data', 'save_compressed_data', 'load_compressed_data', 'save_conditioned_data', 'load_conditioned_data']

TYPE, PATH, X_COLUMNS, F_COLUMN, PRIOR, DOWNSAMPLE = 'type', 'path', 'x_columns', 'f_column', 'prior', 'downsample'
INTERPOLATOR, KERNEL, NUGGET = 'Interpolator', 'kernel', 'nugget'


ASCII_SUFFIXES = tuple(s + z for s in ('txt', 'dat', 'csv') for z in ('', '.gz'))     # io.py:131-132
HDF_SUFFIXES = ('h5', 'hdf', 'hdf5')                                                      # io.py:134


def _delimiter(path, for_output=False):
    """comma for .csv / .csv.gz, whitespace otherwise (io.py:183,195)"""
    if path.endswith(('csv', 'csv.gz')):
        return ','
    return ' ' if for_output else None


def load_ascii_data(path, verbose=False):
    """structured array from a delimited text file with a header row (io.py:177-184)"""
    if verbose:
        print('loading tabular data from: ' + path)
    return np.genfromtxt(path, names=True, delimiter=_delimiter(path))


def save_ascii_data(path, data, verbose=False):
    """write a structured array as delimited text with a header row (io.py:188-196, minus its misspelt
    `delimeter` keyword, which makes the reference's version raise)"""
    if verbose:
        print('writing: ' + path)
    sep = _delimiter(path, for_output=True)
    np.savetxt(path, np.transpose([data[name] for name in data.dtype.names]), header=sep.join(data.dtype.names),
               comments='', delimiter=sep)


def load_npz_data(path, verbose=False):
    """additive: a .npz of equal-length 1-D columns as a structured array (the HDF containers need h5py)"""
    if verbose:
        print('loading tabular data from: ' + path)
    with np.load(path, allow_pickle=False) as z:
        names = list(z.files)
        out = np.empty(len(z[names[0]]), dtype=[(n, float) for n in names])
        for n in names:
            out[n] = z[n]
        return out


def load_data(path, *args, **kwargs):
    """dispatch on the file suffix (io.py:136-144)"""
    if path.endswith(ASCII_SUFFIXES):
        return load_ascii_data(path, *args, **kwargs)
    if path.endswith('.npz'):
        return load_npz_data(path, *args, **kwargs)
    if path.endswith(HDF_SUFFIXES):
        raise ImportError('could not import h5py')           # io.py:203
    raise ValueError('could not interpret filetype for: ' + path)


def _load_by_type(path, filetype, verbose=False):
    if filetype == 'ascii':
        return load_ascii_data(path, verbose=verbose)
    if filetype == 'npz':
        return load_npz_data(path, verbose=verbose)
    if filetype in ('hdftable', 'stellarcollapse'):
        raise ImportError('could not import h5py')           # io.py:203,233
    raise ValueError('could not understand %s=%s' % (TYPE, filetype))


def _only_section(config, path):
    sections = config.sections()
    assert len(sections), 'no sections found!'
    assert len(sections) == 1, 'must specify a section when multiple exist within %s\n%s' % (path, '\n'.join(sections))
    return sections[0]


def parse_table(path, section=None, verbose=False):
    """INI -> ((source_x (N, D), source_f (N,)), (x column names, f column name)); prior cuts are applied before the
    row-stride downsample, exactly in the reference's order (io.py:98-120)"""
    if verbose:
        print('reading tabular data from: ' + path)
    config = ConfigParser()
    assert config.read(path), 'could not read ' + path
    if section is None:
        section = _only_section(config, path)
    for required in (PATH, X_COLUMNS, F_COLUMN, TYPE):
        assert config.has_option(section, required), 'could not find %s in section=%s' % (required, section)
    xcols = config.get(section, X_COLUMNS).split()
    fcol = config.get(section, F_COLUMN)
    assert len(xcols) == len(set(xcols)), 'cannot have repeated values in %s=%s' % (X_COLUMNS, ', '.join(xcols))
    assert fcol not in xcols, 'cannot have %s=%s in %s=%s' % (F_COLUMN, fcol, X_COLUMNS, ', '.join(xcols))

    # `prior <column> = min max`; configparser lower-cases option names, so columns are matched case-insensitively
    priors = {}
    for option in config.options(section):
        words = option.split()
        if words[0] == PRIOR:
            lo, hi = (float(v) for v in config.get(section, option).split())
            priors[words[1]] = (lo, hi)

    table_path = config.get(section, PATH)
    if not os.path.exists(table_path) and not os.path.isabs(table_path):      # additive: relative to the INI file
        beside = os.path.join(os.path.dirname(os.path.abspath(path)), table_path)
        if os.path.exists(beside):
            table_path = beside
    data = _load_by_type(table_path, config.get(section, TYPE), verbose=verbose)
    if verbose:
        print('    found %d samples' % len(data))
    for col in xcols + [fcol]:
        assert col in data.dtype.names, 'required column=%s not present!' % col

    for col in data.dtype.names:
        if col.lower() in priors:
            lo, hi = priors[col.lower()]
            keep = (lo <= data[col]) & (data[col] <= hi)
            if verbose:
                print('retaining %d samples after imposing: %.3e <= %s <= %.3e' % (np.sum(keep), lo, col, hi))
            data = data[keep]
    if config.has_option(section, DOWNSAMPLE):
        data = data[::config.getint(section, DOWNSAMPLE)]
        if verbose:
            print('    retained %d samples' % len(data))

    # an F-ordered (N, D) view, like the reference's np.transpose([columns]) (io.py:123): the device path copies it
    source_x = np.transpose([data[col] for col in xcols])
    return (source_x, data[fcol]), (xcols, fcol)


def _number(config, section, option):
    for getter in (config.getint, config.getfloat):
        try:
            return getter(section, option)
        except ValueError:
            pass
    raise ValueError('could not parse option=%s in section=%s' % (option, section))


def parse_kernel_section(config, section, verbose=False):
    """one INI section -> one kernel: `type` names the class, the remaining options are the constructor's positional
    arguments in file order (io.py:393-433)"""
    assert config.has_option(section, TYPE), 'could not find %s in section=%s' % (TYPE, section)
    klass = factory(Kernel)[config.get(section, TYPE)]
    args = [_number(config, section, option) for option in config.options(section) if option != TYPE]
    kernel = klass(*args)
    if verbose:
        print('    section=%s -> %s' % (section, kernel))
    return kernel


def _kernel_from_sections(config, names, path, verbose):
    kernels = []
    for name in names:
        assert config.has_section(name), 'can not find section=%s' % name
        try:
            kernels.append(parse_kernel_section(config, name, verbose=verbose))
        except Exception as err:                                             # io.py:476-479: warn and skip
            warnings.warn('could not parse section=%s (%s). Skipping...' % (name, err))
    assert kernels, 'could not find any kernels within: ' + path
    return kernels[0] if len(kernels) == 1 else CombinedKernel(*kernels)


def parse_interpolator(path, verbose=False):
    """[Interpolator] section -> interpolator object of the named class; `kernel` / `nugget` list kernel sections,
    every other option becomes a keyword argument (int, float, bool or string) (io.py:442-540)"""
    if verbose:
        print('reading interpolator config from: ' + path)
    config = ConfigParser()
    assert config.read(path), 'could not read ' + path
    assert config.has_section(INTERPOLATOR), 'interpolator config must have section [%s]' % INTERPOLATOR
    assert config.has_option(INTERPOLATOR, TYPE), 'cannot find %s in section=%s' % (TYPE, INTERPOLATOR)
    assert config.has_option(INTERPOLATOR, KERNEL), 'cannot find %s in section=%s' % (KERNEL, INTERPOLATOR)
    kernel = _kernel_from_sections(config, config.get(INTERPOLATOR, KERNEL).split(), path, verbose)
    nugget = None
    if config.has_option(INTERPOLATOR, NUGGET):
        nugget = _kernel_from_sections(config, config.get(INTERPOLATOR, NUGGET).split(), path, verbose)
    kwargs = {}
    for option in config.options(INTERPOLATOR):
        if option in (TYPE, KERNEL, NUGGET):
            continue
        for getter in (config.getint, config.getfloat, config.getboolean, config.get):
            try:
                kwargs[option] = getter(INTERPOLATOR, option)
                break
            except ValueError:
                pass
    klass = factory(Interpolator)[config.get(INTERPOLATOR, TYPE)]
    if verbose:
        print('instantiating interpolator\n  %s\n  kernel = %s' % (klass.__name__, kernel))
        if nugget is not None:
            print('  nugget = %s' % nugget)
    return klass(kernel, nugget=nugget, **kwargs)


#-------------------------------------------------
# containers: the saved model (compressed data) and the conditioned output (mgpi/io.py:284-387)
#
# The reference writes HDF files with h5py.  The field names and shapes below are the reference's; the container is
#   * HDF (suffix .h5 / .hdf / .hdf5) when h5py is importable - files then interoperate with the reference's tools -
#     ImportError otherwise, as upstream;
#   * an ``.npz`` mirror with the same field names (numpy.load without pickle) - always available.

def _is_npz(path):
    return str(path).endswith('.npz')


def _h5py():
    try:
        import h5py
    except ImportError:
        raise ImportError('could not import h5py')        # io.py:297
    return h5py


def load_target_data(path, xcols, verbose=False):
    """the target_x grid from a table file: (M, len(xcols)) in the order of xcols (io.py:284-290)"""
    data = load_data(path, verbose=verbose)
    for col in xcols:
        assert col in data.dtype.names, 'required column=%s missing in %s' % (col, path)
    return np.transpose([data[col] for col in xcols])


def save_compressed_data(path, source_x, compressed, params, xcols=None, fcol='f', verbose=False):
    """fcol, xcols, source_x (N, D), params (P, num_params), compressed (P, N)   (io.py:317-338)"""
    if verbose:
        print('saving %d compressed datasets into: %s' % (len(compressed), path))
    source_x = np.asarray(source_x, dtype=float)
    nsmp, ndim = source_x.shape
    if xcols is None:
        xcols = ['x%d' % dim for dim in range(ndim)]
    fields = dict(fcol=str(fcol), xcols=[str(c) for c in xcols], source_x=source_x, params=np.asarray(params, dtype=float),
                  compressed=np.asarray(compressed, dtype=float))
    if _is_npz(path):
        np.savez(path, **fields)
        return
    with _h5py().File(path, 'w') as obj:
        for name, val in fields.items():
            obj.create_dataset(name=name, data=val)


def load_compressed_data(path, verbose=False):
    """-> source_x, compressed, params, xcols, fcol   (io.py:294-313)"""
    if verbose:
        print('loading compressed datasets from: ' + path)
    if _is_npz(path):
        with np.load(path, allow_pickle=False) as obj:
            fcol, xcols = str(obj['fcol']), [str(c) for c in obj['xcols']]
            source_x, params, compressed = obj['source_x'], obj['params'], obj['compressed']
    else:
        with _h5py().File(path, 'r') as obj:
            fcol = obj['fcol'][()]
            xcols = [c.decode() if isinstance(c, bytes) else str(c) for c in obj['xcols'][:]]
            source_x, params, compressed = obj['source_x'][:], obj['params'][:], obj['compressed'][:]
            fcol = fcol.decode() if isinstance(fcol, bytes) else str(fcol)
    if verbose:
        print('found %d compressed datasets for %d source_x (%s)' % (len(compressed), len(source_x), ', '.join(xcols)))
    return source_x, compressed, params, xcols, fcol


def save_conditioned_data(path, target_x, mean, cov=None, xcols=None, fcol='f', verbose=False):
    """fcol, xcols, target_x (M, D), mean (M,), optionally cov   (io.py:365-387)"""
    target_x = np.asarray(target_x, dtype=float)
    nsmp, ndim = target_x.shape
    if verbose:
        print('saving conditioned mean, cov for %d locations into: %s' % (nsmp, path))
    if xcols is None:
        xcols = ['x%d' % dim for dim in range(ndim)]
    fields = dict(fcol=str(fcol), xcols=[str(c) for c in xcols], target_x=target_x, mean=np.asarray(mean, dtype=float))
    if cov is not None:
        fields['cov'] = np.asarray(cov, dtype=float)
    if _is_npz(path):
        np.savez(path, **fields)
        return
    with _h5py().File(path, 'w') as obj:
        for name, val in fields.items():
            obj.create_dataset(name=name, data=val)


def load_conditioned_data(path, verbose=False):
    """-> mean, cov (None when absent), target_x, xcols, fcol   (io.py:342-361)"""
    if verbose:
        print('loading conditioned mean, cov from: ' + path)
    if _is_npz(path):
        with np.load(path, allow_pickle=False) as obj:
            fcol, xcols = str(obj['fcol']), [str(c) for c in obj['xcols']]
            target_x, mean = obj['target_x'], obj['mean']
            cov = obj['cov'] if 'cov' in obj.files else None
    else:
        with _h5py().File(path, 'r') as obj:
            fcol = obj['fcol'][()]
            fcol = fcol.decode() if isinstance(fcol, bytes) else str(fcol)
            xcols = [c.decode() if isinstance(c, bytes) else str(c) for c in obj['xcols'][:]]
            target_x, mean = obj['target_x'][:], obj['mean'][:]
            cov = obj['cov'][:] if ('cov' in obj.keys()) else None
    return mean, cov, target_x, xcols, fcol
