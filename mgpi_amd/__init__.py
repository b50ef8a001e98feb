"""Importable name of the package that lives in ``muses-gaussian-process-interpolator_amd/``.

A hyphen cannot appear in a Python import statement, so this shim points its ``__path__`` at the real
package directory and runs that directory's ``__init__``: ``import mgpi_amd`` == the real package,
``mgpi_amd.gp`` / ``mgpi_amd._lib`` resolve to the files in there.
"""
import os as _os

_REAL = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      'muses-gaussian-process-interpolator_amd')
__path__ = [_REAL]
with open(_os.path.join(_REAL, '__init__.py')) as _src:
    exec(compile(_src.read(), _os.path.join(_REAL, '__init__.py'), 'exec'))
del _src
