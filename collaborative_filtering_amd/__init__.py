"""Importable alias of the `collaborative-filtering_amd/` directory (a hyphen
cannot appear in a Python package name).  Submodules resolve through __path__."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "collaborative-filtering_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _os, _f
