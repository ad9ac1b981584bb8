"""Importable alias for the package directory ``aocl-compression_amd/`` (hyphenated per the repo layout)."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "aocl-compression_amd")]
__package__ = __name__
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
