"""Importable alias for the ``det-sam2_amd/`` package directory.

The package lives in ``det-sam2_amd/`` (the hyphen is part of the project name and is not a
valid Python identifier), so this stub points ``det_sam2_amd.__path__`` at that directory
and runs its ``__init__``.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "det-sam2_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
