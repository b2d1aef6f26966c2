"""Importable alias for the package directory `multicam-calibration_amd/`.

The repository layout names the package `multicam-calibration_amd` (a hyphen cannot be
imported), so this stub points `__path__` at that directory and runs its `__init__`."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "multicam-calibration_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _f
