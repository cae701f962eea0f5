"""Importable alias of the package directory `gymnasium-planar-robotics_amd/` (a hyphen is not a valid module name).
`import gymnasium_planar_robotics_amd` loads that directory under this name; nothing else lives here."""
import importlib.util as _ilu
import os as _os
import sys as _sys

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), 'gymnasium-planar-robotics_amd')
_spec = _ilu.spec_from_file_location(__name__, _os.path.join(_real, '__init__.py'), submodule_search_locations=[_real])
_mod = _ilu.module_from_spec(_spec)
_sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
