"""Import alias: ``import ngf_amd`` loads the package directory ``neural-gauge-fields_amd/``
(whose name, fixed by the project layout, is not a valid Python identifier)."""
import importlib.util as _u
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "neural-gauge-fields_amd")
_spec = _u.spec_from_file_location("ngf_amd", _os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = _u.module_from_spec(_spec)
_sys.modules["ngf_amd"] = _mod
_spec.loader.exec_module(_mod)
