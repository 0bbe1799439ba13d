"""Import shim: exposes the package directory `disentangled-person-image-generation_amd/`
(whose name is not a valid Python identifier) as the module `dpig_amd`."""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "disentangled-person-image-generation_amd")
_spec = importlib.util.spec_from_file_location(
    "dpig_amd", os.path.join(_PKG_DIR, "__init__.py"), submodule_search_locations=[_PKG_DIR])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["dpig_amd"] = _mod
_spec.loader.exec_module(_mod)
