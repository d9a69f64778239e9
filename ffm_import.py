"""Import helper: loads the package directory ``firefoam-dev_amd`` (hyphen in the name)
as the module ``firefoam_dev_amd``.  Usage: ``from ffm_import import ffm``."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
_PKG = os.path.join(_ROOT, "firefoam-dev_amd")


def _load():
    name = "firefoam_dev_amd"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(
        name, os.path.join(_PKG, "__init__.py"), submodule_search_locations=[_PKG])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod


ffm = _load()
