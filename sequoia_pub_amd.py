"""Import shim: the package directory is ``sequoia-pub_amd/`` (a hyphen is not a
legal module name), so ``import sequoia_pub_amd`` loads it from there."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "sequoia-pub_amd")
_spec = importlib.util.spec_from_file_location(
    "sequoia_pub_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["sequoia_pub_amd"] = _mod
_spec.loader.exec_module(_mod)
