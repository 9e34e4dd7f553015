"""Import shim: the package directory is `scheduler-plugins_amd/` (hyphen, mirroring the upstream
project name), which Python cannot import by name.  Importing `scheduler_plugins_amd` loads that
directory as a regular package under this module name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scheduler-plugins_amd")
_spec = importlib.util.spec_from_file_location(
    "scheduler_plugins_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["scheduler_plugins_amd"] = _mod
_spec.loader.exec_module(_mod)
