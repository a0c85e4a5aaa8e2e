"""Import shim: `import mfsgd_amd` loads the package directory
``matrixfactorizationsgd.java_amd/`` (its name contains a dot, so the normal
import statement cannot reach it) and installs it under this module name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "matrixfactorizationsgd.java_amd")
_spec = importlib.util.spec_from_file_location(
    "mfsgd_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["mfsgd_amd"] = _mod
_spec.loader.exec_module(_mod)
