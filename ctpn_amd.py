"""Import alias: `import ctpn_amd` loads the package that lives in ./text-detection-ctpn_amd/.

The directory name is fixed by the project layout and is not a valid Python identifier, so this
shim loads it under the importable name `ctpn_amd` (one module object, relative imports intact).
"""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "text-detection-ctpn_amd")
_spec = importlib.util.spec_from_file_location(
    "ctpn_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["ctpn_amd"] = _mod
_spec.loader.exec_module(_mod)
