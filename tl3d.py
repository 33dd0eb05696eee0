"""Import shim: `import tl3d` loads the package that lives in `textureless-3d-reconstruction_amd/`
(the directory name the project layout prescribes is not a valid Python identifier)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "textureless-3d-reconstruction_amd")
_spec = importlib.util.spec_from_file_location("tl3d", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["tl3d"] = _mod
_spec.loader.exec_module(_mod)
