"""Import shim: the package directory is named `scimlsensitivity.jl_amd/` (not a valid Python identifier),
so `import scimlsensitivity_jl_amd` loads it from that directory under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "scimlsensitivity.jl_amd")
_spec = importlib.util.spec_from_file_location(
    "scimlsensitivity_jl_amd", os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["scimlsensitivity_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
