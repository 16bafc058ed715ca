"""Import shim: the package directory is `smc.jl_amd/` (not a valid Python identifier), so this module
loads it under the name `smc_jl_amd` and replaces itself in sys.modules."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "smc.jl_amd")
_spec = importlib.util.spec_from_file_location("smc_jl_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["smc_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
