"""Import shim: the package directory is named `pioran.jl_amd/` (not a valid module name), so load it
under the importable name `pioran_jl_amd`.  Usage: `import pioran_jl_amd as pj`."""
import importlib.util
import sys
from pathlib import Path

_pkg_dir = Path(__file__).resolve().parent / "pioran.jl_amd"
_spec = importlib.util.spec_from_file_location(
    "pioran_jl_amd", _pkg_dir / "__init__.py", submodule_search_locations=[str(_pkg_dir)])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["pioran_jl_amd"] = _mod
_spec.loader.exec_module(_mod)
