"""Import shim: exposes the package in `vapoursynth-zip_amd/` as module `vszip_amd`."""
import importlib.util
import sys
from pathlib import Path

_pkg = Path(__file__).resolve().parent / "vapoursynth-zip_amd"
_spec = importlib.util.spec_from_file_location(
    "vszip_amd", _pkg / "__init__.py", submodule_search_locations=[str(_pkg)]
)
_mod = importlib.util.module_from_spec(_spec)
sys.modules["vszip_amd"] = _mod
_spec.loader.exec_module(_mod)
