"""CPU-side check: the C-ABI library loads and exports every symbol that
include/vszip_hip.h declares (no compute calls; no GPU needed)."""
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def test_library_exports_every_declared_symbol():
    import vszip_amd

    lib = vszip_amd.capi.load()
    header = (ROOT / "include" / "vszip_hip.h").read_text()
    declared = set(re.findall(r"\b(vszip_[a-z0-9_]+)\s*\(", header))
    declared -= {"vszip_ctx", "vszip_plane"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"libvszip_hip.so does not export {name}"
    assert declared == set(vszip_amd.capi.SYMBOLS), (declared ^ set(vszip_amd.capi.SYMBOLS))
    assert lib.vszip_abi_version() == 1


def test_no_gpu_fails_loudly():
    """Without a GPU the context refuses to come up; nothing falls back to the CPU."""
    import pytest
    import vszip_amd

    lib = vszip_amd.capi.load()
    import ctypes as C

    ctx = C.c_void_p()
    rc = lib.vszip_ctx_create(0, C.byref(ctx))
    if rc == 0:  # a GPU is present (GPU box): fine, clean up
        lib.vszip_ctx_destroy(ctx)
        pytest.skip("GPU present")
    assert rc < 0 and not ctx.value
