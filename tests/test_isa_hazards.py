"""Static check of the generated gfx950 code (no GPU): the data registers of a 16-byte buffer store must not be written
by a VALU instruction within the next 2 wait states. gfx950 reads them over the following cycles, and LLVM's hazard
recogniser does not guard buffer stores whose soffset is an SGPR — which is how every row store of the BoxBlur ring
kernels is addressed. A violation stores whatever was written next in the last lanes of every 16 (seen twice while
writing boxblur_ctf.hip: DESIGN.md section 3.2); tools/scan_store_hazard.py walks the disassembly."""
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
BUILD = ROOT / "vapoursynth-zip_amd" / "csrc" / "_build"
LLVM = Path("/opt/rocm/lib/llvm/bin")
OBJECTS = ["boxblur_ctf", "boxblur_ct_u16_a", "boxblur_ct_u16_b", "boxblur_ct_u16_c", "boxblur_ct_u8_a", "boxblur_ct_u8_b", "boxblur_ct_u8_c"]


@pytest.fixture(scope="module")
def built():
    if not (LLVM / "llvm-objdump").exists():
        pytest.skip("no ROCm LLVM tools")
    if not all((BUILD / f"{o}.o").is_file() for o in OBJECTS):
        sys.path.insert(0, str(ROOT))
        import __graft_entry__ as g

        g.build()
    return BUILD


@pytest.mark.parametrize("obj", OBJECTS)
def test_no_valu_write_to_store_data_within_two_wait_states(built, obj, tmp_path):
    fat, dev, dis = tmp_path / "fat.bin", tmp_path / "dev.co", tmp_path / "dev.s"
    subprocess.run([str(LLVM / "llvm-objcopy"), f"--dump-section=.hip_fatbin={fat}", str(built / f"{obj}.o")], check=True, capture_output=True)
    subprocess.run([str(LLVM / "clang-offload-bundler"), "--type=o", "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--input={fat}", f"--output={dev}", "--unbundle"],
                   check=True, capture_output=True)
    with open(dis, "w") as f:
        subprocess.run([str(LLVM / "llvm-objdump"), "-d", "--no-show-raw-insn", str(dev)], check=True, stdout=f)
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "scan_store_hazard.py"), str(dis)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout[-2000:]
    assert "16-byte stores:" in r.stdout
    if "u8" not in obj:  # (8-bit clips store 8 bytes per lane: no 16-byte stores to find there)
        assert "16-byte stores: 0," not in r.stdout, r.stdout[-300:]
