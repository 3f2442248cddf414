"""Repeat-run stress (tools/stress_repeat.py): the same BoxBlur / Bilateral / EEDI3 / SSIMULACRA2 launches many times,
every output compared bit for bit with the first run's. Timing-dependent hardware hazards (the buffer-store data hazard
of DESIGN.md section 3.2 showed up in 1 run of 2) and races between the library's streams appear as rare mismatches."""
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


def test_repeated_launches_give_identical_bits():
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "stress_repeat.py"), "24"], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "TOTAL MISMATCHES 0" in r.stdout, (r.stdout + r.stderr)[-3000:]
