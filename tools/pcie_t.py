import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "tests"))
import torch, bench, vszip_amd
for n in (2, 4, 8):
    r = bench.pcie_boxblur(vszip_amd, 0, 13, nctx=n, rounds=24)
    print(n, round(r["value"]), "fps", round(r["pcie_GBps_each_direction"], 1), "GB/s each way")
