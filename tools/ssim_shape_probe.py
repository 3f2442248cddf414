"""development: SSIMULACRA2 scores of the library against the oracle over a list of plane shapes (h x w)"""
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: F401,E402

import fixtures as fx  # noqa: E402
import vszip_amd  # noqa: E402
from oracle import oracle as orc  # noqa: E402

orc.build()


def lin(a):
    return a


def pair(shape, seed, sigma=0.03):
    rng = np.random.default_rng(seed)
    ref = [fx.tiled_natural(shape, np.float32, p) for p in range(3)]
    dis = [np.clip(p + rng.normal(0, sigma, p.shape).astype(np.float32), 0, 1).astype(np.float32) for p in ref]
    return ref, dis


dev = vszip_amd.Device(0)
shapes = [tuple(int(v) for v in a.split("x")) for a in sys.argv[1:]]
for shape in shapes:
    ref, dis = pair(shape, 3)
    r = [dev.upload(np.ascontiguousarray(p), 1) for p in ref]
    d = [dev.upload(np.ascontiguousarray(p), 1) for p in dis]
    got = dev.ssimulacra2(r, d)[0]
    want = orc.ssimulacra2(ref, dis)
    print(f"{shape[0]:4d} x {shape[1]:4d}: got {got:.9f} want {want:.9f} diff {got - want:+.3e}", flush=True)
