#!/usr/bin/env python3
"""GPU box: EEDI3 field=1 dh=1 on 16 x 1080p YUV420PS a call, outputs allocated once: frames/s by parameter set, including the paths beside the tuned line kernel
(mdis > 20, hp, mclip with an all-ones / an edge / an empty mask, sclip)."""
import ctypes as C
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
import torch  # noqa: F401,E402

import bench  # noqa: E402
import fixtures as fx  # noqa: E402
import vszip_amd  # noqa: E402
from vszip_amd.capi import Eedi3Params  # noqa: E402

d = vszip_amd.Device(0)
timed = bench.Timed(d, d.sync)
timed.prewarm_s = 0.2
shapes = [(1080, 1920), (540, 960), (540, 960)]
base = [np.ascontiguousarray(fx.tiled_natural(s, np.float32, p)) for p, s in enumerate(shapes)]
frames = 16
srcs = [d.upload(np.roll(pl, 7 * f, axis=1)) for f in range(frames) for pl in base]
dsts = d.eedi3(srcs, 1, dh=True)
n = len(srcs)
table = d.plane_table(srcs, dsts)


def masks(kind):
    out = []
    for f in range(frames):
        for p, s in zip(base, shapes):
            if kind == "ones":
                m = np.full(s, 255, np.uint8)
            elif kind == "none":
                m = np.zeros(s, np.uint8)
            else:  # edges: where the horizontal gradient is large (about a fifth of the samples)
                g = np.abs(np.diff(np.roll(p, 7 * f, axis=1), axis=1, prepend=0))
                m = (g > np.quantile(g, 0.8)).astype(np.uint8) * 255
            out.append(d.upload(np.ascontiguousarray(m)))
    return out


sc = [d.upload(np.ascontiguousarray(np.repeat(np.roll(pl, 7 * f, axis=1), 2, axis=0))) for f in range(frames) for pl in base]
cases = [("defaults", {}, None, None), ("mdis 10", dict(mdis=10), None, None), ("mdis 21", dict(mdis=21), None, None), ("mdis 24", dict(mdis=24), None, None),
         ("mdis 30", dict(mdis=30), None, None), ("mdis 40", dict(mdis=40), None, None), ("nrad 3", dict(nrad=3), None, None), ("hp", dict(hp=True), None, None),
         ("vcheck 0", dict(vcheck=0), None, None), ("sclip", {}, sc, None), ("mclip all ones", {}, None, "ones"), ("mclip edges (20 %)", {}, None, "edges"),
         ("mclip empty", {}, None, "none")]
for name, kw, scl, mk in cases:
    a = dict(dh=1, alpha=0.2, beta=0.25, gamma=20.0, nrad=2, mdis=20, hp=0, vcheck=2, vthresh0=32.0, vthresh1=64.0, vthresh2=4.0)
    a.update({k: int(v) if isinstance(v, bool) else v for k, v in kw.items()})
    prm = Eedi3Params(a["dh"], a["alpha"], a["beta"], a["gamma"], a["nrad"], a["mdis"], a["hp"], a["vcheck"], a["vthresh0"], a["vthresh1"], a["vthresh2"])
    sp = (C.c_void_p * n)(*[s.ptr for s in scl]) if scl else None
    ss = (C.c_ssize_t * n)(*[s.stride for s in scl]) if scl else None
    ml = masks(mk) if mk else None
    mp = (C.c_void_p * n)(*[m.ptr for m in ml]) if ml else None
    ms = (C.c_ssize_t * n)(*[m.stride for m in ml]) if ml else None

    def step():
        if ml:
            d.check(d.lib.vszip_eedi3_mclip(d.ctx, table, sp, ss, mp, ms, n, 1, 0, C.byref(prm)))
        else:
            d.check(d.lib.vszip_eedi3(d.ctx, table, sp, ss, n, 1, 0, C.byref(prm)))

    _, region_ms, *_ = timed.run(step, 3, 1)
    print(f"{name:22s} {frames * 3 / (region_ms * 1e-3):8.1f} fps", flush=True)
    del ml
