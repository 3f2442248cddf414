import sys, os, time, json
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, bench, vszip_amd, fixtures as fx
dev = vszip_amd.Device(0)
def run(planes, tag):
    for env in ("1", ""):
        if env: os.environ["VSZIP_MINMAX_SINGLE_READ"]="1"
        else: os.environ.pop("VSZIP_MINMAX_SINGLE_READ", None)
        for _ in range(3): r = dev.plane_minmax(planes, 0.1, 0.1)
        t=time.perf_counter()
        for _ in range(30): r = dev.plane_minmax(planes, 0.1, 0.1)
        dt=(time.perf_counter()-t)/30
        print(tag, "single" if env else "two_sweeps", round(dt*1e6,1), "us/call", r[0][:3], r[1][:3], flush=True)
frames=16
base = bench.make_frame(7, 3840, 2160)
run([dev.upload(np.roll(p, f*3, axis=1)) for f in range(frames) for p in base], "noise")
nat = bench.natural_frame(3840, 2160)
run([dev.upload(np.roll(p, f*3, axis=1)) for f in range(frames) for p in nat], "natural*257")
rng=np.random.default_rng(1)
nat2=[(p.astype(np.int64) - 128 + rng.integers(0,256,p.shape)).clip(0,65535).astype(np.uint16) for p in nat]
run([dev.upload(np.roll(p, f*3, axis=1)) for f in range(frames) for p in nat2], "natural+lowbits")
