import sys, time
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, bench, vszip_amd
dev = vszip_amd.Device(0)
ref, dis = bench.yuv420p8_pair(3840, 2160)
r, d = [], []
for p in range(16):
    r += [dev.upload(np.roll(x, p * 8, axis=1)) for x in ref]
    d += [dev.upload(np.roll(x, p * 8, axis=1)) for x in dis]
for lin in (1, 0, 1, 0):
    fmt = dev.ssim_source("YUV", np.uint8, 8, ssw=1, ssh=1, matrix=1, chroma_loc=0)
    fmt.linearize = lin
    for _ in range(2): dev.ssimulacra2_src(fmt, r, d)
    t = time.perf_counter()
    for _ in range(6): dev.ssimulacra2_src(fmt, r, d)
    dt = (time.perf_counter() - t) / 6
    print("linearize", lin, round(dt*1e6), "us per 16 pairs", round(16/dt), "pairs/s", flush=True)
