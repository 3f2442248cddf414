"""End-to-end throughput of the VapourSynth plugin (libvszip.so) through the VapourSynth-free host
tests/fakevs: worker threads call getFrame like fmParallel does; every frame crosses PCIe both ways
from/to ordinary (pageable) frame memory. Prints one JSON line per (filter, threads)."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import fixtures as fx  # noqa: E402
from fakevs import fakevs as vs  # noqa: E402


def clip_4k16(n):
    base = [fx.splitmix64_plane(p, s, np.uint16) for p, s in enumerate([(2160, 3840), (1080, 1920), (1080, 1920)])]
    return vs.source([[np.roll(p, 7 * f, axis=1) for p in base] for f in range(n)], vs.YUV420P16)


def clip_1080p8(n):
    base = [fx.tiled_natural(s, np.uint8, p) for p, s in enumerate([(1080, 1920), (540, 960), (540, 960)])]
    return vs.source([[np.roll(p, 5 * f, axis=1) for p in base] for f in range(n)], vs.YUV420P8)


def clip_1080ps(n):
    base = [np.ascontiguousarray(fx.tiled_natural(s, np.float32, p)) for p, s in enumerate([(1080, 1920), (540, 960), (540, 960)])]
    return vs.source([[np.roll(p, 5 * f, axis=1) for p in base] for f in range(n)], vs.YUV420PS)


def clips_4k_rgbs(n):
    base = [np.ascontiguousarray(fx.tiled_natural((2160, 3840), np.float32, p)) for p in range(3)]
    rng = np.random.default_rng(1)
    noise = rng.normal(0, 0.02, (2160, 3840)).astype(np.float32)
    ref = [[np.roll(p, 9 * f, axis=1) for p in base] for f in range(n)]
    dis = [[np.clip(p + noise, 0, 1) for p in fr] for fr in ref]
    return vs.source(ref, vs.RGBS, props={"_Transfer": 8}), vs.source(dis, vs.RGBS, props={"_Transfer": 8})


def clips_4k_rgb24(n):
    base = [fx.tiled_natural((2160, 3840), np.uint8, p) for p in range(3)]
    rng = np.random.default_rng(1)
    noise = rng.integers(-3, 4, (2160, 3840), dtype=np.int16)
    ref = [[np.roll(p, 9 * f, axis=1) for p in base] for f in range(n)]
    dis = [[np.clip(p.astype(np.int16) + noise, 0, 255).astype(np.uint8) for p in fr] for fr in ref]
    return vs.source(ref, vs.RGB24), vs.source(dis, vs.RGB24)


def clip_8k_rgbs(n):
    base = [np.ascontiguousarray(fx.tiled_natural((4320, 7680), np.float32, p)) for p in range(3)]
    return vs.source([[np.roll(p, 19 * f, axis=1) for p in base] for f in range(n)], vs.RGBS, props={"_Transfer": 8})


def main():
    threads = [int(t) for t in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["1", "4", "8", "16", "32"])]
    vs.lib().fakevs_set_pool_refill(0)  # a real host does not touch recycled frame memory
    src4k = clip_4k16(16)
    src1080 = clip_1080p8(16)
    rec1080 = vs.source([[np.clip(p.astype(np.int16) + 2, 0, 255).astype(np.uint8) for p in [np.asarray(src1080.get_frame(f)[q]) for q in range(3)]] for f in range(16)],
                        vs.YUV420P8)
    only = sys.argv[2].split(",") if len(sys.argv) > 2 else None
    legs = [
        ("BoxBlur r=13 4K YUV420P16", src4k.vszip.BoxBlur(hradius=13, vradius=13), 25 * 2, 128),
        ("Bilateral sigmaS=2 sigmaR=0.02 4K YUV420P16", src4k.vszip.Bilateral(sigmaS=2.0, sigmaR=0.02), 25 * 2, 64),
        ("PlaneAverage 4K YUV420P16", src4k.vszip.PlaneAverage(exclude=[-1]), 25, 128),
        ("XPSNR 1080p YUV420P8", src1080.vszip.XPSNR(rec1080), 3.1 * 2, 256),
    ]
    if only is None or "1080p" in only:
        # round 3: what most scripts run — 1080p YUV420P8 (3.1 MB per frame each way)
        legs.append(("BoxBlur r=2 1080p YUV420P8", src1080.vszip.BoxBlur(hradius=2, vradius=2), 3.1 * 2, 512))
        legs.append(("BoxBlur r=1 2+2 passes 1080p YUV420P8", src1080.vszip.BoxBlur(hradius=1, hpasses=2, vradius=1, vpasses=2), 3.1 * 2, 512))
        legs.append(("Bilateral defaults (sigmaS=3 sigmaR=0.02) 1080p YUV420P8", src1080.vszip.Bilateral(), 3.1 * 2, 512))
        vs.core_standins(True)
        p8 = {"_Matrix": 1, "_ColorRange": 1, "_ChromaLocation": 0}
        fr = [[np.asarray(src1080.get_frame(f)[q]) for q in range(3)] for f in range(8)]
        ya8 = vs.source(fr, vs.YUV420P8, props=p8)
        yb8 = vs.source([[np.clip(p.astype(np.int16) + 2, 0, 255).astype(np.uint8) for p in f] for f in fr], vs.YUV420P8, props=p8)
        legs.append(("SSIMULACRA2 1080p YUV420P8 (device colour pre-stage)", ya8.vszip.SSIMULACRA2(yb8), 6.2, 512))
    if only is None or "eedi3" in only:
        legs.append(("EEDI3 field=1 dh=1 1080p YUV420PS", clip_1080ps(8).vszip.EEDI3(field=1, dh=True), 12.4 + 24.9, 128))
    if only is None or "ssimulacra2" in only:
        ref, dis = clips_4k_rgbs(4)
        legs.append(("SSIMULACRA2 4K RGBS (linear)", ref.vszip.SSIMULACRA2(dis), 199.1, 64))
    if only is None or "ssimulacra2" in only:
        # colour pre-stage on the device: the RGB24 planes go up as they are (50 MB per pair instead of 199);
        # the output clip is the host-converted reference (the test host converts eagerly, outside the clock)
        vs.core_standins(True)
        r8, d8 = clips_4k_rgb24(4)
        legs.append(("SSIMULACRA2 4K RGB24 (device colour pre-stage)", r8.vszip.SSIMULACRA2(d8), 49.8, 64))
    if only is None or "ssimulacra2" in only:
        # round 3: YUV420P8 clips, chroma upsampling + matrix + EOTF on the device (24.9 MB per pair)
        import bench

        yref, ydis = bench.yuv420p8_pair(3840, 2160)
        props = {"_Matrix": 1, "_ColorRange": 1, "_ChromaLocation": 0}
        ya = vs.source([[np.roll(p, 8 * f, axis=1) for p in yref] for f in range(4)], vs.YUV420P8, props=props)
        yb = vs.source([[np.roll(p, 8 * f, axis=1) for p in ydis] for f in range(4)], vs.YUV420P8, props=props)
        legs.append(("SSIMULACRA2 4K YUV420P8 (device colour pre-stage)", ya.vszip.SSIMULACRA2(yb), 24.9, 96))
    if only is None or "limitfilter" in only:
        # round 3: the reference's canonical LimitFilter construction, a vszip chain with a two-input sink (tests/test_int_parity.py:158-167):
        # fused = one upload of src, BoxBlur + LimitFilter on the device, one download
        legs.append(("LimitFilter(flt=src.BoxBlur(2,2), src) 4K YUV420P16 (fused chain)", src4k.vszip.BoxBlur(hradius=2, vradius=2).vszip.LimitFilter(src=src4k, dark_thr=8, bright_thr=8, elast=3), 25 * 2, 96))
        legs.append(("PlaneMinMax(BoxBlur(2,2), minthr=maxthr=0.1) 4K YUV420P16 (fused metric sink)", src4k.vszip.BoxBlur(hradius=2, vradius=2).vszip.PlaneMinMax(minthr=0.1, maxthr=0.1), 25, 96))
    if only is None or "pipeline" in only:
        # BASELINE config 5 as a script writes it: three filter instances, fused into one getFrame by the plugin
        src8k = clip_8k_rgbs(4)
        legs.append(("Pipeline Bilateral->BoxBlur->SSIMULACRA2 8K RGBS (one getFrame per frame when fused)", src8k.vszip.SSIMULACRA2(src8k.vszip.Bilateral(sigmaS=2.0, sigmaR=2.0).vszip.BoxBlur(hradius=2, vradius=2)), 398.1, 16))
    if only is not None:
        legs = [l for l in legs if any(o.lower() in l[0].lower() for o in only)]
    for name, clip, mb_per_frame, count in legs:
        clip.pull(16, 8)  # first touch: LUTs, code objects
        for t in threads:
            # steady state: each worker's GPU context (stream, slabs) exists before the clock starts,
            # as in a VapourSynth session whose worker threads outlive the first frames
            sec = clip.pull(max(count, 8 * t), t, warm_per_thread=3)
            print(json.dumps({"filter": name, "threads": t, "frames_per_s": round(max(count, 8 * t) / sec, 1), "pcie_GBps": round(max(count, 8 * t) * mb_per_frame / 1e3 / sec, 2)}), flush=True)


if __name__ == "__main__":
    main()
