#!/usr/bin/env python3
"""Headline bench: frames/s of the vszip hot path on MI355X, inputs resident in HBM.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--workload boxblur|bilateral|ssimulacra2|pipeline]
                    [--frames F] [--no-cpu] [--no-others]

Default workload (the JSON line's `value` and `roofline`): vszip.BoxBlur(hradius=vradius=13)
on 3840x2160 YUV420P16 — the north-star roofline target. A step is one pass of the hot path
over one batch of F = 64 synthetic frames already resident in HBM (F*3 = 192 planes = one launch).
The other two headline filters (Bilateral sigmaS=2 sigmaR=2, SSIMULACRA2 ref vs dist) are
measured in the same run on their BASELINE configs and reported under "others", each beside
the CPU oracle timed on this box's host cores.
N > 1: one process per GPU over RCCL. Either the caller starts the ranks (python -m torch.distributed.run
... bench.py --gpus N: RANK / WORLD_SIZE in the environment) or `python bench.py --gpus N` starts them
itself as CHILD processes before anything in this process has touched the GPU, relays rank 0's line
and exits with the children's code. Frames shard across ranks with no data-path collective (weak
scaling: every rank owns its own F frames); the collectives are the max-reduce of the timings and, outside
the timed region, the per-clip XPSNR accumulators. Every N also reports the PCIe-fed rate (pinned host
frames in and out, all ranks at once): config.pcie_fed_fps.
Rank 0 prints ONE JSON line (DESIGN.md section "Measurement").
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))

W4K, H4K = 3840, 2160
W1080, H1080 = 1920, 1080
RADIUS = 13
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
UNTIMED_CALLS = 0


def yuv420_shapes(w, h):
    return [(h, w), (h // 2, w // 2), (h // 2, w // 2)]


def make_frame(seed: int, w: int, h: int):
    """Deterministic noise planes (splitmix64), u16 full range."""
    import fixtures as fx

    return [fx.splitmix64_plane(0x5A170000 + 16 * seed + p, s, np.uint16) for p, s in enumerate(yuv420_shapes(w, h))]


def natural_frame(w: int, h: int):
    """Natural-content YUV420P16-shaped planes: the reference test picture tiled."""
    import fixtures as fx

    planes = [fx.tiled_natural(s, np.uint16, p) for p, s in enumerate(yuv420_shapes(w, h))]
    content = os.environ.get("VSZIP_BENCH_CONTENT", "")  # development: how content-sensitive a kernel is
    if content:
        rng = np.random.default_rng(5)
        if content == "noise":  # full-range white noise: the worst case of every data-dependent lookup
            planes = [rng.integers(0, 65536, p.shape, dtype=np.uint16) for p in planes]
        else:  # "grain<sigma>": the picture plus Gaussian grain of that many 16-bit LSB
            sigma = float(content.replace("grain", "") or 300)
            planes = [np.clip(p + rng.normal(0, sigma, p.shape), 0, 65535).astype(np.uint16) for p in planes]
    return planes


def _lin(v):
    v = v.astype(np.float64)
    return np.where(v <= 0.04045, v / 12.92, ((v + 0.055) / 1.055) ** 2.4).astype(np.float32)


def rgbs_pair(w: int, h: int, seed: int = 1):
    """ref = tiled natural picture (linear light), dist = ref + noise sigma 0.02, clipped."""
    import fixtures as fx

    rng = np.random.default_rng(seed)
    ref = [_lin(fx.tiled_natural((h, w), np.float32, p)) for p in range(3)]
    dis = [np.clip(p + rng.normal(0, 0.02, p.shape).astype(np.float32), 0, 1).astype(np.float32) for p in ref]
    return ref, dis


# ---------------------------------------------------------------------------
# CPU baselines: the oracle (scalar C++ port of the reference arithmetic), one unit
# per thread on the host cores this process may actually USE, bounded sample, with a
# single-thread rate beside it so the record shows how well the threads scaled.
# ---------------------------------------------------------------------------
def _cgroup_cpu_quota():
    """CPUs' worth of time the cgroup grants this process (None: unlimited / unknown).
    cgroup v2 `cpu.max` ("<quota> <period>" | "max <period>") of this process's group and its
    ancestors; cgroup v1 `cpu.cfs_quota_us / cpu.cfs_period_us`."""
    best = None

    def take(q):
        nonlocal best
        if q is not None and q > 0:
            best = q if best is None else min(best, q)

    try:
        rel = ""
        for ln in Path("/proc/self/cgroup").read_text().splitlines():
            parts = ln.split(":", 2)
            if len(parts) == 3 and parts[0] == "0":
                rel = parts[2].strip("/")
        d = Path("/sys/fs/cgroup") / rel if rel else Path("/sys/fs/cgroup")
        seen = 0
        while seen < 32:
            f = d / "cpu.max"
            if f.is_file():
                q, per = (f.read_text().split() + ["100000"])[:2]
                if q != "max":
                    take(float(q) / float(per))
            if d == Path("/sys/fs/cgroup") or d == d.parent:
                break
            d = d.parent
            seen += 1
    except (OSError, ValueError):
        pass
    for base in ("/sys/fs/cgroup/cpu", "/sys/fs/cgroup/cpu,cpuacct"):
        try:
            q = float(Path(base, "cpu.cfs_quota_us").read_text())
            per = float(Path(base, "cpu.cfs_period_us").read_text())
            if q > 0 and per > 0:
                take(q / per)
        except (OSError, ValueError):
            pass
    return best


def effective_cpus(quota=..., affinity=None, nominal=None):
    """{nominal, affinity, quota, effective}: `effective` = the threads that can run at once =
    min(os.cpu_count(), scheduler affinity, ceil(cgroup CPU quota)), at least 1. The GPU boxes of this pool
    show 256 logical CPUs under a 16-CPU quota: 256 threads there are 16 CPUs' worth of time, sliced."""
    import math

    nominal = nominal or os.cpu_count() or 1
    if affinity is None:
        try:
            affinity = len(os.sched_getaffinity(0))
        except (AttributeError, OSError):
            affinity = nominal
    if quota is ...:
        quota = _cgroup_cpu_quota()
    eff = min(nominal, affinity)
    if quota:
        eff = min(eff, max(1, math.ceil(quota - 1e-9)))
    return {"nominal": int(nominal), "affinity": int(affinity), "quota": quota, "effective": max(1, int(eff))}


def cpu_threads() -> int:
    """threads of every CPU leg: the effective CPU count (VSZIP_BENCH_CPU_THREADS overrides, for scaling studies)"""
    ov = os.environ.get("VSZIP_BENCH_CPU_THREADS")
    return max(1, int(ov)) if ov else effective_cpus()["effective"]


def _timed_pool(fn, cores: int, budget_s: float, unit_desc: str, t_single: float, single_budget_s: float = 1.0):
    """First ONE thread alone (whole units for `single_budget_s`, at least one: `single_thread_value`), then
    waves of `cores` units (one per thread) until the budget is used, at least one wave. `scaling` =
    value / (threads x single_thread_value): 1.0 = every thread ran as fast as the lone one; well below means the
    host could not run `cores` threads at once (quota, SMT siblings, memory bandwidth) and says so in the record.
    `t_single` (the caller's untimed first unit, cold) is kept only as `first_unit_s`."""
    n1 = 0
    t0 = time.perf_counter()
    while True:
        fn(0)
        n1 += 1
        dt1 = time.perf_counter() - t0
        if dt1 >= single_budget_s:
            break
    single = n1 / dt1
    n = 0
    with ThreadPoolExecutor(cores) as ex:
        t0 = time.perf_counter()
        while True:
            list(ex.map(fn, range(cores)))
            n += cores
            dt = time.perf_counter() - t0
            if dt >= budget_s or dt + dt / (n // cores) > 2.5 * budget_s:
                break
    value = n / dt
    eff = effective_cpus()
    rec = {"value": value, "unit": "frames/s", "cores": cores, "kind": "port",
           "threads": cores, "single_thread_value": single, "scaling": value / (cores * single),
           "cores_nominal": eff["nominal"], "cores_effective": eff["effective"], "cpu_quota": eff["quota"],
           "first_unit_s": t_single,
           "sample": f"{n} x {unit_desc}, one per thread on {cores} threads ({eff['nominal']} logical CPUs, "
                     f"{eff['effective']} usable: affinity {eff['affinity']}, cgroup quota {eff['quota']}), {dt:.1f}s; "
                     f"one thread alone: {n1} in {dt1:.1f}s"}
    if rec["scaling"] < 0.7:
        rec["scaling_note"] = ("threads ran slower together than alone: SMT siblings share a core's units, all threads share the "
                               "sockets' memory bandwidth, and a cgroup quota slices time when threads > quota")
    return rec


def cpu_boxblur(budget_s=6.0, w=W4K, h=H4K, blank=False):
    from oracle import oracle as orc

    orc.build()
    cores = cpu_threads()
    frame = [np.zeros(sh, np.uint16) for sh in yuv420_shapes(w, h)] if blank else make_frame(0, w, h)

    def one(_):
        for p in frame:
            orc.boxblur(p, RADIUS, 1, RADIUS, 1)

    t0 = time.perf_counter()
    one(0)
    return _timed_pool(one, cores, budget_s, f"{w}x{h} YUV420P16 frame{' (BlankClip)' if blank else ''}, BoxBlur r=13", time.perf_counter() - t0)


def cpu_bilateral(w, h, budget_s=6.0):
    from oracle import oracle as orc

    cores = cpu_threads()
    frame = natural_frame(w, h)
    prm = orc.bilateral_params([2], [2], yuv=True, ssw=1, ssh=1)

    def one(_):
        for i, p in enumerate(frame):
            orc.bilateral_plane(p, prm["sigmaS"][i], prm["sigmaR"][i], prm["algorithm"][i], prm["radius"][i], prm["step"][i], prm["PBFICnum"][i])

    t0 = time.perf_counter()
    one(0)
    return _timed_pool(one, cores, budget_s, f"{w}x{h} YUV420P16 frame, Bilateral sigmaS=2 sigmaR=2", time.perf_counter() - t0)


def cpu_ssimulacra2(w, h, budget_s=8.0):
    from oracle import oracle as orc

    cores = cpu_threads()
    ref, dis = rgbs_pair(w, h)

    def one(_):
        orc.ssimulacra2(ref, dis)

    t0 = time.perf_counter()
    one(0)
    r = _timed_pool(one, cores, budget_s, f"{w}x{h} RGBS pair, SSIMULACRA2", time.perf_counter() - t0)
    r["unit"] = "pairs/s"
    return r


def profile_traffic(kernel_prefix: str, frames: int):
    """HBM bytes per launch of the dominant kernel from the committed PMC passes of this same
    command (profiles/r*_boxblur_pmc.json, written by tools/prof.sh + tools/summarize_prof.py:
    FETCH_SIZE x2 on gfx950 + WRITE_SIZE, separate --pmc runs). None when no matching profile."""
    best = None
    for f in sorted((ROOT / "profiles").glob("r*_boxblur_pmc.json")):
        try:
            j = json.loads(f.read_text())
            if j["bench_line_under_tracing"]["config"]["frames_per_step_per_gpu"] != frames:
                continue
            for k, m in j["kernels"].items():
                if k.startswith(kernel_prefix) and "hbm_traffic_bytes_per_launch" in m:
                    best = m["hbm_traffic_bytes_per_launch"]
        except Exception:
            continue
    return best


def profile_launch_us(kernel_sub: str):
    """the dominant kernel's average launch duration in the newest committed rocprofv3 --kernel-trace --stats summary of this command
    (profiles/r*_boxblur_kernel_stats.csv) - REPLAYED, the second basis of roofline.frac; (us, calls, file name) or None"""
    import csv

    files = sorted((ROOT / "profiles").glob("r*_boxblur_kernel_stats.csv"))
    for f in reversed(files):
        try:
            for r in csv.DictReader(f.open()):
                if kernel_sub in r["Name"]:
                    return float(r["AverageNs"]) / 1e3, int(r["Calls"]), f.name
        except Exception:
            continue
    return None


def limit_from_profile(name: str, kernel_sub: str):
    """What bounds a leg that HBM does not: issue and LDS-array busy fractions of its dominant kernel, from the committed PMC
    passes of the same workload (profiles/r*_<name>_pmc.json, tools/prof_all.sh: separate --pmc runs) — REPLAYED, not measured in
    this run. cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the 8 XCDs); VALU busy = SQ_ACTIVE_INST_VALU x 4 (the SQ counts in
    quad-cycles) / 1024 SIMDs / cycles; LDS array busy = SQ_LDS_IDX_ACTIVE / 256 CUs / cycles (MI355X_MICROARCH.md, LDS:
    SQ_LDS_IDX_ACTIVE = all LDS-array cycles, SQ_LDS_BANK_CONFLICT = the extra ones)."""
    best = None
    for f in sorted((ROOT / "profiles").glob(f"r*_{name}_pmc.json")):
        try:
            for k, m in json.loads(f.read_text())["kernels"].items():
                if kernel_sub in k and "SQ_ACTIVE_INST_VALU" in m and "GRBM_GUI_ACTIVE" in m:
                    best = (f.name, k, m)
        except Exception:
            continue
    if not best:
        return None
    fname, k, m = best
    cycles = m["GRBM_GUI_ACTIVE"] / 8.0
    valu = m["SQ_ACTIVE_INST_VALU"] * 4.0 / 1024.0 / cycles
    out = {"kernel": k, "valu_busy_frac": valu, "source": f"profiles/{fname} (replayed; formulas in bench.limit_from_profile)"}
    if "SQ_LDS_IDX_ACTIVE" in m:
        out["lds_array_busy_frac"] = m["SQ_LDS_IDX_ACTIVE"] / 256.0 / cycles
        if m["SQ_LDS_IDX_ACTIVE"] > 0 and "SQ_LDS_BANK_CONFLICT" in m:
            out["lds_bank_conflict_share"] = m["SQ_LDS_BANK_CONFLICT"] / m["SQ_LDS_IDX_ACTIVE"]
    out["bound"] = "lds" if out.get("lds_array_busy_frac", 0) > valu else "valu"
    out["frac"] = max(valu, out.get("lds_array_busy_frac", 0))
    return out


# ---------------------------------------------------------------------------
# GPU workloads
# ---------------------------------------------------------------------------
class Timed:
    def __init__(self, dev, barrier, prewarm_s=0.0):
        self.dev, self.barrier, self.prewarm_s = dev, barrier, prewarm_s

    def run(self, step, steps, warmup):
        """-> (wall seconds, stream ms over the region, summed dominant-kernel ms, dominant-kernel launches).
        All three GPU figures are HIP events recorded on the stream the kernels are launched on
        (the library's own stream, not torch's): timer_* brackets the whole region, the probe
        brackets every launch of the filter's dominant kernel inside it."""
        # Clocks and power state first: the first launches after idle run up to 10 % slower (rocprofv3: 624 us
        # against 563 us in steady state). Untimed, like the W warm-up steps that follow.
        t_pre = time.perf_counter()
        ncalls = 0
        while self.prewarm_s > 0 and time.perf_counter() - t_pre < self.prewarm_s:
            step()
            self.dev.sync()
            ncalls += 1
        for _ in range(warmup):
            step()
        self.calls = getattr(self, "calls", 0) + ncalls + warmup + steps  # every call of `step` a profiler sees (tools/roofline_check.py)
        # the closing barrier is inside the clock (contract): warm its path first (the first RCCL barriers of a process cost a millisecond each,
        # which a 20-step region of 0.58 ms steps would carry as 8 %), and time one for the record (`config.barrier_ms`)
        for _ in range(3):
            self.barrier()
        tb = time.perf_counter()
        self.barrier()
        self.barrier_ms = (time.perf_counter() - tb) * 1e3
        self.dev.probe_enable(True)
        t0 = time.perf_counter()
        self.dev.timer_start()
        for _ in range(steps):
            step()
        region_ms = self.dev.timer_stop_ms()  # synchronises the stream
        self.barrier()
        dt = time.perf_counter() - t0
        dom_ms, launches, each = self.dev.probe_read_each()
        self.dev.probe_enable(False)
        self.each_ms = each
        return dt, region_ms, dom_ms, launches

    def launch_stats(self, step, min_seconds=1.0, max_launches=20000):
        """min / median / max duration of the dominant kernel's launches over at least `min_seconds` of
        kernel time (the contract's K-step region can be a few ms; this sample is not part of `value`)."""
        each = list(getattr(self, "each_ms", []))
        while sum(each) < min_seconds * 1e3 and len(each) < max_launches:
            self.dev.probe_enable(True)
            self.calls = getattr(self, "calls", 0) + 64
            for _ in range(64):
                step()
            _, _, e = self.dev.probe_read_each()
            self.dev.probe_enable(False)
            if not e:
                break
            each += e
        if not each:
            return None
        a = np.sort(np.asarray(each, np.float64)) * 1e3
        return {"min": float(a[0]), "median": float(a[len(a) // 2]), "max": float(a[-1]), "mean": float(a.mean()), "p10": float(a[len(a) // 10]),
                "p90": float(a[(9 * len(a)) // 10]), "n": int(len(a)), "sample_s": float(a.sum() * 1e-6)}


class Arena:
    """A batch of planes inside one device allocation: plane k starts on a 2 MiB boundary plus a pseudo-random multiple
    of 256 B below 1 MiB (planes that start on identical offsets inside their pages collide more often)."""

    def __init__(self, dev, shapes, dtype, seed, ptr=None):
        import ctypes as C

        rng = np.random.default_rng(seed)
        isz = np.dtype(dtype).itemsize
        spacing = int(float(os.environ.get("VSZIP_BENCH_PLANE_SPACING_MIB", "0")) * (1 << 20))  # extra distance between consecutive planes
        self.offs, total = [], 0
        for h, w in shapes:
            total = (total + (2 << 20) - 1) // (2 << 20) * (2 << 20)
            o = total + int(rng.integers(0, 4096)) * 256
            self.offs.append(o)
            total = o + h * w * isz + spacing
        self.dev, self.shapes, self.dtype, self.nbytes, self.ptr, self.planes = dev, shapes, dtype, total + 256, 0, []
        if ptr is None:
            p = C.c_void_p()
            dev.check(dev.lib.vszip_dev_alloc(dev.ctx, self.nbytes, C.byref(p)))
            ptr = p.value
        self.bind(ptr)

    def bind(self, ptr):
        from vszip_amd.capi import DevPlane

        self.ptr = ptr
        self.planes = [DevPlane(self.dev, ptr + o, w, h, w, self.dtype, own=False) for o, (h, w) in zip(self.offs, self.shapes)]
        return self

    def view(self, ptr):
        """the same layout on another allocation (not owned)"""
        v = Arena.__new__(Arena)
        v.dev, v.shapes, v.dtype, v.nbytes, v.offs = self.dev, self.shapes, self.dtype, self.nbytes, self.offs
        v.ptr = 0  # a view never frees
        from vszip_amd.capi import DevPlane

        v.planes = [DevPlane(self.dev, ptr + o, w, h, w, self.dtype, own=False) for o, (h, w) in zip(self.offs, self.shapes)]
        return v

    def free(self):
        if self.ptr:
            self.dev.lib.vszip_dev_free(self.dev.ctx, self.ptr)
            self.ptr = 0

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


def placed_batch(dev, host_planes, dtype, args, seed, tries=None, probe_args=None):
    """`host_planes` (a list of 2-D arrays, one per plane of the batch) in a source arena, a destination arena of the
    same geometry, and the BoxBlur launch `args` on them. Both arenas are plain vszip_dev_alloc requests: what a caller gets.
    (Rounds 2-3 searched for a fast placement here; since round 4 the allocator does, for every caller - include/vszip_hip.h.)
    Returns (step, keep, info)."""
    shapes = [p.shape for p in host_planes]
    isz = np.dtype(dtype).itemsize
    t_alloc = time.perf_counter()
    src = Arena(dev, shapes, dtype, seed + 1)
    dst = Arena(dev, shapes, dtype, seed + 2)
    t_alloc = time.perf_counter() - t_alloc
    for a, d in zip(host_planes, src.planes):
        a = np.ascontiguousarray(a)
        dev.check(dev.lib.vszip_copy_h2d_2d(dev.ctx, d.ptr, d.stride * isz, a.ctypes.data, a.strides[0], a.shape[1] * isz, a.shape[0]))
    dev.sync()
    table = dev.plane_table(src.planes, dst.planes)
    info = {"tries": 1, "alloc_seconds": round(t_alloc, 3), "arena": dev.arena_info(dst.ptr)}
    return (lambda: dev.boxblur_table(dtype, table, *args)), (src, dst, info), info


def setup_boxblur(dev, rank, frames, radius, tries=None):
    """The headline batch: `frames` 4K YUV420P16 frames (distinct content per frame: noise rolled by the frame index),
    placement probed (placed_batch). The probe launches use the neighbouring radius: the same access shape, but another
    instance of the kernel template, so that the rocprofv3 statistics of the measured kernel
    (boxblur_ct_ring_kernel<u16, 13>) hold the timed launches only."""
    base = make_frame(rank, W4K, H4K)
    planes = [np.roll(plane, f * 17 + 1, axis=1) for f in range(frames) for plane in base]
    probe_r = radius - 1 if radius > 1 else radius + 1
    step, keep, _ = placed_batch(dev, planes, np.uint16, (radius, 1, radius, 1), 1000 * rank, tries, (probe_r, 1, probe_r, 1))
    return step, keep


def setup_bilateral(dev, w, h, frames, sigma_s=2, sigma_r=2, eight_bit=False):
    base = natural_frame(w, h)
    if eight_bit:  # YUV420P8: the picture's own 8-bit samples
        base = [(p >> 8).astype(np.uint8) for p in base]
    cfg = dev.bilateral_cfg([sigma_s], [sigma_r], yuv=True, ssw=1, ssh=1, hist_len=256 if eight_bit else 65536)
    srcs, dsts, idx = [], [], []
    for f in range(frames):
        for i, plane in enumerate(base):
            srcs.append(dev.upload(np.roll(plane, f * 13, axis=1)))
            dsts.append(dev.empty(plane.shape[0], plane.shape[1], plane.dtype))
            idx.append(i)
    keep = (srcs, dsts, cfg)
    if eight_bit:
        return (lambda: dev.bilateral(srcs, dsts, cfg, idx, peak=255.0)), keep
    return (lambda: dev.bilateral(srcs, dsts, cfg, idx)), keep


def setup_ssimulacra2_rgb24(dev, w, h, pairs):
    """The same comparison from the clips' own 8-bit RGB planes: the colour pre-stage (hz.toRGBS + sRGBtoLinearRGB)
    runs fused into the first SSIMULACRA2 pass, a pair is 50 MB instead of 199 MB."""
    import fixtures as fx

    rng = np.random.default_rng(1)
    ref = [fx.tiled_natural((h, w), np.uint8, p) for p in range(3)]
    dis = [np.clip(p.astype(np.int16) + rng.integers(-5, 6, p.shape, dtype=np.int16), 0, 255).astype(np.uint8) for p in ref]
    fmt = dev.ssim_source("RGB", np.uint8, 8)
    r, d = [], []
    for p in range(pairs):
        r += [dev.upload(np.roll(x, p * 7, axis=1)) for x in ref]
        d += [dev.upload(np.roll(x, p * 7, axis=1)) for x in dis]
    return (lambda: dev.ssimulacra2_src(fmt, r, d)), (r, d)


def yuv420p8_pair(w, h):
    """A 4:2:0 8-bit pair: the reference's own YUV420P8 fixture (tests/conftest.py:88-102, restated in oracle/vs_host.py) tiled
    to w x h; dist = ref + small integer noise on every plane."""
    import fixtures as fx

    rng = np.random.default_rng(2)
    y, u, v = fx.crop_yuv(8)
    tile = lambda p, hh, ww: np.ascontiguousarray(np.tile(p, (-(-hh // p.shape[0]), -(-ww // p.shape[1])))[:hh, :ww])
    ref = [tile(y, h, w), tile(u, h // 2, w // 2), tile(v, h // 2, w // 2)]
    dis = [np.clip(p.astype(np.int16) + rng.integers(-4, 5, p.shape, dtype=np.int16), 0, 255).astype(np.uint8) for p in ref]
    return ref, dis


def setup_ssimulacra2_yuv420p8(dev, w, h, pairs):
    """SSIMULACRA2 straight from YUV420P8 planes (round 3): chroma upsampling (zimg's Catmull-Rom), the YUV -> RGB matrix and the
    transfer table run fused into the first pass; a pair is 25 MB instead of 199 MB of linear RGBS."""
    ref, dis = yuv420p8_pair(w, h)
    fmt = dev.ssim_source("YUV", np.uint8, 8, ssw=1, ssh=1, matrix=1, chroma_loc=0)
    r, d = [], []
    for p in range(pairs):
        r += [dev.upload(np.roll(x, p * 8, axis=1)) for x in ref]
        d += [dev.upload(np.roll(x, p * 8, axis=1)) for x in dis]
    return (lambda: dev.ssimulacra2_src(fmt, r, d)), (r, d)


def setup_ssimulacra2_yuv444p8(dev, w, h, pairs):
    """SSIMULACRA2 from YUV444P8 planes (round 6): no resampling, the matrix and the transfer table in a pre-stage pass of its own with the table in LDS."""
    ref, dis = yuv420p8_pair(w, h)
    up2 = lambda p: np.ascontiguousarray(np.repeat(np.repeat(p, 2, axis=0), 2, axis=1)[:h, :w])
    ref = [ref[0], up2(ref[1]), up2(ref[2])]
    dis = [dis[0], up2(dis[1]), up2(dis[2])]
    fmt = dev.ssim_source("YUV", np.uint8, 8, ssw=0, ssh=0, matrix=1, chroma_loc=0)
    r, d = [], []
    for p in range(pairs):
        r += [dev.upload(np.roll(x, p * 8, axis=1)) for x in ref]
        d += [dev.upload(np.roll(x, p * 8, axis=1)) for x in dis]
    return (lambda: dev.ssimulacra2_src(fmt, r, d)), (r, d)


def setup_ssimulacra2(dev, w, h, pairs):
    ref, dis = rgbs_pair(w, h)
    r, d = [], []
    for p in range(pairs):
        r += [dev.upload(np.roll(x, p * 7, axis=1), 1) for x in ref]
        d += [dev.upload(np.roll(x, p * 7, axis=1), 1) for x in dis]
    return (lambda: dev.ssimulacra2(r, d)), (r, d)


W8K, H8K = 7680, 4320


def setup_pipeline(dev, w, h, frames):
    """BASELINE config 5: Bilateral(sigmaS=2, sigmaR=2) -> BoxBlur(r=2) -> SSIMULACRA2(source, processed)
    on RGBS frames; every intermediate plane stays in HBM, only the per-frame scores leave the device
    (the chain tests/test_gpu_fullsize.py checks against the oracle stage by stage)."""
    import fixtures as fx

    k0 = np.float32(0.0404482362771082)  # sRGB -> linear, as the SSIMULACRA2 wrapper's input contract wants
    base = []
    for p in range(3):
        v = np.ascontiguousarray(fx.tiled_natural((h, w), np.float32, p))
        base.append(np.where(v <= k0, v / np.float32(12.92), ((v + np.float32(0.055)) / np.float32(1.055)) ** np.float32(2.4)).astype(np.float32))
    cfg = dev.bilateral_cfg([2], [2], yuv=False, ssw=0, ssh=0, hist_len=65536)
    srcs, mid, outp, idx = [], [], [], []
    for f in range(frames):
        for i, plane in enumerate(base):
            srcs.append(dev.upload(np.roll(plane, f * 19, axis=1), 1))
            mid.append(dev.empty(h, w, np.float32, 1))
            outp.append(dev.empty(h, w, np.float32, 1))
            idx.append(i)
    scores = []

    def step():
        dev.bilateral(srcs, mid, cfg, idx)
        dev.boxblur(mid, outp, 2, 1, 2, 1)
        scores[:] = dev.ssimulacra2(srcs, outp)

    return step, (srcs, mid, outp, cfg), scores


def pipeline_line(dev, timed, world, frames, steps, warmup, max_over_ranks, reduce_scalars):
    step, keep, scores = setup_pipeline(dev, W8K, H8K, frames)
    dt, region_ms, _, _ = timed.run(step, steps, warmup)
    dt = max_over_ranks(dt)
    frame_bytes = 3 * W8K * H8K * 4  # 398 131 200
    # compulsory traffic per frame: source read by Bilateral and by SSIMULACRA2, the two intermediates written and read once
    alg = 6 * frame_bytes
    t0 = time.perf_counter()
    tot = reduce_scalars(np.array([float(np.sum(scores)), float(len(scores))]))
    reduce_ms = (time.perf_counter() - t0) * 1e3
    res = {"value": world * frames * steps / dt, "unit": "frames/s", "ms_per_step": dt * 1e3 / steps, "frames_per_step_per_gpu": frames,
           "stream_ms_per_frame": region_ms / (steps * frames),
           "roofline": {"bound": "hbm", "achieved": alg * frames * steps / (region_ms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                        "frac": alg * frames * steps / (region_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                        "alg_bytes_per_call": alg * frames, "basis": "stream",
                        "kernel": "whole chain (bilateral_lds16<f32> + boxblur_ctf_ring + the SSIMULACRA2 kernels); algorithmic bytes = 6 x frame "
                                  "(source read twice, each intermediate written and read once)"},
           "clip_mean_score": {"value": float(tot[0] / tot[1]), "frames": int(tot[1]), "reduced_over_ranks": world, "allreduce_ms": reduce_ms},
           "workload": "Bilateral(sigmaS=2,sigmaR=2) -> BoxBlur(r=2) -> SSIMULACRA2 on 7680x4320 RGBS, intermediates HBM-resident, per-clip mean score all-reduced"}
    dev.bilateral_free(keep[3])
    return res


def eedi3_leg(dev, timed, no_cpu, frames=16):
    """BASELINE config 4: EEDI3 field=1 dh=1 on 1920x1080 YUV420PS -> 1920x2160 (f32; the reference
    rejects integer input). Latency/compute bound: reported as frames/s and interpolated lines/s."""
    import fixtures as fx

    base = [np.ascontiguousarray(fx.tiled_natural(s, np.float32, p)) for p, s in enumerate(yuv420_shapes(W1080, H1080))]
    srcs = []
    for f in range(frames):
        srcs += [dev.upload(np.roll(p, f * 11, axis=1)) for p in base]
    dsts = dev.eedi3(srcs, 1, dh=True)  # allocates the outputs once

    def step():
        table = dev.plane_table(srcs, dsts)
        prm = _eedi3_params()
        dev.check(dev.lib.vszip_eedi3(dev.ctx, table, None, None, len(srcs), 1, 0, prm))

    dt, _, _, _ = timed.run(step, 5, 1)
    lines = sum(s[0] for s in yuv420_shapes(W1080, H1080))  # one interpolated line per source line (dh)
    fb = 3 * sum(p.nbytes for p in base)  # one read of the frame, one write of the double-height frame
    gbs = fb * frames * 5 / dt / 1e9
    res = {"value": frames * 5 / dt, "unit": "frames/s", "interpolated_lines_per_s": frames * 5 * lines / dt, "frames_per_call": frames,
           "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                        "alg_bytes_per_call": fb * frames, "basis": "wall",
                        "note": "not an HBM-bound filter: 41 directions x a 5-tap window per pixel and a dynamic programme along every line "
                                "(eedi3_line_kernel: VALU about 80 % busy, DESIGN.md 3.5); the fraction is reported for completeness"},
           "workload": "vszip.EEDI3 field=1 dh=1 (defaults: mdis 20, nrad 2, vcheck 2), 1920x1080 YUV420PS -> 1920x2160, HBM-resident"}
    # about 1.7 kflop per interpolated pixel x 3.11 Mpx = 5.4 GFLOP per frame against the 157.3 TFLOPS fp32 vector peak
    tf = 5.4e9 * frames * 5 / dt / 1e12
    res["limit"] = {"bound": "fp32_vector", "achieved": tf, "peak": 157.3, "unit": "TFLOP/s", "frac": tf / 157.3,
                    "note": "5.4 GFLOP per 1080p frame (SURVEY 8d); the Viterbi chain and the vertical-consistency chain are serial, so the fraction is a lower bound on "
                            "how busy the vector units are (eedi3_line_kernel: about 80 % VALU issue utilisation, DESIGN.md 3.5)"}
    if not no_cpu:
        from oracle import oracle as orc

        cores = cpu_threads()

        def one(_):
            for p in base:
                orc.eedi3(p, 1, dh=True)

        t0 = time.perf_counter()
        one(0)
        res["cpu_baseline"] = _timed_pool(one, cores, 4.0, "1920x1080 YUV420PS frame, EEDI3 field=1 dh=1", time.perf_counter() - t0)
    return res


def _eedi3_params():
    import ctypes as C

    from vszip_amd.capi import Eedi3Params

    return C.byref(Eedi3Params(1, 0.2, 0.25, 20.0, 2, 20, 0, 2, 32.0, 64.0, 4.0))


def xpsnr_leg(dev, timed, no_cpu, frames=8, workers=8, batch=64, per_frame=True):
    """XPSNR (getWSSE) on 1920x1080 YUV420P8 with temporal weighting. `value` is the batched entry
    point (vszip_xpsnr_wsse_batch: `batch` frames per launch, one result copy). Per frame, every call
    synchronises (the result is a host scalar), so one caller is latency bound; `per_frame_calls` is
    what VapourSynth's fmParallel gives the plugin: `workers` host threads, each with its own
    context (stream), pulling frames at once."""
    import threading

    import fixtures as fx
    import vszip_amd

    rng = np.random.default_rng(3)
    base = [fx.tiled_natural(s, np.uint8, p) for p, s in enumerate(yuv420_shapes(W1080, H1080))]
    noise = [rng.integers(-3, 4, p.shape).astype(np.int16) for p in base]
    org = [[np.roll(p, 5 * f, axis=1) for p in base] for f in range(batch)]
    rec = [[np.clip(p.astype(np.int16) + np.roll(nz, f, axis=0), 0, 255).astype(np.uint8) for p, nz in zip(fr, noise)] for f, fr in enumerate(org)]
    dorg = [[dev.upload(p) for p in fr] for fr in org]
    drec = [[dev.upload(p) for p in fr] for fr in rec]
    p1s = [dorg[f - 1][0] if f >= 1 else None for f in range(batch)]
    p2s = [dorg[f - 2][0] if f >= 2 else None for f in range(batch)]
    batch_call = dev.xpsnr_batch_call(dorg, drec, p1s, p2s, depth=8, frame_rate=24)  # pointer arrays built once, like a C host's
    dtb, _, dom_ms, launches = timed.run(batch_call, 10, 2)
    batched = batch * 10 / dtb
    fb = 2 * sum(s[0] * s[1] for s in yuv420_shapes(W1080, H1080)) + W1080 * H1080  # org + rec + the previous luma
    if not per_frame:  # (tools/roofline_check.sh: the batch call alone, so that a profile's xpsnr_strip_kernel rows are 64-frame launches only)
        return {"value": batched, "unit": "frames/s", "frames_per_call": batch,
                "roofline": {"bound": "hbm", "achieved": batch * fb / (dom_ms / launches * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                             "frac": batch * fb / (dom_ms / launches * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "kernel": "xpsnr_strip_kernel<u8>",
                             "alg_bytes_per_call": batch * fb, "basis": "dominant kernel", "kernel_match": "xpsnr_strip_kernel", "avg_launch_us": dom_ms / launches * 1e3}}

    def step():
        for f in range(frames):
            dev.xpsnr_wsse(dorg[f], drec[f], dorg[f - 1][0] if f >= 1 else None, dorg[f - 2][0] if f >= 2 else None, depth=8, frame_rate=24)

    step()
    t0 = time.perf_counter()
    for _ in range(5):
        step()
    dt = time.perf_counter() - t0
    single = frames * 5 / dt

    # the same frames from `workers` threads, one context each (device memory is shared)
    devs = [vszip_amd.Device(dev.device) for _ in range(workers)]
    reps = 20

    def worker(d):
        for _ in range(reps):
            for f in range(frames):
                d.xpsnr_wsse(dorg[f], drec[f], dorg[f - 1][0] if f >= 1 else None, dorg[f - 2][0] if f >= 2 else None, depth=8, frame_rate=24)

    for d in devs:  # warm-up (scratch allocation per context)
        d.xpsnr_wsse(dorg[0], drec[0], None, None, depth=8, frame_rate=24)
    threads = [threading.Thread(target=worker, args=(d,)) for d in devs]
    t0 = time.perf_counter()
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    dtw = time.perf_counter() - t0
    multi = workers * reps * frames / dtw
    for d in devs:
        d.close()
    res = {"value": batched, "unit": "frames/s", "frames_per_call": batch, "algorithmic_GBps": batched * fb / 1e9,
           "kernel_us_per_call": (dom_ms / launches * 1e3) if launches else None,
           "kernel_hbm_frac": (batch * fb / (dom_ms / launches * 1e-3) / 8e12) if launches else None,
           "per_frame_calls": {"workers": workers, "frames_per_s": multi, "single_caller_frames_per_s": single},
           "roofline": ({"bound": "hbm", "achieved": batch * fb / (dom_ms / launches * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": batch * fb / (dom_ms / launches * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None, "kernel": "xpsnr_strip_kernel<u8>",
                         "alg_bytes_per_call": batch * fb, "basis": "dominant kernel", "kernel_match": "xpsnr_strip_kernel",
                         "avg_launch_us": dom_ms / launches * 1e3} if launches else None),
           "workload": f"vszip.XPSNR getWSSE, 1920x1080 YUV420P8 org vs rec, temporal, {batch} frames per call; per_frame_calls: one synchronising call per frame, "
                       f"{workers} host threads with a context each"}
    if not no_cpu:
        from oracle import oracle as orc

        cores = cpu_threads()

        def one(_):
            orc.xpsnr_wsse(org[2], rec[2], org[1][0], org[0][0], depth=8, frame_rate=24)

        t0 = time.perf_counter()
        one(0)
        res["cpu_baseline"] = _timed_pool(one, cores, 2.0, "1920x1080 YUV420P8 frame pair, XPSNR", time.perf_counter() - t0)
    return res


def planestats_leg(dev, timed, frames=64, only=None):
    """PlaneAverage / PlaneMinMax on 3840x2160 YUV420P16: single-pass readers, HBM roofline = bytes read once."""
    base = make_frame(7, W4K, H4K)
    planes = []
    for f in range(frames):
        planes += [dev.upload(np.roll(p, f * 3, axis=1)) for p in base]
    fb = sum(2 * s[0] * s[1] for s in yuv420_shapes(W4K, H4K)) * frames
    out = {}
    # (argument blocks built once, as a per-frame caller in C holds them: building a 192-entry ctypes table per call cost the legs 30-40 us of Python)
    for name, fn in (("plane_average_4k", dev.prepared_plane_average(planes, exclude=[-1])),
                     ("plane_minmax_4k", dev.prepared_plane_minmax(planes)),
                     ("plane_minmax_thr_4k", dev.prepared_plane_minmax(planes, 0.1, 0.1))):
        if only and name != only:
            continue
        dt, region_ms, dom_ms, launches = timed.run(fn, 10, 2)
        gbs = fb * 10 / (dom_ms * 1e-3) / 1e9 if dom_ms > 0 else 0.0
        whole = fb * 10 / (region_ms * 1e-3) / 1e9
        out[name] = {"value": frames * 10 / dt, "unit": "frames/s",
                     "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                                  "kernel": "the plane reader (average_kernel / minmax_kernel / hist_kernel pass 0)", "avg_launch_us": dom_ms * 1e3 / max(launches, 1),
                                  "alg_bytes_per_call": fb, "basis": "dominant kernel",
                                  "kernel_match": {"plane_average_4k": "average_kernel", "plane_minmax_4k": "minmax_kernel", "plane_minmax_thr_4k": "hist_sweep_kernel"}[name],
                                  "whole_call": {"note": "all kernels of the call + the scalars' way to the host + the one sync", "achieved": whole, "frac": whole / HBM_PEAK_GBS}},
                     "whole_call_frac": whole / HBM_PEAK_GBS,
                     "workload": f"{name}: {frames} x 3840x2160 YUV420P16 per call ({3 * frames} planes in launches of up to 192, one sync), HBM-resident; value includes the sync"}
        if name == "plane_minmax_thr_4k":
            # round 6: the timed calls are a clip's steady state - every call after the first sweeps its planes once over the ranges the previous call's answers
            # predict; the same call with the prediction off (what a first call or a scene cut pays: the two sweeps) beside it
            with dev.options(VSZIP_MINMAX_NO_PREDICT=1):
                dt2, _, _, _ = timed.run(fn, 6, 1)
            out[name]["two_sweeps_value"] = frames * 6 / dt2
            out[name]["workload"] += "; steady state of a clip (ranges predicted from the previous call); two_sweeps_value: VSZIP_MINMAX_NO_PREDICT=1"
    return out


def limiter_leg(dev, timed, frames=64):
    """vszip.Limiter (tv_range bounds) on 3840x2160 YUV420P16: a streaming read + write, HBM roofline = both."""
    base = make_frame(9, W4K, H4K)
    srcs, dsts = [], []
    for f in range(frames):
        for p in base:
            srcs.append(dev.upload(np.roll(p, f * 5, axis=1)))
            dsts.append(dev.empty(p.shape[0], p.shape[1], p.dtype))
    lo, hi = [4096.0] * len(srcs), [60160.0, 61440.0, 61440.0] * frames
    dt, _, dom_ms, launches = timed.run(dev.prepared_limiter(srcs, dsts, lo, hi), 10, 2)  # (queued like the BoxBlur steps: argument blocks built once, one sync at the end)
    fb = 2 * sum(2 * s[0] * s[1] for s in yuv420_shapes(W4K, H4K)) * frames
    gbs = fb * 10 / (dom_ms * 1e-3) / 1e9
    return {"limiter_4k": {"value": frames * 10 / dt, "unit": "frames/s",
                           "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                                        "kernel": "limiter_kernel<u16>", "avg_launch_us": dom_ms * 1e3 / launches,
                                        "alg_bytes_per_call": fb, "basis": "dominant kernel", "kernel_match": "limiter_kernel"},
                           "workload": f"vszip.Limiter tv_range: {frames} x 3840x2160 YUV420P16 per call, HBM-resident"}}


def limit_filter_leg(dev, timed, frames=64):
    """vszip.LimitFilter(flt, src) on 3840x2160 YUV420P16: two streams in, one out."""
    base = make_frame(11, W4K, H4K)
    flts, srcs, dsts = [], [], []
    for f in range(frames):
        for p in base:
            srcs.append(dev.upload(np.roll(p, f * 5, axis=1)))
            flts.append(dev.upload(np.roll(p, f * 5 + 1, axis=1)))
            dsts.append(dev.empty(p.shape[0], p.shape[1], p.dtype))
    n = len(srcs)
    dt, _, dom_ms, launches = timed.run(dev.prepared_limit_filter(flts, srcs, dsts, [2056.0] * n, [2056.0] * n, [3.0] * n), 10, 2)
    fb = 3 * sum(2 * s[0] * s[1] for s in yuv420_shapes(W4K, H4K)) * frames
    gbs = fb * 10 / (dom_ms * 1e-3) / 1e9
    return {"limit_filter_4k": {"value": frames * 10 / dt, "unit": "frames/s",
                                "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                                             "kernel": "limit_filter_kernel<u16>", "avg_launch_us": dom_ms * 1e3 / launches,
                                             "alg_bytes_per_call": fb, "basis": "dominant kernel", "kernel_match": "limit_filter_kernel"},
                                "workload": f"vszip.LimitFilter dark_thr=bright_thr=8 elast=3: {frames} x 3840x2160 YUV420P16 per call, HBM-resident"}}


def boxblur_other_paths_leg(dev, timed, frames_in=8, only=None):
    """The BoxBlur paths beside the headline one: the runtime path (radius > 22 or several passes,
    boxblur_runtime.zig) on 4K YUV420P16 and the compile-time float path on 4K YUV420PS."""
    out = {}
    base16 = make_frame(3, W4K, H4K)
    basef = [(p.astype(np.float32) / 65535.0) for p in base16]
    fmt_names = {'uint16': 'YUV420P16', 'float32': 'YUV420PS', 'uint8': 'YUV420P8'}
    base8 = [(p >> 8).astype(np.uint8) for p in base16]
    for name, base, args, dt_ in (("boxblur_rt_r30_4k", base16, (30, 1, 30, 1), np.uint16), ("boxblur_rt_r5x3_4k", base16, (5, 3, 5, 3), np.uint16),
                                  ("boxblur_ct_float_r13_4k", basef, (13, 1, 13, 1), np.float32), ("boxblur_rt_float_r5x3_4k", basef, (5, 3, 5, 3), np.float32),
                                  ("boxblur_ct_u8_r13_4k", base8, (13, 1, 13, 1), np.uint8)):
        if only and name != only:
            continue
        srcs, dsts = [], []
        frames = 64 if (dt_ == np.uint8 or name == "boxblur_rt_float_r5x3_4k") else frames_in  # (the float chain is bound by the parallelism a call offers: quoted at 64 frames like the headline)
        placement = None
        if name.startswith("boxblur_ct_"):  # the ring kernels: placement probed like the headline's
            step, keep_, placement = placed_batch(dev, [np.roll(p, f + 1, axis=1) for f in range(frames) for p in base], dt_, args, 8000)
        else:
            for f in range(frames):
                for p in base:
                    srcs.append(dev.upload(np.roll(p, f + 1, axis=1)))
                    dsts.append(dev.empty(p.shape[0], p.shape[1], p.dtype))
            table = dev.plane_table(srcs, dsts)
            step = lambda: dev.boxblur_table(dt_, table, *args)
        dt, kms, _, _ = timed.run(step, 5, 1)
        fb = 2 * sum(p.nbytes for p in base) * frames
        gbs = fb * 5 / (kms * 1e-3) / 1e9
        out[name] = {"value": frames * 5 / dt, "unit": "frames/s",
                     "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                                  "alg_bytes_per_call": fb, "basis": "stream",
                                  "note": "algorithmic bytes = one read + one write of the frame, whatever the number of passes"},
                     "workload": f"vszip.BoxBlur hradius={args[0]} hpasses={args[1]} vradius={args[2]} vpasses={args[3]}, 3840x2160 {fmt_names[np.dtype(dt_).name]}, {frames} frames per call, HBM-resident"}
        if placement:
            out[name]["placement"] = placement
            del keep_
        del srcs, dsts
    return out


def frames_per_call_sweep(dev, timed):
    """Frames per call 1 / 4 / 16 / 64 of the headline launch and of the float pass chain: every HBM-resident figure of
    this file is quoted at 64 frames (192 planes) a launch, a shape a batch API reaches and a one-frame-per-getFrame host does not — the sweep
    says what the same kernels give a caller that passes fewer frames. Plain allocations (no placement search); frames/s per entry."""
    out = {}
    base16 = make_frame(5, W4K, H4K)
    basef = [(p.astype(np.float32) / 65535.0) for p in base16]
    with dev.options(VSZIP_PLACEMENT=0):
        for name, base, args, dt_ in (("boxblur_r13_4k_u16", base16, (RADIUS, 1, RADIUS, 1), np.uint16), ("boxblur_rt_float_r5x3_4k", basef, (5, 3, 5, 3), np.float32)):
            srcs = [dev.upload(np.roll(p, f + 1, axis=1)) for f in range(64) for p in base]
            dsts = [dev.empty(p.shape[0], p.shape[1], p.dtype) for f in range(64) for p in base]
            vals = []
            for n in (1, 4, 16, 64):
                table = dev.plane_table(srcs[: 3 * n], dsts[: 3 * n])
                step = lambda: dev.boxblur_table(dt_, table, *args)
                steps = max(3, (200 if dt_ == np.uint16 else 24) // n)
                dt, _, _, _ = timed.run(step, steps, 2)
                vals.append(n * steps / dt)
            out[name] = vals
            del srcs, dsts
    out["frames_per_call"] = [1, 4, 16, 64]
    return out


def pcie_boxblur(vszip_amd, device_index: int, radius: int, nctx: int = 4, rounds: int = 12, barrier=None):
    """PCIe-inclusive BoxBlur rate, the path a VapourSynth host pays: every frame is copied from
    pinned host memory to the GPU, blurred, and copied back. `nctx` contexts (one stream each,
    like the plugin's per-worker contexts) keep H2D, kernels and D2H of different frames in flight."""
    ctxs = []
    for c in range(nctx):
        d = vszip_amd.Device(device_index)
        frame = make_frame(c, W4K, H4K)
        hin = [d.pinned_array(p.shape, p.dtype) for p in frame]
        hout = [d.pinned_array(p.shape, p.dtype) for p in frame]
        for a, p in zip(hin, frame):
            a[...] = p
        dsrc = [d.empty(p.shape[0], p.shape[1], p.dtype) for p in frame]
        ddst = [d.empty(p.shape[0], p.shape[1], p.dtype) for p in frame]
        ctxs.append((d, hin, hout, dsrc, ddst, d.plane_table(dsrc, ddst)))

    def one_round():
        # a context waits for ITS previous frame only (like a worker thread that owns it), so the other contexts'
        # copies and kernels stay in flight across rounds; the clock stops after every context has drained
        for d, hin, hout, dsrc, ddst, table in ctxs:
            d.sync()
            for a, dp in zip(hin, dsrc):
                d.copy_in(dp, a)
            d.boxblur_table(np.uint16, table, radius, 1, radius, 1)
            for a, dp in zip(hout, ddst):
                d.copy_out(a, dp)

    def drain():
        for c in ctxs:
            c[0].sync()

    one_round()
    drain()
    if barrier:
        barrier()  # every rank starts its timed rounds together: the host links are shared
    t0 = time.perf_counter()
    for _ in range(rounds):
        one_round()
    drain()
    dt = time.perf_counter() - t0
    fps = nctx * rounds / dt
    fb = sum(2 * s[0] * s[1] for s in yuv420_shapes(W4K, H4K))
    res = {"value": fps, "unit": "frames/s", "host_to_host": True, "contexts": nctx, "rounds": rounds, "seconds": dt,
           "pcie_GBps_each_direction": fps * fb / 1e9,
           "workload": "vszip.BoxBlur r=13 3840x2160 YUV420P16, pinned host frame -> GPU -> pinned host frame (never the headline value)"}
    for c in ctxs:
        c[0].close()
    return res


def plugin_legs():
    """Through libvszip.so under the VapourSynth-free test host (tests/fakevs): worker threads call getFrame like
    fmParallel, frames live in ordinary host memory — what a .vpy user sees. Two legs that this round changed:
    SSIMULACRA2 from RGB24 clips (colour pre-stage on the device: 50 MB per pair over the link instead of 199 MB;
    the output clip is converted by the host, outside the clock here: the test host converts eagerly) and BASELINE
    config 5 written as a script — three filter instances fused into one getFrame per frame."""
    import fixtures as fx
    from fakevs import fakevs as vs

    out = {}
    vs.lib().fakevs_set_pool_refill(0)
    vs.core_standins(True)
    try:
        # round 6: the headline filter as a script runs it — every frame up and down the host link (24.9 MB each way; config.pcie_fed_fps is the same
        # traffic from hipHostMalloc'd buffers through the C ABI, the bound this leg can reach)
        y16 = make_frame(0, W4K, H4K)
        clip = vs.source([[np.roll(p, 7 * f, axis=1) for p in y16] for f in range(8)], vs.YUV420P16).vszip.BoxBlur(hradius=RADIUS, vradius=RADIUS)
        clip.pull(32, 16)
        sec = clip.pull(256, 16, warm_per_thread=3)
        out["plugin_boxblur_4k"] = {"value": 256 / sec, "unit": "frames/s", "threads": 16, "host_link_GBps": 256 * 2 * 24.8832e-3 / sec,
                                    "workload": "libvszip.so: vszip.BoxBlur(hradius=vradius=13) on 3840x2160 YUV420P16 frames in host memory, 16 worker threads; "
                                                "host_link_GBps counts both directions"}
        del clip, y16
        base = [fx.tiled_natural((H4K, W4K), np.uint8, p) for p in range(3)]
        rng = np.random.default_rng(1)
        noise = rng.integers(-3, 4, (H4K, W4K), dtype=np.int16)
        ref = [[np.roll(p, 9 * f, axis=1) for p in base] for f in range(4)]
        dis = [[np.clip(p.astype(np.int16) + noise, 0, 255).astype(np.uint8) for p in fr] for fr in ref]
        clip = vs.source(ref, vs.RGB24).vszip.SSIMULACRA2(vs.source(dis, vs.RGB24))
        clip.pull(16, 8)
        sec = clip.pull(128, 16, warm_per_thread=2)
        out["plugin_ssimulacra2_4k_rgb24"] = {"value": 128 / sec, "unit": "pairs/s", "threads": 16, "host_link_GBps": 128 * 49.8e-3 / sec,
                                              "workload": "libvszip.so: vszip.SSIMULACRA2 on 3840x2160 RGB24 clips in host memory, 16 worker threads"}
        del clip, ref, dis
        # round 3: the same from YUV420P8 clips — the format people feed SSIMULACRA2: 12.4 MB per frame over the link
        yref, ydis = yuv420p8_pair(W4K, H4K)
        props = {"_Matrix": 1, "_ColorRange": 1, "_ChromaLocation": 0}
        ra = vs.source([[np.roll(p, 8 * f, axis=1) for p in yref] for f in range(4)], vs.YUV420P8, props=props)
        rb = vs.source([[np.roll(p, 8 * f, axis=1) for p in ydis] for f in range(4)], vs.YUV420P8, props=props)
        clip = ra.vszip.SSIMULACRA2(rb)
        clip.pull(16, 8)
        sec = clip.pull(192, 16, warm_per_thread=2)
        out["plugin_ssimulacra2_4k_yuv420p8"] = {"value": 192 / sec, "unit": "pairs/s", "threads": 16, "host_link_GBps": 192 * 24.9e-3 / sec,
                                                 "workload": "libvszip.so: vszip.SSIMULACRA2 on 3840x2160 YUV420P8 clips in host memory (colour pre-stage on the device; the output "
                                                             "clip is the host-converted reference, converted eagerly by the test host outside the clock), 16 worker threads"}
        del clip, ra, rb, yref, ydis
        # round 6 (late): BASELINE config 4 as a script. Half of a dh frame's lines are the source's own and never cross the link (the plugin copies them on the host):
        # 12.4 MB up + 12.4 MB of interpolated lines down a frame
        e3 = [np.ascontiguousarray(fx.tiled_natural(s, np.float32, p)) for p, s in enumerate(yuv420_shapes(W1080, H1080))]
        clip = vs.source([[np.roll(p, 5 * f, axis=1) for p in e3] for f in range(8)], vs.YUV420PS).vszip.EEDI3(field=1, dh=True)
        clip.pull(32, 16)
        sec = clip.pull(256, 16, warm_per_thread=3)
        out["plugin_eedi3_1080p"] = {"value": 256 / sec, "unit": "frames/s", "threads": 16, "host_link_GBps": 256 * 2 * 12.4416e-3 / sec,
                                     "workload": "libvszip.so: vszip.EEDI3(field=1, dh=True) on 1920x1080 YUV420PS frames in host memory, 16 worker threads; the kept field's lines "
                                                 "are copied on the host, the interpolated lines come down the link; host_link_GBps counts both directions"}
        del clip, e3
        b8 = [np.ascontiguousarray(fx.tiled_natural((H8K, W8K), np.float32, p)) for p in range(3)]
        src = vs.source([[np.roll(p, 19 * f, axis=1) for p in b8] for f in range(2)], vs.RGBS, props={"_Transfer": 8})
        f0, s0 = vs.fusion_stats()
        clip = src.vszip.SSIMULACRA2(src.vszip.Bilateral(sigmaS=2.0, sigmaR=2.0).vszip.BoxBlur(hradius=2, vradius=2))
        clip.pull(4, 4)
        sec = clip.pull(24, 8, warm_per_thread=1)
        f1, s1 = vs.fusion_stats()
        out["plugin_pipeline_8k_rgbs"] = {"value": 24 / sec, "unit": "frames/s", "threads": 8, "host_link_GBps": 24 * 398.1e-3 / sec,
                                          "fused_getframes": f1 - f0, "fused_stages": s1 - s0,
                                          "workload": "libvszip.so: src.vszip.SSIMULACRA2(src.vszip.Bilateral(2,2).vszip.BoxBlur(2,2)) on 7680x4320 RGBS in host memory: "
                                                      "one upload per frame, the upstream instances' kernels run inside SSIMULACRA2's getFrame"}
    finally:
        vs.core_standins(False)
    return out


def boxblur_1080p_leg(dev, timed, no_cpu, frames=64):
    """BASELINE configs[0] — the reference README's own benchmark (README.md:34-44: BlankClip 1920x1080
    YUV420P16, BoxBlur hradius=vradius=13, 1046 fps on an unstated CPU) — on HBM-resident frames, 64 per call."""
    planes = [np.zeros(sh, np.uint16) for _ in range(frames) for sh in yuv420_shapes(W1080, H1080)]
    step, keep, placement = placed_batch(dev, planes, np.uint16, (RADIUS, 1, RADIUS, 1), 7000)
    dt, _, dom_ms, launches = timed.run(step, 200, 5)
    fb = 2 * sum(2 * sh[0] * sh[1] for sh in yuv420_shapes(W1080, H1080)) * frames
    gbs = fb * launches / (dom_ms * 1e-3) / 1e9
    res = {"value": frames * 200 / dt, "unit": "frames/s", "readme_reference_fps": 1046.11,
           "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                        "kernel": "boxblur_ct_ring_kernel<u16,13>", "avg_launch_us": dom_ms * 1e3 / launches,
                        "alg_bytes_per_call": fb, "basis": "dominant kernel", "kernel_match": "boxblur_ct_ring_kernel"},
           "placement": placement,
           "workload": f"vszip.BoxBlur hradius=vradius=13 on 1920x1080 YUV420P16 BlankClip (README bench), {frames} frames per call, HBM-resident"}
    del keep
    if not no_cpu:
        res["cpu_baseline"] = cpu_boxblur(4.0, W1080, H1080, blank=True)
    return res


def boxblur_gauss_leg(dev, timed, frames=64):
    """How scripts approximate a Gaussian: BoxBlur(hradius=1, hpasses=2, vradius=1, vpasses=2) on 1920x1080 YUV420P8 (natural content) — the
    runtime path with its small-radius kernels (all horizontal passes in one launch, both vertical ones in one launch)."""
    import fixtures as fx

    base = [fx.tiled_natural(sh, np.uint8, p) for p, sh in enumerate(yuv420_shapes(W1080, H1080))]
    srcs = [dev.upload(np.roll(p, f * 3, axis=1)) for f in range(frames) for p in base]
    dsts = [dev.empty(p.shape[0], p.shape[1], p.dtype) for f in range(frames) for p in base]
    table = dev.plane_table(srcs, dsts)
    step = lambda: dev.boxblur_table(np.uint8, table, 1, 2, 1, 2)
    dt, kms, _, _ = timed.run(step, 10, 2)
    fb = 2 * sum(p.nbytes for p in base) * frames
    gbs = fb * 10 / (kms * 1e-3) / 1e9
    return {"value": frames * 10 / dt, "unit": "frames/s",
            "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                         "kernel": "boxblur_rt_hsmall_kernel<u8,1> + boxblur_rt_ichain_kernel<u8,2,1|2> (the vertical passes in bands)", "alg_bytes_per_call": fb, "basis": "stream",
                         "note": "algorithmic bytes = one read + one write of the frame"},
            "workload": f"vszip.BoxBlur hradius=1 hpasses=2 vradius=1 vpasses=2 on 1920x1080 YUV420P8 (natural content tiled), {frames} frames per call, HBM-resident"}


def boxblur_1080p_5pass_leg(dev, timed, no_cpu, frames=32):
    """The reference README's third benchmark (README.md:46-49): BoxBlur(hradius=13, hpasses=5, vradius=13, vpasses=5) on a
    1920x1080 YUV420P16 BlankClip, 367.01 fps there (unstated CPU) — the runtime multi-pass path (src/filters/boxblur_runtime.zig
    :81-119). Algorithmic bytes = one read + one write of the frame, whatever the number of passes."""
    srcs, dsts = [], []
    for _ in range(frames):
        for sh in yuv420_shapes(W1080, H1080):
            srcs.append(dev.upload(np.zeros(sh, np.uint16)))
            dsts.append(dev.empty(sh[0], sh[1], np.uint16))
    table = dev.plane_table(srcs, dsts)
    step = lambda: dev.boxblur_table(np.uint16, table, RADIUS, 5, RADIUS, 5)
    dt, kms, _, _ = timed.run(step, 20, 2)
    fb = 2 * sum(2 * sh[0] * sh[1] for sh in yuv420_shapes(W1080, H1080)) * frames
    gbs = fb * 20 / (kms * 1e-3) / 1e9
    res = {"value": frames * 20 / dt, "unit": "frames/s", "readme_reference_fps": 367.01,
           "roofline": {"bound": "hbm", "achieved": gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gbs / HBM_PEAK_GBS, "traffic": None,
                        "kernel": "boxblur_rt_hsmall_kernel<u16,13> (5 horizontal passes) + boxblur_rt_ichain_kernel<u16,5,1|2> (5 vertical passes in bands)",
                        "alg_bytes_per_call": fb, "basis": "stream",
                        "note": "algorithmic bytes = one read + one write of the frame; every pass that goes through HBM divides the fraction"},
           "workload": f"vszip.BoxBlur hradius=13 hpasses=5 vradius=13 vpasses=5 on 1920x1080 YUV420P16 BlankClip (README bench 3), {frames} frames per call, HBM-resident"}
    if not no_cpu:
        from oracle import oracle as orc

        cores = cpu_threads()
        frame = [np.zeros(sh, np.uint16) for sh in yuv420_shapes(W1080, H1080)]

        def one(_):
            for p in frame:
                orc.boxblur(p, RADIUS, 5, RADIUS, 5)

        t0 = time.perf_counter()
        one(0)
        res["cpu_baseline"] = _timed_pool(one, cores, 3.0, "1920x1080 YUV420P16 frame (BlankClip), BoxBlur r=13 x 5+5 passes", time.perf_counter() - t0)
    return res


def pcie_path(local_rank: int):
    """The GPU's PCIe endpoint and every bridge above it with the link each negotiated (sysfs): the PCIe-fed rate is 1.10 k
    4K-YUV420P16 frames/s on some boxes of the pool and 1.8 k on others, whatever the process does (tools/pcie_ab.py)."""
    try:
        import torch

        pr = torch.cuda.get_device_properties(local_rank)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        dev = Path(f"/sys/bus/pci/devices/{bdf}").resolve()
        out = []
        while dev.name != "" and (dev / "current_link_speed").exists() and len(out) < 8:
            rd = lambda n: (dev / n).read_text().strip() if (dev / n).exists() else None
            out.append({"bdf": dev.name, "speed": rd("current_link_speed"), "width": rd("current_link_width"), "max_speed": rd("max_link_speed"), "max_width": rd("max_link_width")})
            dev = dev.parent
        return out
    except Exception as e:
        return [{"error": str(e)}]


def bind_to_gpu_numa(local_rank: int):
    """Best effort: run this rank's host threads (and first-touch its pinned buffers) on the NUMA node
    the GPU hangs off — the PCIe-fed rate is limited by host memory placement."""
    try:
        import torch

        pr = torch.cuda.get_device_properties(local_rank)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        node = int(Path(f"/sys/bus/pci/devices/{bdf}/numa_node").read_text())
        if node < 0:
            return None
        cpus = []
        for part in Path(f"/sys/devices/system/node/node{node}/cpulist").read_text().strip().split(","):
            lo, _, hi = part.partition("-")
            cpus += list(range(int(lo), int(hi or lo) + 1))
        os.sched_setaffinity(0, cpus)
        return node
    except Exception:
        return None


def ranks_max(dt: float, use_dist: bool, coll_dev=None) -> float:
    """the contract's timing rule: every rank's wall time of the timed region, MAX-reduced (RCCL on the GPUs, gloo in the CPU tests)"""
    if not use_dist:
        return dt
    import torch
    import torch.distributed as dist

    t = torch.tensor([dt], dtype=torch.float64, device=coll_dev or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def whole_job_value(units_per_rank_per_step: int, steps: int, world: int, max_dt: float) -> float:
    """`value` at N ranks: the units ALL ranks processed (weak scaling: the same work on every rank) divided by the slowest rank's time"""
    return world * units_per_rank_per_step * steps / max_dt


def gather_rank_records(rec: dict, use_dist: bool) -> list:
    """every rank's small record (device, NUMA node, arena, launch time) on rank 0, for the sidecar"""
    if not use_dist:
        return [rec]
    import torch.distributed as dist

    out = [None] * dist.get_world_size()
    dist.all_gather_object(out, rec)
    return out


def xpsnr_clip_leg(dev, vszip_amd, rank, world, coll_dev, frames_per_rank=8):
    """The one real exchange of the design: XPSNR's per-clip accumulators {sum_wdist[3], sum_xpsnr[3], n}
    (src/vapoursynth/xpsnr.zig:89-96) from actual vszip_xpsnr_wsse_batch output on each rank's frames
    (frame n belongs to rank n mod world), SUM-all-reduced over RCCL, averaged with getAvgXPSNR
    (src/filters/xpsnr.zig:359-368) and checked on rank 0 against a single-rank pass over the whole clip."""
    import ctypes as C

    import fixtures as fx
    from vszip_amd import cluster

    n_total = world * frames_per_rank
    shapes = yuv420_shapes(W1080, H1080)
    base = [fx.tiled_natural(sh, np.uint8, p) for p, sh in enumerate(shapes)]

    def frame(n):
        org = [np.roll(b, 5 * n, axis=1) for b in base]
        rng = np.random.default_rng(1000 + n)
        rec = [np.clip(o.astype(np.int16) + rng.integers(-3, 4, o.shape, dtype=np.int16), 0, 255).astype(np.uint8) for o in org]
        return org, rec

    def accumulate(frame_ids):
        orgs, recs, p1 = [], [], []
        for n in frame_ids:
            o, r = frame(n)
            orgs.append([dev.upload(x) for x in o])
            recs.append([dev.upload(x) for x in r])
            p1.append(dev.upload(frame(n - 1)[0][0]) if n > 0 else None)
        wsse = dev.xpsnr_wsse_batch(orgs, recs, p1, None, depth=8, frame_rate=24, temporal=True)
        acc = np.zeros(7)
        for wf in wsse:
            for c in range(3):
                acc[c] += np.sqrt(float(wf[c]))
                acc[3 + c] += dev.lib.vszip_xpsnr_value(C.c_uint64(wf[c]), shapes[c][1], shapes[c][0], 8)
            acc[6] += 1
        return acc

    mine = list(cluster.frames_of_rank(n_total, rank, world))
    t0 = time.perf_counter()
    tot = cluster.allreduce_clip_scalars(accumulate(mine), device=coll_dev)
    ms = (time.perf_counter() - t0) * 1e3
    avg = cluster.xpsnr_clip_average(tot[0:3], tot[3:6], tot[6], [sh[1] for sh in shapes], [sh[0] for sh in shapes], 8, dev.lib)
    res = {"avg_xpsnr_yuv": [float(v) for v in avg], "frames": int(tot[6]), "reduced_over_ranks": world, "ms": ms}
    if rank == 0:
        serial = accumulate(range(n_total)) if world > 1 else tot
        res["max_rel_diff_vs_single_rank"] = float(np.max(np.abs(tot - serial) / np.maximum(np.abs(serial), 1e-300)))
        res["matches_single_rank"] = bool(res["max_rel_diff_vs_single_rank"] <= 1e-12)
    return res


# ---------------------------------------------------------------------------
# The ONE stdout line. The driver keeps the last 8 KB of stdout and parses the JSON from it, so the line
# must fit that window whole (round 3's 23 KB line left the driver with `parsed: null`). The line carries the
# contract keys, `config` (workload + scalars), `roofline`, `cpu_baseline` and per `others` leg value / unit /
# frac (+ limit); every other field goes to the sidecar file named in the line's `detail`.
# ---------------------------------------------------------------------------
LINE_MAX_BYTES = 7600
CONTRACT_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "rccl_ranks", "collective_backend")
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "kernel", "avg_launch_us", "launches", "algorithmic_bytes_per_launch", "frac_median", "frac_rocprof", "rocprof_avg_launch_us", "rocprof_source")


def _sig(v, digits=6):
    """floats to 6 significant digits (no NaN / Infinity tokens: they are not JSON)"""
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v
    if isinstance(v, float):
        if v != v or v in (float("inf"), float("-inf")):
            return None
        if v == int(v) and abs(v) < 2.0 ** 53:
            return int(v) if abs(v) >= 1e6 else v  # byte counts stay exact
        return float(f"{v:.{digits}g}")
    if isinstance(v, dict):
        return {k: _sig(x, digits) for k, x in v.items()}
    if isinstance(v, (list, tuple)):
        return [_sig(x, digits) for x in v]
    if isinstance(v, (np.integer,)):
        return int(v)
    if isinstance(v, (np.floating,)):
        return _sig(float(v), digits)
    return str(v)


def _scalar(v):
    return v is None or isinstance(v, (bool, int, float, str))


def compact_leg(leg: dict) -> dict:
    """one `others` leg on the line: value, unit, frac (of its roofline), limit.bound / limit.frac, error"""
    if not isinstance(leg, dict):
        return {"error": str(leg)[:120]}
    if "error" in leg:
        return {"error": str(leg["error"])[:120]}
    o = {k: leg[k] for k in ("value", "unit") if k in leg}
    rf = leg.get("roofline")
    if isinstance(rf, dict) and isinstance(rf.get("frac"), (int, float)):
        o["frac"] = rf["frac"]
    if isinstance(leg.get("whole_call_frac"), (int, float)):
        o["whole_call_frac"] = leg["whole_call_frac"]
    lim = leg.get("limit")
    if isinstance(lim, dict) and "bound" in lim and "frac" in lim:
        o["limit"] = {"bound": lim["bound"], "frac": lim["frac"]}
    for k in ("threads", "host_link_GBps"):  # the plugin legs: worker threads and what crossed the host link
        if isinstance(leg.get(k), (int, float)):
            o[k] = leg[k]
    cpu = leg.get("cpu_baseline")  # the CPU side of the same workload, timed in this run (same unit as `value`)
    if isinstance(cpu, dict) and isinstance(cpu.get("value"), (int, float)):
        o["cpu"] = {"value": cpu["value"], "cores": cpu.get("cores")}
        if isinstance(o.get("value"), (int, float)) and cpu["value"] > 0:
            o["cpu"]["x"] = o["value"] / cpu["value"]
        for src, dst in (("single_thread_value", "single"), ("scaling", "scaling")):  # how the threads scaled
            if isinstance(cpu.get(src), (int, float)):
                o["cpu"][dst] = cpu[src]
    return o


def compact_line(full: dict, detail_name: str | None = None) -> dict:
    """The dict that is printed: see the section comment. `full` is untouched (it is what the sidecar holds)."""
    line = {k: full[k] for k in CONTRACT_KEYS if k in full}
    cfg = {}
    for k, v in full.get("config", {}).items():
        if _scalar(v):
            cfg[k] = v if not isinstance(v, str) or len(v) <= 200 else v[:200]
        elif isinstance(v, dict):  # small all-scalar records stay (clip sums); anything with prose or nesting is detail
            flat = {a: b for a, b in v.items() if _scalar(b) and not (isinstance(b, str) and len(b) > 60)}
            lists = {a: b for a, b in v.items() if isinstance(b, (list, tuple)) and len(b) <= 4 and all(_scalar(x) for x in b)}
            flat.update(lists)
            if flat and len(json.dumps(_sig(flat))) <= 260:
                cfg[k] = flat
    line["config"] = cfg
    rf = full.get("roofline")
    if isinstance(rf, dict):
        line["roofline"] = {k: rf[k] for k in ROOFLINE_KEYS if k in rf}
        ts = line["roofline"].get("traffic_source")
        if isinstance(ts, str) and len(ts) > 80:
            line["roofline"]["traffic_source"] = "replayed from profiles/ (PMC passes of this command)" if "replayed" in ts else ts[:80]
    if "cpu_baseline" in full:
        line["cpu_baseline"] = full["cpu_baseline"]
    if isinstance(full.get("others"), dict):
        line["others"] = {k: compact_leg(v) for k, v in full["others"].items()}
    if detail_name:
        line["detail"] = detail_name
    line = _sig(line)
    # hard bound: shed the least important parts until the line fits
    for shed in ("limit", "unit", "threads", "cpu", "others"):
        if len(json.dumps(line)) <= LINE_MAX_BYTES:
            break
        if shed == "others":
            line.pop("others", None)
        else:
            for leg in line.get("others", {}).values():
                leg.pop(shed, None)
    return line


def emit_line(full: dict, stream, detail_path: Path | None = None) -> str:
    """writes the sidecar (everything) and ONE compact line to `stream`; returns the line"""
    name = None
    if detail_path is not None:
        try:
            detail_path.write_text(json.dumps(_sig(full, 9), indent=1) + "\n")
            name = detail_path.name
        except OSError as e:  # a read-only checkout must not cost the line
            print(f"bench.py: could not write {detail_path}: {e}", file=sys.stderr)
    s_ = json.dumps(compact_line(full, name))
    assert len(s_) < 8000, len(s_)
    stream.write(s_ + "\n")
    stream.flush()
    return s_


def detail_path_default() -> Path:
    """bench_detail.json next to bench.py (VSZIP_BENCH_DETAIL overrides); also mirrored into gpurun_out/ when that exists"""
    return Path(os.environ.get("VSZIP_BENCH_DETAIL") or (ROOT / "bench_detail.json"))



def launch_ranks(a, argv) -> int:
    """`python bench.py --gpus N` with no rendezvous in the environment: start the N ranks as child
    processes (this process never initialises a GPU, so no process that has touched one is ever
    replaced), relay rank 0's single JSON line and return the children's exit code."""
    import socket
    import subprocess

    s_ = socket.socket()
    s_.bind(("127.0.0.1", 0))
    port = s_.getsockname()[1]
    s_.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={a.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), str(Path(__file__).resolve()), *argv]
    if a.dry_launch:
        print(json.dumps({"dry_launch": cmd}))
        return 0
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    r = subprocess.run(cmd, stdout=subprocess.PIPE, env=env)
    lines = [ln for ln in r.stdout.decode(errors="replace").splitlines() if ln.lstrip().startswith("{")]
    if lines:
        print(lines[-1])
    elif r.returncode == 0:
        print("bench.py: the ranks printed no JSON line", file=sys.stderr)
        return 3
    return r.returncode



def plugin_legs_child():
    """The plugin legs in a CHILD process (`bench.py --plugin-legs-only`), waited for: a VapourSynth host is a process in which the plugin is loaded before any HIP
    call, and the order matters - `VapourSynthPluginInit2` sets GPU_MAX_HW_QUEUES=16 (twelve streams folded onto the default four hardware queues make one frame's
    2 ms chain hold up three other frames' kernels) and the runtime reads that when it initialises. Round 6: EEDI3 through the plugin ran 2.5 k fps in such a process
    and 1.1-1.5 k in this one, whose own context had initialised the runtime first; loading the plugin first IN this process gave all of it sixteen queues and cost the
    headline its placement search (profiles/r06_notes.md 13). The child is started where the legs always ran - after the other legs, this process idle - because a
    child run FIRST leaves the driver scrubbing the memory it freed: the headline's first candidate allocations then took 1.3 s each and the search's budget ended
    after one or two (0.63 on one run). Not under a profiler (the child would be an exec behind a preloaded tool): None -> the caller runs the legs in-process."""
    import subprocess

    profiled = any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", "")
    if profiled or os.environ.get("VSZIP_BENCH_PLUGIN_INPROC") == "1":
        return None
    try:
        r = subprocess.run([sys.executable, str(Path(__file__).resolve()), "--plugin-legs-only"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, timeout=600)
        return json.loads(r.stdout.decode().strip().splitlines()[-1])
    except Exception as e:
        print(f"bench.py: the plugin legs' child process failed ({e}); running them in-process", file=sys.stderr)
        return None


def main() -> int:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=2000, help="timed steps (default: ~1.2 s of BoxBlur launches)")
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--frames", type=int, default=64, help="frames per step per GPU (64 4K YUV420P16 frames = 192 planes = one BoxBlur launch)")
    ap.add_argument("--workload", default="boxblur", choices=["boxblur", "bilateral", "ssimulacra2", "pipeline"])
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-others", action="store_true")
    ap.add_argument("--exchange-only", action="store_true", help="--no-cpu --no-others, but the per-clip exchange step (XPSNR accumulators over the collective backend) still runs: the RCCL smoke test")
    ap.add_argument("--radius", type=int, default=RADIUS, help="development: BoxBlur radius (headline = 13)")
    ap.add_argument("--min-seconds", type=float, default=1.0, help="kernel time of the launch-duration sample (roofline.launch_us)")
    ap.add_argument("--dry-launch", action="store_true", help="with --gpus N > 1: print the rank-launch command and exit (no GPU)")
    ap.add_argument("--plugin-legs-only", action="store_true", help="internal: run the legs through libvszip.so and print their records as one JSON line (see plugin_child below)")
    a = ap.parse_args()
    if a.plugin_legs_only:
        sys.path.insert(0, str(ROOT / "tests"))
        out_fd = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
        out_fd.write(json.dumps(plugin_legs()) + "\n")
        out_fd.flush()
        return 0
    if a.exchange_only:
        a.no_cpu = a.no_others = True

    if a.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return launch_ranks(a, [x for x in sys.argv[1:] if x != "--dry-launch"])
    if a.workload != "boxblur" and a.steps == 2000:
        a.steps = 50  # the other workloads' steps are milliseconds, not 0.6 ms

    # stdout carries exactly ONE line, the JSON. Libraries that print to fd 1 (RCCL's version banner
    # does) are sent to stderr for the lifetime of the process; the JSON goes to the saved fd.
    sys.stdout.flush()
    json_out = os.fdopen(os.dup(1), "w")
    os.dup2(2, 1)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != max(1, a.gpus):
        print(f"bench.py: --gpus {a.gpus} but WORLD_SIZE={world}", file=sys.stderr)
        return 2

    # torch first: its bundled HIP runtime must be the one libvszip_hip.so binds to
    import torch
    import torch.distributed as dist

    if not torch.cuda.is_available():
        print("bench.py needs a GPU (no CPU fallback)", file=sys.stderr)
        return 2
    # development only (1-GPU boxes): VSZIP_BENCH_SHARE_GPU=1 folds the ranks onto the visible devices and runs
    # the collectives over gloo (RCCL refuses two ranks on one GPU) — exercises launch, sharding and reduce logic
    share_gpu = os.environ.get("VSZIP_BENCH_SHARE_GPU") == "1"
    if share_gpu:
        local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    # VSZIP_BENCH_FORCE_DIST=1 (under torch.distributed.run with one process): run the RCCL path
    # — barrier, timing max-reduce, per-clip scalar all-reduce — even with a single rank
    use_dist = world > 1 or (os.environ.get("VSZIP_BENCH_FORCE_DIST") == "1" and "MASTER_ADDR" in os.environ)
    if use_dist:
        if share_gpu:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        assert dist.get_world_size() == world
    coll_dev = None if (share_gpu or not use_dist) else "cuda"
    rccl_ranks = 0 if share_gpu else (dist.get_world_size() if use_dist else 1)
    numa_node = bind_to_gpu_numa(local_rank)

    import vszip_amd

    dev = vszip_amd.Device(local_rank)

    def barrier():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()
        dev.sync()

    timed = Timed(dev, barrier, prewarm_s=0.3)

    def max_over_ranks(dt):  # (the legs take a one-argument callable)
        return ranks_max(dt, use_dist, coll_dev)

    def reduce_scalars(v):
        return vszip_amd.cluster.allreduce_clip_scalars(v, device=coll_dev)

    F = a.frames
    out = None
    if a.workload == "pipeline":
        nf = max(1, min(F // 16, 8))  # 64 -> 4 frames of 398 MB per step per GPU
        res = pipeline_line(dev, timed, world, nf, a.steps, a.warmup, max_over_ranks, reduce_scalars)
        out = {"metric": "frames/sec: Bilateral->BoxBlur->SSIMULACRA2 pipeline 7680x4320 RGBS", "value": res.pop("value"), "unit": res.pop("unit"),
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": res.pop("ms_per_step"), "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": res.pop("workload"), "frames_per_step_per_gpu": nf, "parallelism": f"frame-parallel x{world}",
                          "clip_mean_score": res.pop("clip_mean_score")},
               "roofline": res.pop("roofline")}
    elif a.workload == "boxblur":
        step, keep = setup_boxblur(dev, rank, F, a.radius)
        dt, region_ms, dom_ms, launches = timed.run(step, a.steps, a.warmup)
        dt = max_over_ranks(dt)
        frame_bytes = sum(2 * s[0] * s[1] for s in yuv420_shapes(W4K, H4K))  # 24 883 200
        alg_bytes = 2 * frame_bytes * F * a.steps / launches   # per ring-kernel launch: every pixel read once + written once
        avg_s = dom_ms * 1e-3 / launches
        achieved = alg_bytes / avg_s / 1e9
        group_s = region_ms * 1e-3 / launches                   # ring kernel + launch gaps
        out = {
            "metric": "frames/sec at 4K YUV420P16: Bilateral, BoxBlur, SSIMULACRA2 on 1/2/4/8 MI355X",
            "value": whole_job_value(F, a.steps, world, dt), "unit": "frames/s", "n_gpus": world, "steps": a.steps, "warmup": a.warmup,
            "ms_per_step": dt * 1e3 / a.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u16", "data": "synthetic",
            "rccl_ranks": rccl_ranks,
            "collective_backend": dist.get_backend() if use_dist else None,  # "nccl" = RCCL; None: one rank, no process group
            "config": {"workload": "vszip.BoxBlur hradius=vradius=13, 3840x2160 YUV420P16, splitmix64 noise, HBM-resident",
                       "frames_per_step_per_gpu": F, "parallelism": f"frame-parallel x{world}"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                         "traffic": profile_traffic("boxblur_ct_ring_kernel<unsigned short, 13", F) if a.radius == RADIUS else None,
                         "kernel": "boxblur_ct_ring_kernel<u16,13>", "avg_launch_us": avg_s * 1e6, "launches": launches,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "whole_filter": {"note": "ring kernel + launch gaps, HIP events over the whole timed region",
                                          "avg_us": group_s * 1e6, "achieved": alg_bytes / group_s / 1e9, "frac": alg_bytes / group_s / 1e9 / HBM_PEAK_GBS}},
        }
        # How the two arenas were allocated, as SCALARS (the driver's record keeps scalar config fields only)
        pl_info = keep[2]
        ar = pl_info.get("arena", {})
        out["config"]["placement_policy"] = ("vszip_dev_alloc: the fastest of up to 24 probed candidate allocations, nothing kept (include/vszip_hip.h)"
                                             if ar.get("candidates") else "plain hipMalloc")
        out["config"]["arena_candidates"] = ar.get("candidates")
        out["config"]["arena_probe_TBps"] = (ar.get("probe_bytes_per_second") or 0) / 1e12
        out["config"]["arena_search_ms"] = ar.get("search_ms")
        out["config"]["placement_seconds"] = pl_info.get("alloc_seconds")
        out["config"]["value_is"] = "frames of all ranks (the same batch on every rank) / the slowest rank's time of the timed region"
        # every rank's device, NUMA node, arena and launch time (sidecar): one process per GPU, local_rank == device index
        assert share_gpu or local_rank == torch.cuda.current_device(), (local_rank, torch.cuda.current_device())
        ranks = gather_rank_records({"rank": rank, "local_rank": local_rank, "device": torch.cuda.current_device(), "numa_node": numa_node, "pcie_path": pcie_path(local_rank),
                                     "arena": pl_info.get("arena"), "alloc_seconds": pl_info.get("alloc_seconds"), "avg_launch_us": avg_s * 1e6,
                                     "frac": achieved / HBM_PEAK_GBS}, use_dist)
        if rank == 0:
            out["config"]["ranks"] = ranks
            out["config"]["slowest_rank_frac"] = min(r["frac"] for r in ranks)
        prof = profile_launch_us("boxblur_ct_ring_kernel<unsigned short, 13") if a.radius == RADIUS else None
        if prof:  # both bases of the fraction on the line: HIP events of THIS run (frac) and the committed rocprofv3 average (frac_rocprof)
            out["roofline"]["frac_rocprof"] = alg_bytes / (prof[0] * 1e-6) / 1e9 / HBM_PEAK_GBS
            out["roofline"]["rocprof_avg_launch_us"] = prof[0]
            out["roofline"]["rocprof_source"] = f"profiles/{prof[2]} ({prof[1]} calls; replayed)"
        out["roofline"]["traffic_source"] = ("replayed from the committed PMC passes of this command (profiles/r*_boxblur_pmc.json), not measured in this run"
                                             if out["roofline"]["traffic"] is not None else None)
        if rank == 0:
            st_ = timed.launch_stats(step, a.min_seconds)
            if st_:
                out["roofline"]["launch_us"] = st_
                out["roofline"]["frac_median"] = alg_bytes / (st_["median"] * 1e-6) / 1e9 / HBM_PEAK_GBS
        barrier()
        # Outside the timed region: the one collective of the design — a per-clip scalar (here the
        # mean luma of the blurred clip, PlaneAverage over every rank's frames) all-reduced over
        # RCCL, like XPSNR's / SSIMULACRA2's per-clip sums (vszip_amd.cluster).
        try:
            luma = keep[1].planes[0::3]
            avgs, _ = dev.plane_average(luma[:48], exclude=[-1])
            tot = vszip_amd.cluster.allreduce_clip_scalars(np.array([float(np.sum(avgs)), float(len(avgs))]), device=coll_dev)
            out["config"]["clip_mean_luma"] = {"value": float(tot[0] / tot[1]), "frames": int(tot[1]), "reduced_over_ranks": world}
        except Exception as e:  # informative only
            out["config"]["clip_mean_luma"] = {"error": str(e)}
        del keep
        # Side scalar only (never `value`): the same launch on arenas the allocator did NOT place (VSZIP_PLACEMENT=0: plain hipMalloc)
        if rank == 0 and world == 1 and not a.no_others and a.radius == RADIUS:
            try:
                with dev.options(VSZIP_PLACEMENT=0):
                    step_p, keep_p = setup_boxblur(dev, rank, F, a.radius)
                _, _, dom_p, n_p = timed.run(step_p, 200, 5)
                out["config"]["unplaced_frac"] = alg_bytes / (dom_p * 1e-3 / n_p) / 1e9 / HBM_PEAK_GBS
                out["config"]["unplaced_launch_us"] = dom_p * 1e3 / n_p
                del step_p, keep_p
            except Exception as e:
                out["config"]["unplaced_error"] = str(e)[:100]
        # The exchange step on real data: XPSNR's per-clip accumulators over RCCL (all ranks take part)
        try:
            if not a.no_others or a.exchange_only:
                out["config"]["xpsnr_clip"] = xpsnr_clip_leg(dev, vszip_amd, rank, world, coll_dev)
        except Exception as e:
            out["config"]["xpsnr_clip"] = {"error": str(e)}
        # PCIe-fed rate at this N: every rank feeds its GPU from pinned host frames at the same time
        try:
            if a.no_others:
                raise KeyboardInterrupt
            pc = pcie_boxblur(vszip_amd, local_rank, a.radius, barrier=barrier)
            dt_p = max_over_ranks(pc["seconds"])
            out["config"]["pcie_fed_fps"] = world * pc["contexts"] * pc["rounds"] / dt_p
            out["config"]["pcie_fed"] = {"rank0_fps": pc["value"], "rank0_GBps_each_direction": pc["pcie_GBps_each_direction"], "contexts_per_gpu": pc["contexts"],
                                         "rank0_numa_node": numa_node, "rank0_pcie_path": pcie_path(local_rank),
                                         "note": "pinned host frame -> GPU -> pinned host frame, every rank at once (barrier, then the same number of rounds); "
                                                 "whole-job frames / max-over-ranks time"}
        except KeyboardInterrupt:
            pass
        except Exception as e:
            out["config"]["pcie_fed"] = {"error": str(e)}
        if rank == 0 and world == 1 and not a.no_cpu:
            out["cpu_baseline"] = cpu_boxblur()
        if world == 1 and not a.no_others:
            others = {}
            for name, (w, h, nf) in {"bilateral_1080p": (W1080, H1080, 64), "bilateral_4k": (W4K, H4K, 16)}.items():
                st, keep = setup_bilateral(dev, w, h, nf)
                dt2, kms, dms, nl = timed.run(st, 10, 2)
                fb2 = sum(2 * s_[0] * s_[1] for s_ in yuv420_shapes(w, h))
                others[name] = {"value": nf * 10 / dt2, "unit": "frames/s", "kernel_ms_per_frame": kms / (10 * nf), "frames_per_call": nf,
                                "roofline": {"bound": "hbm", "achieved": 2 * fb2 * nf * 10 / (dms * 1e-3) / 1e9, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                             "frac": 2 * fb2 * nf * 10 / (dms * 1e-3) / 1e9 / HBM_PEAK_GBS, "traffic": None,
                                             "kernel": "bilateral_walk16_kernel<3,2> luma + <2,1> chroma (column-walking waves, exact range LUT packed in LDS)", "avg_launch_us": dms * 1e3 / nl,
                                             "alg_bytes_per_call": 2 * fb2 * nf, "basis": "dominant kernel", "kernel_match": "bilateral_walk16_kernel"},
                                "workload": f"vszip.Bilateral sigmaS=2 sigmaR=2 {w}x{h} YUV420P16 (natural content tiled), HBM-resident"}
                others[name]["roofline"]["kernel"] = "bilateral_walk16_kernel<3,2> + <2,1> (symmetric weights looked up once, table in LDS)"
                others[name]["limit"] = limit_from_profile("bilateral", "bilateral_walk16_kernel<3")
                if not a.no_cpu:
                    others[name]["cpu_baseline"] = cpu_bilateral(w, h, 5.0 if h < 2000 else 3.0)
                del keep
            # round 3: the same filter at the range sigma people use it with (the BASELINE's sigmaR = 2 is a gentle table; the filter's
            # default is 0.02, a steep one) and at the filter's default spatial sigma (3: luma radius 5, three tap distances)
            for name, (ss, sr, b8) in {"bilateral_1080p_sigmaR0p02": (2, 0.02, False), "bilateral_1080p_defaults": (3, 0.02, False), "bilateral_1080p_yuv420p8": (2, 2, True)}.items():
                st, keep = setup_bilateral(dev, W1080, H1080, 64, ss, sr, b8)
                dt2, kms, _, _ = timed.run(st, 10, 2)
                fb2 = sum((1 if b8 else 2) * s_[0] * s_[1] for s_ in yuv420_shapes(W1080, H1080))
                gb = 2 * fb2 * 64 * 10 / dt2 / 1e9
                others[name] = {"value": 64 * 10 / dt2, "unit": "frames/s", "frames_per_call": 64,
                                "roofline": {"bound": "hbm", "achieved": gb, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": gb / HBM_PEAK_GBS, "traffic": None,
                                             "alg_bytes_per_call": 2 * fb2 * 64, "basis": "wall",
                                             "note": "whole call; the range table's computed part (up to the reference's `upper` cut) held as it is in LDS (DESIGN.md 3.3, PLATEAU form)"},
                                "workload": f"vszip.Bilateral sigmaS={ss} sigmaR={sr} 1920x1080 {'YUV420P8' if b8 else 'YUV420P16'} (natural content tiled), 64 frames per call, HBM-resident"}
                del keep
            st, keep = setup_ssimulacra2(dev, W4K, H4K, 16)
            dt3, kms, _, _ = timed.run(st, 5, 1)
            ss_gbs = 2 * 3 * W4K * H4K * 4 * 16 * 5 / dt3 / 1e9  # algorithmic bytes: the two RGBS input frames
            others["ssimulacra2_4k"] = {"value": 16 * 5 / dt3, "unit": "pairs/s", "ms_per_pair": dt3 * 1e3 / 80, "pairs_per_call": 16,
                                        "roofline": {"bound": "hbm", "achieved": ss_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": ss_gbs / HBM_PEAK_GBS, "traffic": None,
                                                     "alg_bytes_per_call": 2 * 3 * W4K * H4K * 4 * 16, "basis": "wall",
                                                     "note": "whole pipeline on the two input frames; ssim_maps_kernel (65 % of a pair) is issue bound: "
                                                             "about 230 instructions per plane-pixel, a quarter of them f64 (DESIGN.md 3.4)"},
                                        "workload": "vszip.SSIMULACRA2 ref vs dist, 3840x2160 RGBS (linear), HBM-resident; includes the scalar D2H + sync"}
            others["ssimulacra2_4k"]["limit"] = limit_from_profile("ssimulacra2", "ssim_maps_kernel")
            if not a.no_cpu:
                others["ssimulacra2_4k"]["cpu_baseline"] = cpu_ssimulacra2(W4K, H4K, 6.0)
            del keep
            st, keep = setup_ssimulacra2_rgb24(dev, W4K, H4K, 16)
            dt3, _, _, _ = timed.run(st, 5, 1)
            others["ssimulacra2_4k_rgb24"] = {"value": 16 * 5 / dt3, "unit": "pairs/s", "ms_per_pair": dt3 * 1e3 / 80, "pairs_per_call": 16,
                                              "workload": "vszip.SSIMULACRA2 ref vs dist from 3840x2160 RGB24 planes (colour pre-stage on the device), HBM-resident"}
            del keep
            st, keep = setup_ssimulacra2_yuv420p8(dev, W4K, H4K, 16)
            dt3, _, _, _ = timed.run(st, 5, 1)
            others["ssimulacra2_4k_yuv420p8"] = {"value": 16 * 5 / dt3, "unit": "pairs/s", "ms_per_pair": dt3 * 1e3 / 80, "pairs_per_call": 16,
                                                 "workload": "vszip.SSIMULACRA2 ref vs dist from 3840x2160 YUV420P8 planes (chroma upsampling + matrix + EOTF fused into "
                                                             "the first pass; 24.9 MB per pair), HBM-resident"}
            del keep
            st, keep = setup_ssimulacra2_yuv444p8(dev, W4K, H4K, 16)
            dt3, _, _, _ = timed.run(st, 5, 1)
            with dev.options(VSZIP_SSIM_NO_YUV420_LDS=1):  # the fused tile kernel these clips took until round 6
                dt4, _, _, _ = timed.run(st, 3, 1)
            others["ssimulacra2_4k_yuv444p8"] = {"value": 16 * 5 / dt3, "unit": "pairs/s", "ms_per_pair": dt3 * 1e3 / 80, "pairs_per_call": 16, "fused_tile_kernel_pairs_s": 16 * 3 / dt4,
                                                 "workload": "vszip.SSIMULACRA2 ref vs dist from 3840x2160 YUV444P8 planes (matrix + EOTF as a pre-stage pass with the table in LDS; "
                                                             "49.8 MB per pair), HBM-resident; fused_tile_kernel_pairs_s: the same call with VSZIP_SSIM_NO_YUV420_LDS=1"}
            del keep
            for leg_name, leg in (("eedi3_1080p", lambda: eedi3_leg(dev, timed, a.no_cpu)), ("xpsnr_1080p", lambda: xpsnr_leg(dev, timed, a.no_cpu))):
                try:
                    others[leg_name] = leg()
                except Exception as e:
                    others[leg_name] = {"error": str(e)}
            try:
                others.update(boxblur_other_paths_leg(dev, timed))
            except Exception as e:
                others["boxblur_other_paths"] = {"error": str(e)}
            try:
                others.update(planestats_leg(dev, timed))
            except Exception as e:
                others["plane_stats_4k"] = {"error": str(e)}
            try:
                others.update(limiter_leg(dev, timed))
                others.update(limit_filter_leg(dev, timed))
            except Exception as e:
                others["limiter_4k"] = {"error": str(e)}
            try:
                others["pipeline_8k_rgbs"] = pipeline_line(dev, timed, 1, 2, 3, 1, lambda t: t, reduce_scalars)
            except Exception as e:
                others["pipeline_8k_rgbs"] = {"error": str(e)}
            try:
                others["boxblur_1080p"] = boxblur_1080p_leg(dev, timed, a.no_cpu)
            except Exception as e:
                others["boxblur_1080p"] = {"error": str(e)}
            try:
                others["boxblur_1080p_5pass"] = boxblur_1080p_5pass_leg(dev, timed, a.no_cpu)
                others["boxblur_1080p_r1x2_yuv420p8"] = boxblur_gauss_leg(dev, timed)
            except Exception as e:
                others["boxblur_1080p_5pass"] = {"error": str(e)}
            try:
                out["config"]["frames_per_call_sweep"] = frames_per_call_sweep(dev, timed)
            except Exception as e:
                out["config"]["frames_per_call_sweep"] = {"error": str(e)[:100]}
            try:
                plugin_child = plugin_legs_child()
                others.update(plugin_child if plugin_child is not None else plugin_legs())
            except Exception as e:
                others["plugin_legs"] = {"error": str(e)}
            out["others"] = others
            # the metric string's other two filters (and EEDI3) as scalars, so that they survive into the driver's record
            for key, leg in (("bilateral_1080p_fps", "bilateral_1080p"), ("bilateral_4k_fps", "bilateral_4k"), ("ssimulacra2_4k_pairs_s", "ssimulacra2_4k"),
                             ("bilateral_1080p_sigmaR0p02_fps", "bilateral_1080p_sigmaR0p02"), ("bilateral_1080p_defaults_fps", "bilateral_1080p_defaults"), ("bilateral_1080p_yuv420p8_fps", "bilateral_1080p_yuv420p8"), ("ssimulacra2_4k_yuv420p8_pairs_s", "ssimulacra2_4k_yuv420p8"), ("eedi3_1080p_fps", "eedi3_1080p"), ("xpsnr_1080p_fps", "xpsnr_1080p"),
                             ("boxblur_1080p_fps", "boxblur_1080p"), ("boxblur_1080p_5pass_fps", "boxblur_1080p_5pass"), ("boxblur_1080p_r1x2_yuv420p8_fps", "boxblur_1080p_r1x2_yuv420p8"), ("pipeline_8k_fps", "pipeline_8k_rgbs"),
                             ("boxblur_rt_float_r5x3_4k_fps", "boxblur_rt_float_r5x3_4k"), ("boxblur_rt_r5x3_4k_fps", "boxblur_rt_r5x3_4k"),
                             ("plugin_ssimulacra2_4k_yuv420p8_pairs_s", "plugin_ssimulacra2_4k_yuv420p8"), ("plugin_ssimulacra2_4k_rgb24_pairs_s", "plugin_ssimulacra2_4k_rgb24")):
                v = others.get(leg, {}).get("value")
                if isinstance(v, (int, float)):
                    out["config"][key] = v
    elif a.workload == "bilateral":
        step, keep = setup_bilateral(dev, W1080, H1080, F)
        dt, _, dom_ms, launches = timed.run(step, a.steps, a.warmup)
        dt = max_over_ranks(dt)
        fb = sum(2 * s[0] * s[1] for s in yuv420_shapes(W1080, H1080))
        avg_s = dom_ms * 1e-3 / a.steps
        achieved = 2 * fb * F / avg_s / 1e9
        out = {"metric": "frames/sec: vszip.Bilateral sigmaS=2 sigmaR=2 1920x1080 YUV420P16", "value": world * F * a.steps / dt, "unit": "frames/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt * 1e3 / a.steps, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "vszip.Bilateral sigmaS=2 sigmaR=2, 1920x1080 YUV420P16, natural content tiled", "frames_per_step_per_gpu": F},
               "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                            "kernel": "bilateral_walk16_kernel<3,2> luma + <2,1> chroma (column-walking waves, exact range LUT packed in LDS)"}}
        if rank == 0 and world == 1 and not a.no_cpu:
            out["cpu_baseline"] = cpu_bilateral(W1080, H1080)
    else:
        pairs = max(1, F // 4)
        step, keep = setup_ssimulacra2(dev, W4K, H4K, pairs)
        dt, kernel_ms, _, _ = timed.run(step, a.steps, a.warmup)
        dt = max_over_ranks(dt)
        avg_s = kernel_ms * 1e-3 / a.steps  # whole pipeline (6 scales x {xyb/downscale, maps} + final), not one kernel
        achieved = 2 * 3 * W4K * H4K * 4 * pairs / avg_s / 1e9
        out = {"metric": "pairs/sec: vszip.SSIMULACRA2 3840x2160 RGBS", "value": world * pairs * a.steps / dt, "unit": "pairs/s",
               "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": dt * 1e3 / a.steps, "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": "vszip.SSIMULACRA2 ref vs dist, 3840x2160 RGBS linear", "pairs_per_step_per_gpu": pairs},
               "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                            "kernel": "whole SSIMULACRA2 pipeline (ssim_pyr: scales 0+1 in one pass over the source; ssim_xyb_down x3, ssim_maps x5, ssim_final); algorithmic bytes = the two input frames"}}
        if rank == 0 and world == 1 and not a.no_cpu:
            out["cpu_baseline"] = cpu_ssimulacra2(W4K, H4K)

    if rank == 0:
        out["config"]["timed_calls"] = getattr(timed, "calls", 0) + UNTIMED_CALLS
        out["config"]["barrier_ms"] = getattr(timed, "barrier_ms", None)  # one barrier (all ranks + device sync) as timed before the last timed region  # calls of the workload's step in this process (tools/summarize_prof.py: launches per call)
        emit_line(out, json_out, detail_path_default())
        try:  # gpurun merges gpurun_out/ back: keep a copy of the full record there
            if (ROOT / "gpurun_out").is_dir():
                (ROOT / "gpurun_out" / "bench_detail.json").write_text(detail_path_default().read_text())
        except OSError:
            pass
    if use_dist:
        dist.destroy_process_group()
    dev.close()
    return 0


if __name__ == "__main__":
    sys.exit(main())
