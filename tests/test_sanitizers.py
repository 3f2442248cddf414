"""CPU-side sanitizer runs (the reference's CI builds Zig Debug = bounds/overflow checked,
.github/workflows/test.yml:27): the plugin (1.8 k lines of C++ with manual freeFrame / freeNode on every error
path), the test host and the oracle under AddressSanitizer + UndefinedBehaviorSanitizer. libvszip.so links a
GPU-less stand-in for libvszip_hip.so (tests/sanitize/stub_hip.cpp: real allocations and copies in host memory,
every filter entry point touches the extents of the planes it is handed and then fails), so staging, error and
unwind paths run without a GPU. No GPU sanitizer exists on this pool; these are CPU builds only."""
import os
import subprocess
import sys
import textwrap
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
SAN = ROOT / "tests" / "sanitize"


def _runtime(name):
    p = subprocess.run(["g++", f"-print-file-name={name}"], capture_output=True, text=True).stdout.strip()
    return p if p and os.path.isabs(p) and os.path.exists(p) else None


@pytest.fixture(scope="module")
def san_env():
    asan = _runtime("libasan.so")
    if not asan:
        pytest.skip("g++ has no libasan")
    r = subprocess.run(["make", "-C", str(SAN)], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    r = subprocess.run(["make", "-C", str(ROOT / "oracle"), "asan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    env = dict(os.environ)
    env.update(LD_PRELOAD=asan, ASAN_OPTIONS="detect_leaks=0:abort_on_error=0:halt_on_error=1:exitcode=99", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1:exitcode=98",
               VSZIP_FAKEVS_LIB=str(SAN / "_build" / "libfakevs.so"), VSZIP_PLUGIN_LIB=str(SAN / "_build" / "libvszip.so"),
               VSZIP_ORACLE_LIB=str(ROOT / "oracle" / "_asan" / "liboracle_vszip.so"), PYTHONMALLOC="malloc")
    return env


def _clean(r):
    out = r.stdout + r.stderr
    assert "AddressSanitizer" not in out and "runtime error:" not in out and r.returncode not in (98, 99), out[-4000:]


def test_plugin_boundary_suite_under_asan(san_env):
    """tests/test_plugin_boundary.py (registration, create-time validation, error strings: the reference's
    validation tests) against the instrumented plugin + host."""
    r = subprocess.run([sys.executable, "-m", "pytest", str(ROOT / "tests" / "test_plugin_boundary.py"), "-x", "-q", "-p", "no:cacheprovider"],
                       capture_output=True, text=True, env=san_env, cwd=str(ROOT), timeout=900)
    _clean(r)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]


SCENARIO = textwrap.dedent("""
    import sys, numpy as np
    sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
    from fakevs import fakevs as vs
    rng = np.random.default_rng(0)
    def yuv(dt, w=134, h=96, n=3, hi=255):
        shapes = [(h, w), (h // 2, w // 2), (h // 2, w // 2)]
        return [[(rng.random(s) * hi).astype(dt) for s in shapes] for _ in range(n)]
    c8 = vs.source(yuv(np.uint8), vs.YUV420P8)
    c16 = vs.source(yuv(np.uint16, hi=65535), vs.YUV420P16, extra_stride=24, offset=8)   # cropped-clip layout: pitch > width
    cf = vs.source(yuv(np.float32, hi=1.0), vs.YUV420PS)
    rgb = vs.source([[(rng.random((96, 134)) * 255).astype(np.uint8) for _ in range(3)] for _ in range(2)], vs.RGB24)
    rgbs = vs.source([[rng.random((96, 134)).astype(np.float32) for _ in range(3)] for _ in range(2)], vs.RGBS, props={{"_Transfer": 8}})
    g8 = vs.source([[(rng.random((96, 134)) * 255).astype(np.uint8)] for _ in range(2)], vs.GRAY8)
    gs = vs.source([[rng.random((96, 134)).astype(np.float32)] for _ in range(2)], vs.GRAYS)
    vs.core_standins(True)
    clips = [
        c16.vszip.BoxBlur(hradius=13, vradius=13), c8.vszip.BoxBlur(hradius=3, vradius=5, hpasses=2, planes=[0, 2]),
        c16.vszip.Bilateral(sigmaS=2.0, sigmaR=2.0), c16.vszip.Bilateral(ref=c16, sigmaS=3.0, sigmaR=0.02, planes=[0]),
        c8.vszip.PlaneAverage(exclude=[-1]), c8.vszip.PlaneAverage(exclude=[3, 4], clipb=c8), c16.vszip.PlaneMinMax(minthr=0.1, maxthr=0.1, clipb=c16),
        c8.vszip.XPSNR(c8), c8.vszip.XPSNR(c8, temporal=False), rgbs.vszip.SSIMULACRA2(rgbs), rgb.vszip.SSIMULACRA2(rgb), g8.vszip.SSIMULACRA2(g8),
        rgb.vszip.SSIMULACRA2(rgbs), cf.vszip.EEDI3(field=1, dh=True), cf.vszip.EEDI3H(field=0), gs.vszip.EEDI3(field=1, sclip=gs, mclip=g8, vcheck=3),
        c16.vszip.Limiter(tv_range=True), c16.vszip.LimitFilter(c16, dark_thr=8.0), c16.vszip.LimitFilter(c16, c16, elast=3.0), g8.vszip.AdaptiveBinarize(g8, c=3),
        # fused chains (one getFrame runs the upstream stages): their unwind paths, and the references they hold
        c16.vszip.Bilateral(sigmaS=2.0, sigmaR=0.05).vszip.BoxBlur(hradius=3, vradius=3, planes=[0]).vszip.Limiter(tv_range=True, planes=[1, 2]),
        rgbs.vszip.SSIMULACRA2(rgbs.vszip.Bilateral(sigmaS=2.0, sigmaR=2.0).vszip.BoxBlur(hradius=2, vradius=2)),
        rgb.vszip.SSIMULACRA2(rgb.vszip.BoxBlur(hradius=1, vradius=1)),
        # YUV sources take the device pre-stage; a fused reference whose root IS the (linear RGBS) distorted clip (ADVICE r2)
        c8.vszip.SSIMULACRA2(c8), c16.vszip.SSIMULACRA2(c16.vszip.BoxBlur(hradius=1, vradius=1)), rgbs.vszip.BoxBlur(hradius=2, vradius=2).vszip.SSIMULACRA2(rgbs),
    ]
    failed = 0
    for c in clips:
        for n in range(2):
            try:
                c.get_frame(n)
            except vs.Error as e:
                failed += 1
                assert any(s in str(e) for s in ("GPU kernel failed", "device staging failed", "LUT upload failed", "colour pre-stage failed")), str(e)
    assert failed == 2 * len(clips), failed
    # worker threads pulling frames at once (fmParallel): every frame fails, nothing may be touched after release
    try:
        clips[0].pull(12, 4)
    except vs.Error:
        pass
    del clips, c  # every filter instance is freed: chains release the upstream nodes they hold
    print("scenario ok", failed)
""")


@pytest.mark.parametrize("mode", ["kernel", "alloc", "copy"])
def test_error_and_unwind_paths_under_asan(san_env, mode):
    """Every filter's getFrame through staging, a failing kernel / allocation / copy, and the release of every
    frame and device pointer it held."""
    env = dict(san_env)
    if mode != "kernel":
        env["VSZIP_STUB_FAIL"] = mode
    r = subprocess.run([sys.executable, "-c", SCENARIO.format(root=str(ROOT), tests=str(ROOT / "tests"))], capture_output=True, text=True, env=env, timeout=900)
    _clean(r)
    assert r.returncode == 0 and "scenario ok" in r.stdout, (r.stdout + r.stderr)[-3000:]


def test_oracle_under_asan(san_env):
    """The CPU oracle (2 k lines of pointer arithmetic) on its golden tests, instrumented."""
    r = subprocess.run([sys.executable, "-m", "pytest", str(ROOT / "tests" / "test_oracle_goldens.py"), str(ROOT / "tests" / "test_oracle_boxblur.py"), "-x", "-q",
                        "-p", "no:cacheprovider", "-k", "not exhaustive and not closed_forms"], capture_output=True, text=True, env=san_env, cwd=str(ROOT), timeout=1800)
    _clean(r)
    assert r.returncode == 0, (r.stdout + r.stderr)[-3000:]


SUCCESS = textwrap.dedent("""
    import sys, numpy as np
    sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
    from fakevs import fakevs as vs
    rng = np.random.default_rng(0)
    def yuv(dt, w=134, h=96, n=6, hi=255):
        shapes = [(h, w), (h // 2, w // 2), (h // 2, w // 2)]
        return [[(rng.random(s) * hi).astype(dt) for s in shapes] for _ in range(n)]
    c8 = vs.source(yuv(np.uint8), vs.YUV420P8)
    c16 = vs.source(yuv(np.uint16, hi=65535), vs.YUV420P16, extra_stride=24, offset=8)
    cf = vs.source(yuv(np.float32, hi=1.0), vs.YUV420PS)
    rgb = vs.source([[(rng.random((96, 134)) * 255).astype(np.uint8) for _ in range(3)] for _ in range(6)], vs.RGB24)
    rgbs = vs.source([[rng.random((96, 134)).astype(np.float32) for _ in range(3)] for _ in range(6)], vs.RGBS, props={{"_Transfer": 8}})
    vs.core_standins(True)
    clips = [
        c16.vszip.BoxBlur(hradius=13, vradius=13), c16.vszip.Bilateral(sigmaS=2.0, sigmaR=2.0), c8.vszip.PlaneAverage(exclude=[-1]),
        c16.vszip.PlaneMinMax(minthr=0.1, maxthr=0.1, clipb=c16), c8.vszip.XPSNR(c8), rgb.vszip.SSIMULACRA2(rgb), cf.vszip.EEDI3(field=1, dh=True),
        c16.vszip.LimitFilter(c16, dark_thr=8.0),
        c16.vszip.Bilateral(sigmaS=2.0, sigmaR=0.05).vszip.BoxBlur(hradius=3, vradius=3, planes=[0]).vszip.Limiter(tv_range=True, planes=[1, 2]),
        rgbs.vszip.SSIMULACRA2(rgbs.vszip.Bilateral(sigmaS=2.0, sigmaR=2.0).vszip.BoxBlur(hradius=2, vradius=2)),
        # ADVICE r2 (high): the fused reference's root is the distorted clip itself, which takes the host-RGBS path
        rgbs.vszip.BoxBlur(hradius=2, vradius=2).vszip.SSIMULACRA2(rgbs), c8.vszip.SSIMULACRA2(c8.vszip.BoxBlur(hradius=1, vradius=1)),
    ]
    for c in clips:
        c.get_frame(0)              # every filter's success path once, single threaded
        c.pull(6, {threads})        # ... and from worker threads at once (fmParallel): contexts, the per-GPU gate, the stage registry
    del clips, c
    print("success ok")
""")


def test_success_paths_under_asan(san_env):
    """The stub's kernels SUCCEED (VSZIP_STUB_FAIL=none): download, frame properties, fused chains and the release of
    every frame / node / device pointer on the good path, single threaded and from 4 worker threads."""
    env = dict(san_env, VSZIP_STUB_FAIL="none")
    r = subprocess.run([sys.executable, "-c", SUCCESS.format(root=str(ROOT), tests=str(ROOT / "tests"), threads=4)], capture_output=True, text=True, env=env, timeout=900)
    _clean(r)
    assert r.returncode == 0 and "success ok" in r.stdout, (r.stdout + r.stderr)[-3000:]


RANDOM_GRAPHS = textwrap.dedent("""
    import sys, numpy as np
    sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
    from fakevs import fakevs as vs
    import test_gpu_plugin_random as g
    n = 0
    for seed in range({graphs}):
        rng = np.random.default_rng(77000 + seed)
        sink = ["pixel", "limitfilter", "average", "minmax", "xpsnr", "ssim"][seed % 6]
        fmt = vs.YUV420P8 if sink == "xpsnr" else (vs.RGBS if sink == "ssim" else vs.YUV420P16)
        props = {{"_Transfer": 8}} if fmt == vs.RGBS else None
        ra = vs.source(g._frames(rng, fmt), fmt, props=props)
        rb = vs.source(g._frames(rng, fmt), fmt, props=props) if rng.integers(0, 3) == 0 else ra
        a = g._chain(ra, [g._stage(rng) for _ in range(int(rng.integers(1, 4)))], fmt, True)
        b = g._chain(rb, [g._stage(rng) for _ in range(int(rng.integers(0, 3)))], fmt, True)
        if sink == "pixel": c = a.vszip.BoxBlur(hradius=2, vradius=2)
        elif sink == "limitfilter": c = a.vszip.LimitFilter(b, dark_thr=8, ref=g._chain(ra, [g._stage(rng)], fmt, True))
        elif sink == "average": c = a.vszip.PlaneAverage(exclude=[-1], planes=[0, 1, 2], clipb=b)
        elif sink == "minmax": c = a.vszip.PlaneMinMax(minthr=0.1, maxthr=0.1, planes=[0, 1, 2], clipb=b)
        elif sink == "ssim": c = a.vszip.SSIMULACRA2(b)
        else: c = b.vszip.XPSNR(a, verbose=False)
        c.get_frame(0)
        c.pull(g.NFR, 3)
        del a, b, c, ra, rb
        n += 1
    print("graphs ok", n)
""")


def test_random_filter_graphs_under_asan(san_env):
    """The seeded random scripts of tests/test_gpu_plugin_random.py (fused side only) against the stub with succeeding kernels: the
    fusion registry, the multi-input staging and every release on graphs nobody wrote by hand, single threaded and from 3 workers."""
    env = dict(san_env, VSZIP_STUB_FAIL="none")
    r = subprocess.run([sys.executable, "-c", RANDOM_GRAPHS.format(root=str(ROOT), tests=str(ROOT / "tests"), graphs=48)], capture_output=True, text=True, env=env, timeout=1200)
    _clean(r)
    assert r.returncode == 0 and "graphs ok 48" in r.stdout, (r.stdout + r.stderr)[-3000:]


def test_worker_threads_under_tsan():
    """Race detection: the plugin, the stub device library and the test host built with
    -fsanitize=thread; 8 worker threads pull frames of every filter kind at once (shared per-instance state:
    context pools, the per-GPU gate, LUT caches, the fusion registry, XPSNR's accumulators)."""
    tsan = _runtime("libtsan.so")
    if not tsan:
        pytest.skip("g++ has no libtsan")
    r = subprocess.run(["make", "-C", str(SAN), "tsan"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    env = dict(os.environ)
    env.update(LD_PRELOAD=tsan, TSAN_OPTIONS="halt_on_error=0:exitcode=97:report_signal_unsafe=0", VSZIP_STUB_FAIL="none",
               VSZIP_FAKEVS_LIB=str(SAN / "_build_tsan" / "libfakevs.so"), VSZIP_PLUGIN_LIB=str(SAN / "_build_tsan" / "libvszip.so"))
    r = subprocess.run([sys.executable, "-c", SUCCESS.format(root=str(ROOT), tests=str(ROOT / "tests"), threads=8)], capture_output=True, text=True, env=env, timeout=1800)
    out = r.stdout + r.stderr
    assert "ThreadSanitizer" not in out and r.returncode != 97, out[-6000:]
    assert r.returncode == 0 and "success ok" in r.stdout, out[-3000:]


ROUND_ROBIN = textwrap.dedent("""
    import ctypes, os, sys, numpy as np
    sys.path.insert(0, {root!r}); sys.path.insert(0, {tests!r})
    from fakevs import fakevs as vs
    rng = np.random.default_rng(1)
    n = 12
    frames = [[(rng.random(s) * 60000).astype(np.uint16) + 1000 for s in ((64, 96), (32, 48), (32, 48))] for _ in range(n)]
    clip = vs.source(frames, vs.YUV420P16)
    out = clip.vszip.BoxBlur(hradius=2, vradius=2)
    # the stub stamps the device index of the context a frame's BoxBlur ran on into every plane's first sample
    for i in range(n):
        f = out.get_frame(i)
        stamps = [int(f[p][0, 0]) for p in range(3)]
        assert stamps == [i % 3] * 3, (i, stamps)
    stub = ctypes.CDLL(os.path.join(os.path.dirname(os.environ["VSZIP_PLUGIN_LIB"]), "libvszip_hip.so"))
    stub.vszip_stub_device_calls.restype = ctypes.c_long
    calls = [stub.vszip_stub_device_calls(d) for d in range(4)]
    assert calls == [4, 4, 4, 0], calls
    out.pull(2 * n, 6)  # and from six worker threads at once: three gates, three sets of slot contexts
    calls = [stub.vszip_stub_device_calls(d) for d in range(4)]
    assert calls == [12, 12, 12, 0], calls
    print("round robin ok")
""")


def test_frames_round_robin_over_three_devices(san_env):
    """Multi-GPU through the plugin: frame n runs on GPU n mod G, every GPU with its own gate
    slots and contexts. The stub library offers three fake devices and says which one each frame ran on; 12 frames from
    6 worker threads, under ASan."""
    env = dict(san_env, VSZIP_STUB_FAIL="none", VSZIP_STUB_DEVICES="3")
    r = subprocess.run([sys.executable, "-c", ROUND_ROBIN.format(root=str(ROOT), tests=str(ROOT / "tests"))], capture_output=True, text=True, env=env, timeout=600)
    _clean(r)
    assert r.returncode == 0 and "round robin ok" in r.stdout, (r.stdout + r.stderr)[-3000:]
