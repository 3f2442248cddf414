"""Build libvszip_hip.so (HIP kernels + C ABI) for gfx950 with hipcc, in tree.

    python vapoursynth-zip_amd/build.py [--force] [--keep-temps]

hipcc cross-compiles without a GPU. -ffp-contract=off: the float kernels must
keep the reference's unfused f32 operation order; fmaf() is written where the
reference writes @mulAdd.
"""
from __future__ import annotations

import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

PKG = Path(__file__).resolve().parent
CSRC = PKG / "csrc"
OBJ = CSRC / "_build"
LIB = PKG / "libvszip_hip.so"
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = [
    "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
    "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function",
]
# per-source flags. eedi3: the SLP vectoriser pairs the line kernel's f32 adds into v_pk_add_f32, which on gfx950's 32-lane SIMDs saves no cycles,
# and pays for the pairs with register copies (measured: +1.7 % with it off, profiles/r04_notes.md section 10)
FILE_FLAGS = {"eedi3": ["-fno-slp-vectorize"]}


def _stale(out: Path, deps) -> bool:
    if not out.is_file():
        return True
    t = out.stat().st_mtime
    return any(p.stat().st_mtime > t for p in deps)


def build(force: bool = False, keep_temps: bool = False) -> Path:
    OBJ.mkdir(exist_ok=True)
    flags = list(FLAGS)
    if os.environ.get("VSZIP_DEV_R"):  # development only: one BoxBlur CT radius
        flags.append("-DVSZIP_DEV_R=" + os.environ["VSZIP_DEV_R"])
        force = True
    if os.environ.get("VSZIP_EXTRA_FLAGS"):  # development only
        flags += os.environ["VSZIP_EXTRA_FLAGS"].split()
        force = True
    # a change of flags (e.g. a development build before) invalidates every object
    stamp = OBJ / "flags.txt"
    stamp_text = " ".join(flags) + " | " + repr(sorted(FILE_FLAGS.items()))
    if not stamp.is_file() or stamp.read_text() != stamp_text:
        force = True
    srcs = sorted(CSRC.glob("*.hip")) + sorted(CSRC.glob("*.cpp"))  # *.cpp: device-free host code (also built by tests/sanitize)
    hdrs = list(CSRC.glob("*.hpp")) + list(CSRC.glob("*.inc")) + list((PKG.parent / "include").glob("*.h"))
    jobs = []
    for s in srcs:
        o = OBJ / (s.stem + ".o")
        if force or _stale(o, [s] + hdrs):
            cmd = [HIPCC, *flags, *FILE_FLAGS.get(s.stem, []), "-c", str(s), "-o", str(o)]
            if keep_temps:
                cmd += ["-save-temps=obj"]
            jobs.append(cmd)

    def run(cmd):
        r = subprocess.run(cmd, capture_output=True, text=True, cwd=str(OBJ))
        if r.returncode != 0:
            raise RuntimeError("hipcc failed: " + " ".join(cmd) + "\n" + r.stdout + r.stderr)
        return r.stderr

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        for warn in ex.map(run, jobs):
            if warn.strip():
                sys.stderr.write(warn)
    stamp.write_text(stamp_text)
    objs = [OBJ / (s.stem + ".o") for s in srcs]
    if force or jobs or _stale(LIB, objs):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", str(LIB), *map(str, objs)]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
    build_plugin(force or bool(jobs))
    return LIB


PLUGIN = PKG / "libvszip.so"
FAKEVS = PKG.parent / "tests" / "fakevs" / "libfakevs.so"


def build_plugin(force: bool = False) -> None:
    """libvszip.so (the VapourSynth plugin: plain C++ over the C ABI, linked to libvszip_hip.so
    next to it) and the VapourSynth-free test host tests/fakevs/libfakevs.so."""
    psrc = PKG / "plugin" / "vszip_plugin.cpp"
    pdeps = [psrc, PKG / "plugin" / "VapourSynth4_min.h", PKG / "plugin" / "vsapi_layout_check.h", PKG.parent / "include" / "vszip_hip.h"]
    if force or _stale(PLUGIN, pdeps):
        cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall", "-o", str(PLUGIN), str(psrc),
               "-L" + str(PKG), "-lvszip_hip", "-Wl,-rpath,$ORIGIN", "-lpthread"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("plugin build failed:\n" + r.stdout + r.stderr)
    fsrc = FAKEVS.parent / "fakevs.cpp"
    if fsrc.is_file() and (force or _stale(FAKEVS, [fsrc, PKG / "plugin" / "VapourSynth4_min.h"])):
        cmd = ["g++", "-O1", "-g", "-std=c++17", "-fPIC", "-shared", "-fvisibility=hidden", "-Wall", "-o", str(FAKEVS), str(fsrc), "-ldl"]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("fakevs build failed:\n" + r.stdout + r.stderr)


if __name__ == "__main__":
    p = build(force="--force" in sys.argv, keep_temps="--keep-temps" in sys.argv)
    print(p)
