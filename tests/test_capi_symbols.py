"""CPU-side check: the C-ABI library loads and exports every symbol that
include/vszip_hip.h declares (no compute calls; no GPU needed)."""
import re
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]


def test_library_exports_every_declared_symbol():
    import vszip_amd

    lib = vszip_amd.capi.load()
    header = (ROOT / "include" / "vszip_hip.h").read_text()
    declared = set(re.findall(r"\b(vszip_[a-z0-9_]+)\s*\(", header))
    declared -= {"vszip_ctx", "vszip_plane"}
    assert declared, "no declarations parsed"
    for name in sorted(declared):
        assert hasattr(lib, name), f"libvszip_hip.so does not export {name}"
    assert declared == set(vszip_amd.capi.SYMBOLS), (declared ^ set(vszip_amd.capi.SYMBOLS))
    assert lib.vszip_abi_version() == 4


def test_no_gpu_fails_loudly():
    """Without a GPU the context refuses to come up; nothing falls back to the CPU."""
    import pytest
    import vszip_amd

    lib = vszip_amd.capi.load()
    import ctypes as C

    ctx = C.c_void_p()
    rc = lib.vszip_ctx_create(0, C.byref(ctx))
    if rc == 0:  # a GPU is present (GPU box): fine, clean up
        lib.vszip_ctx_destroy(ctx)
        pytest.skip("GPU present")
    assert rc < 0 and not ctx.value


def test_resample_table_equals_the_oracle_restatement():
    """vszip_resample_table (device-free host code of the product: the tables the YUV colour pre-stage of
    SSIMULACRA2 runs on) against oracle/vs_host.py::zimg_filter, the restatement the reference's YUV goldens pin:
    same first tap and bit-identical f32 coefficients for every geometry the pre-stage meets (4:2:0 / 4:2:2 /
    4:1:0, every chroma siting, odd and tiny sizes)."""
    import numpy as np

    import vszip_amd
    from oracle import vs_host as vh

    for src, dst, shift in [(320, 640, 0.25), (160, 320, 0.0), (319, 638, 0.25), (159, 318, 0.0), (6, 12, 0.25), (3, 6, 0.0), (1, 2, 0.0), (2, 4, 0.25),
                            (960, 1920, 0.0), (540, 1080, 0.25), (540, 1080, -0.25), (160, 640, 0.375), (5, 10, 0.0), (7, 13, 0.25), (4, 7, 0.0)]:
        left, coef = vszip_amd.capi.resample_table(src, dst, shift)
        l0, c0 = vh.zimg_filter("bicubic", src, dst, shift)
        c0 = c0.astype(np.float32)
        w = c0.shape[1]
        assert w <= 4
        assert np.array_equal(left, l0.astype(np.int32)), (src, dst, shift)
        assert np.array_equal(coef[:, :w].view(np.uint32), c0.view(np.uint32)), (src, dst, shift)
        assert not coef[:, w:].any()

def test_resample_tables_as_the_420_prestage_pass_assumes():
    """What ssim_yuv420_rgb_kernel's period path and ssim_yuv420_plan rely on, checked on the product's own tables for 2x enlargement under
    the sitings VapourSynth has (shift 0: centred chroma, +-0.25 source samples: co-sited):
      * co-sited tables are NOT monotonic in their first tap (a sample on a chroma sample has one non-zero tap, its neighbour four that start
        one sample earlier) - the round-5 bug: a tile must take the smallest first tap of its rows, not its first row's;
      * away from the ends the table has period 4 in the destination (first tap = (x >> 1) + d[x & 3], coefficients by x & 3): the interior path
        keeps four taps-and-coefficients sets in the kernel argument;
      * the four first taps of a group differ by at most 2 source samples, a 16-row tile taps at most 12 source rows and a 4-row block at
        most 6 (the LDS slice, the dword windows and the filtered rows a thread holds are sized by these)."""
    import numpy as np

    import vszip_amd

    for src, dst in [(960, 1920), (540, 1080), (1920, 3840), (1080, 2160), (160, 320)]:
        for shift in (0.25, 0.0, -0.25):
            left, coef = vszip_amd.capi.resample_table(src, dst, shift)
            if shift != 0.0:  # (+-0.25 source samples: a destination sample sits ON every source sample)
                assert (np.diff(left) < 0).any(), (src, dst, "co-sited tables step back")
            ref = (dst // 2 // 4) * 4
            d = [int(left[ref + i]) - (ref >> 1) for i in range(4)]
            regular = np.array([left[x] == ((x - (x & 3)) >> 1) + d[x & 3] and np.array_equal(coef[x].view(np.uint32), coef[ref + (x & 3)].view(np.uint32)) and left[x] + 3 <= src - 1
                                for x in range(dst)])
            first = last = ref  # the run of regular samples around the reference group
            while first > 0 and regular[first - 1]:
                first -= 1
            while last < dst and regular[last]:
                last += 1
            assert first <= 8 and dst - last <= 8, (src, dst, shift, first, last)  # only the ends leave the period
            if shift >= 0.0:
                assert max(d) - min(d) <= 2, (src, dst, shift, d)
            else:  # bottom-sited chroma (shift -0.25, vertical axis only): the group spans 3 and ssim_yuv420_plan sends the clip to the fused tile kernel
                assert max(d) - min(d) == 3, (src, dst, shift, d)
                continue
            right = np.minimum(left + 3, src - 1)
            for t0 in range(0, dst, 16):
                assert right[t0:t0 + 16].max() - left[t0:t0 + 16].min() + 1 <= 12, (src, dst, shift, t0)
            for t0 in range(0, dst, 4):
                assert right[t0:t0 + 4].max() - left[t0:t0 + 4].min() + 1 <= 6, (src, dst, shift, t0)



def test_the_environment_is_read_in_one_place():
    """every switch is parsed once, in vszip_ctx_create (ctx.hip, from csrc/options.inc); no dispatch calls getenv."""
    import re
    from pathlib import Path

    csrc = Path(__file__).resolve().parents[1] / "vapoursynth-zip_amd" / "csrc"
    hits = {}
    for f in sorted(csrc.glob("*")):
        if f.suffix in (".hip", ".hpp", ".cpp", ".inc", ".h"):
            n = len(re.findall(r"\bgetenv\s*\(", f.read_text()))
            if n:
                hits[f.name] = n
    assert hits == {"ctx.hip": 1}, hits
    # every option line names a VSZIP_ variable exactly once
    names = re.findall(r'VSZIP_(?:DEV_)?OPT\(\w+, "(VSZIP_[A-Z0-9_]+)"', (csrc / "options.inc").read_text())
    assert len(names) == len(set(names)) >= 40
    # the default build does not define the development variants
    build = (csrc.parent / "build.py").read_text()
    assert "VSZIP_DEV_VARIANTS" not in build.split("FLAGS = [")[1].split("]")[0]
