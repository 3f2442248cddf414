"""CPU oracle vs the reference's own goldens and identities (BoxBlur).

Golden keys come from the reference's tests/goldens/boxblur.json (cases declared
in tests/test_boxblur.py:13-49); identities from tests/test_boxblur.py:86-101.
"""
import numpy as np
import pytest

import fixtures as fx

REL = 1e-6  # reference tests/golden.py:173


def _check(stats: dict, gold: dict, rel=REL):
    for k in ("avg", "min", "max"):
        assert stats[k] == pytest.approx(gold[k], rel=rel, abs=1e-9), (k, stats[k], gold[k])


def test_golden_rgbs_ct_r2(oracle):
    g = fx.ref_goldens()["exact"]["boxblur"]["RGBS|full|hradius=2,vradius=2"]
    for p in range(3):
        out = oracle.boxblur(np.ascontiguousarray(fx.crop_rgbs()[p]), 2, 1, 2, 1)
        _check(fx.plane_stats(out), g[f"p{p}"])


def test_golden_rgbs_rt_multipass(oracle):
    g = fx.ref_goldens()["exact"]["boxblur"]["RGBS|full|hpasses=2,hradius=6,vpasses=3,vradius=3"]
    for p in range(3):
        out = oracle.boxblur(np.ascontiguousarray(fx.crop_rgbs()[p]), 6, 2, 3, 3)
        _check(fx.plane_stats(out), g[f"p{p}"])


def test_golden_gray8_ct_r2(oracle):
    g = fx.ref_goldens()["exact"]["boxblur"]["GRAY8|full|hradius=2,vradius=2"]
    out = oracle.boxblur(np.ascontiguousarray(fx.crop_gray8()), 2, 1, 2, 1)
    _check(fx.plane_stats(out), g["p0"])


SOFT = {
    "GRAY16|full|hradius=1,vradius=1": (1, 1, 1, 1),
    "GRAY16|full|hradius=2,vradius=2": (2, 1, 2, 1),
    "GRAY16|full|hradius=8,vradius=8": (8, 1, 8, 1),
    "GRAY16|full|hradius=22,vradius=22": (22, 1, 22, 1),
    "GRAY16|full|hradius=23,vradius=23": (23, 1, 23, 1),
    "GRAY16|full|hradius=40,vradius=40": (40, 1, 40, 1),
    "GRAY16|full|hradius=4,vradius=9": (4, 1, 9, 1),
    "GRAY16|full|hradius=9,vradius=4": (9, 1, 4, 1),
    "GRAY16|full|hpasses=3,hradius=5,vpasses=3,vradius=5": (5, 3, 5, 3),
    "GRAY16|full|hpasses=1,hradius=5,vpasses=2,vradius=5": (5, 1, 5, 2),
    "GRAY16|full|hpasses=2,hradius=5,vpasses=1,vradius=5": (5, 2, 5, 1),
    "GRAY16|full|hpasses=0,hradius=0,vradius=7": (0, 0, 7, 1),
    "GRAY16|full|hradius=7,vpasses=0,vradius=0": (7, 1, 0, 0),
}


@pytest.mark.parametrize("key", sorted(SOFT))
def test_soft_gray16(oracle, key):
    """Approximate GRAY16 fixture (a few pixels differ from zimg by 1 LSB):
    min/max within 1 LSB, avg to 1e-7 (the golden itself is rel 1e-6)."""
    hr, hp, vr, vp = SOFT[key]
    g = fx.ref_goldens()["soft"]["boxblur"][key]["p0"]
    out = oracle.boxblur(np.ascontiguousarray(fx.crop_gray16()), hr, hp, vr, vp)
    st = fx.plane_stats(out)
    assert st["avg"] == pytest.approx(g["avg"], rel=1e-7)
    assert abs(st["min"] - g["min"]) <= 1 and abs(st["max"] - g["max"]) <= 1


@pytest.mark.parametrize("dtype", [np.uint8, np.uint16, np.float32, np.float16])
def test_identity_pass_composition(oracle, dtype):
    """BoxBlur(h=7, passes=2) == BoxBlur(h=7) o BoxBlur(h=7) and BoxBlur(4,9) == v9 o h4
    bit-exactly (reference tests/test_boxblur.py:86-101)."""
    src = fx.splitmix64_plane(11, (45, 67), dtype)
    two = oracle.boxblur(src, 7, 2, 0, 0)
    one = oracle.boxblur(oracle.boxblur(src, 7, 1, 0, 0), 7, 1, 0, 0)
    assert np.array_equal(two.view(np.uint8), one.view(np.uint8))
    hv = oracle.boxblur(src, 4, 1, 9, 1)
    sep = oracle.boxblur(oracle.boxblur(src, 4, 1, 0, 0), 0, 0, 9, 1)
    assert np.array_equal(hv.view(np.uint8), sep.view(np.uint8))


def test_stride_independence(oracle):
    src = fx.splitmix64_plane(5, (40, 96), np.uint16)
    view = src[:, 7:71]  # stride > width, offset base pointer
    a = oracle.boxblur(view, 13, 1, 13, 1)
    b = oracle.boxblur(np.ascontiguousarray(view), 13, 1, 13, 1)
    assert np.array_equal(a, b)


def test_ct_int_closed_forms():
    """The integer identities the HIP CT kernel relies on, checked exhaustively:
      (col*inv + 2^31) >> 32 == (col + r) // k == mulhi(col + r, ceil(2^32/k))
    for every reachable column sum (boxblur_comptime.zig:28,114-128)."""
    for r in range(1, 23):
        k = 2 * r + 1
        inv = ((1 << 32) + r) // k
        col = np.arange(0, k * 65535 + 1, dtype=np.uint64)
        ref = (col * np.uint64(inv) + np.uint64(1 << 31)) >> np.uint64(32)
        assert np.array_equal(ref, (col + np.uint64(r)) // np.uint64(k)), r
        m = -(-(1 << 32) // k)
        assert np.array_equal(ref, ((col + np.uint64(r)) * np.uint64(m)) >> np.uint64(32)), r
        # a 24-bit reciprocal (full-rate v_mul_u32_u24 pair) is exact too; measured slower than mulhi on MI355X, not used
        sh = 0
        while (1 << sh) <= (65535 * k + r) * k:
            sh += 1
        m24 = -(-(1 << sh) // k)
        assert m24 < (1 << 24) and sh < 32
        assert np.array_equal(ref, ((col + np.uint64(r)) * np.uint64(m24)) >> np.uint64(sh)), r


def test_ct_hblur_closed_form(oracle):
    """running 16.16 sum == (inv2*E_x + 32768 + ((E_0*invlo) >> 16)) >> 16 with E_x the
    edge-duplicating mirrored window sum (boxblur_comptime.zig:130-159): compare a
    numpy evaluation of the closed form with the oracle's sequential running sum on
    a plane whose vertical pass is the identity (constant columns)."""
    rng = np.random.default_rng(3)
    for r in (1, 5, 13, 22):
        k = 2 * r + 1
        w = 200
        row = rng.integers(0, 65536, size=w, dtype=np.uint16)
        plane = np.repeat(row[None, :], 2 * r + 5, axis=0)
        out = oracle.boxblur(plane, r, 1, r, 1)[r + 2]
        inv = ((1 << 32) + r) // k
        inv2, invlo = inv >> 16, inv & 0xFFFF
        # vertical pass on constant columns: (k*v*inv + 2^31) >> 32 == v
        idx = np.arange(-r, w + r)
        idx = np.where(idx < 0, -idx - 1, idx)
        idx = np.where(idx >= w, 2 * w - 1 - idx, idx)
        padded = row[idx].astype(np.int64)
        c = np.concatenate([[0], np.cumsum(padded)])
        e = c[k:] - c[:-k]
        expect = (inv2 * e + 32768 + ((e[0] * invlo) >> 16)) >> 16
        assert np.array_equal(out.astype(np.int64), expect), r
