"""vszip_dev_alloc (include/vszip_hip.h): large requests are placed (the fastest of a few probed candidates), small ones are plain; context options."""
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
# the device's free memory moves under a test's feet when other xdist workers share the GPU: those assertions hold for a serial run (the driver's)
ALONE = "PYTEST_XDIST_WORKER" not in os.environ


@pytest.fixture(scope="module")
def dev():
    import vszip_amd

    d = vszip_amd.Device(0)
    yield d
    d.close()


def _alloc(dev, nbytes):
    p = C.c_void_p()
    dev.check(dev.lib.vszip_dev_alloc(dev.ctx, nbytes, C.byref(p)))
    return p.value


def _roundtrip(dev, ptr, rows, cols, pitch=None, seed=0):
    """rows x cols of u16 through vszip_copy_h2d_2d / _d2h_2d at `ptr` with a device pitch in bytes"""
    pitch = pitch or cols * 2
    a = np.random.default_rng(seed).integers(0, 65536, (rows, cols), dtype=np.uint16)
    dev.check(dev.lib.vszip_copy_h2d_2d(dev.ctx, ptr, pitch, a.ctypes.data, cols * 2, cols * 2, rows))
    b = np.zeros_like(a)
    dev.check(dev.lib.vszip_copy_d2h_2d(dev.ctx, b.ctypes.data, cols * 2, ptr, pitch, cols * 2, rows))
    dev.sync()
    return np.array_equal(a, b)


# ---- vszip_dev_alloc places large requests: the fastest of a few probed candidates, nothing kept (ctx.hip "placed allocations") ----------
@pytest.fixture()
def pdev():
    """a context of its own with small thresholds: requests of 128 MiB and more are placed, three candidates at most"""
    import vszip_amd

    d = vszip_amd.Device(0)
    d.set_option("VSZIP_PLACEMENT", 1)
    d.set_option("VSZIP_PLACEMENT_MIN_MIB", 128)
    d.set_option("VSZIP_PLACEMENT_TRIES", 3)
    yield d
    d.close()


def test_small_requests_are_plain_and_large_ones_are_classified(pdev):
    small = _alloc(pdev, 8 << 20)
    assert pdev.arena_info(small)["candidates"] == 0
    big = _alloc(pdev, 200 << 20)
    info = pdev.arena_info(big)
    assert 1 <= info["candidates"] <= 3 and info["search_ms"] > 0
    assert 2e12 < info["probe_bytes_per_second"] < 8e12  # the classification copy's rate on the candidate that was kept
    # it is ordinary device memory
    assert _roundtrip(pdev, big + (60 << 20), 2160, 3840 * 4, seed=1)
    assert _roundtrip(pdev, big + (1 << 20) + 512, 1080, 1920, pitch=4096, seed=2)
    # the diagnostic probe on a caller's region, alone and as a pair
    r1 = pdev.probe_region(big, 200 << 20)
    other = _alloc(pdev, 200 << 20)
    r2 = pdev.probe_region(other, 200 << 20, big)
    assert 2e12 < r1 < 8e12 and 2e12 < r2 < 8e12
    for p in (small, big, other):
        pdev.check(pdev.lib.vszip_dev_free(pdev.ctx, p))
    assert pdev.arena_info(big)["candidates"] == 0


def test_everything_goes_back_to_the_driver(pdev):
    """nothing is cached: the losing candidates are gone when vszip_dev_alloc returns, the allocation when vszip_dev_free does"""
    import torch

    pdev.sync()
    free0 = torch.cuda.mem_get_info(0)[0]
    a = _alloc(pdev, 256 << 20)
    held = free0 - torch.cuda.mem_get_info(0)[0]
    if ALONE:
        assert (256 << 20) <= held < (256 << 20) + (64 << 20), held  # the request, not three candidates
    with pdev.options(VSZIP_PLACEMENT_TRIES=1):  # one candidate: not probed
        b = _alloc(pdev, 700 << 20)
    ib = pdev.arena_info(b)
    assert ib["candidates"] == 1 and ib["probe_bytes_per_second"] == 0.0
    pdev.check(pdev.lib.vszip_dev_free(pdev.ctx, a))
    pdev.check(pdev.lib.vszip_dev_free(pdev.ctx, b))
    pdev.check(pdev.lib.vszip_dev_free(pdev.ctx, None))  # NULL is accepted
    if ALONE:
        assert free0 - torch.cuda.mem_get_info(0)[0] < (8 << 20)


def test_the_search_stops_at_its_wall_clock_budget(pdev):
    """no candidate is started once VSZIP_PLACEMENT_BUDGET_MS have passed — with a budget of 0 the first candidate is the only one"""
    with pdev.options(VSZIP_PLACEMENT_BUDGET_MS=0, VSZIP_PLACEMENT_TRIES=24):
        a = _alloc(pdev, 200 << 20)
    ia = pdev.arena_info(a)
    assert ia["candidates"] == 1 and ia["search_ms"] < 500
    with pdev.options(VSZIP_PLACEMENT_BUDGET_MS=60000, VSZIP_PLACEMENT_TRIES=6):
        b = _alloc(pdev, 200 << 20)
    ib = pdev.arena_info(b)
    assert 1 <= ib["candidates"] <= 6
    assert pdev.get_option("VSZIP_PLACEMENT_BUDGET_MS") == 300
    for p in (a, b):
        pdev.check(pdev.lib.vszip_dev_free(pdev.ctx, p))


def test_any_context_frees_any_allocation_and_placement_off_is_plain(pdev, dev):
    a = _alloc(pdev, 256 << 20)
    assert dev.arena_info(a)["candidates"] >= 1  # the record is the process's, not the context's
    dev.check(dev.lib.vszip_dev_free(dev.ctx, a))
    assert pdev.arena_info(a)["candidates"] == 0
    with pdev.options(VSZIP_PLACEMENT=0):
        p = _alloc(pdev, 256 << 20)
        assert pdev.arena_info(p)["candidates"] == 0
    pdev.check(pdev.lib.vszip_dev_free(pdev.ctx, p))


def test_a_request_the_device_cannot_hold_fails_cleanly(pdev):
    import torch
    import vszip_amd

    pdev.sync()
    free0 = torch.cuda.mem_get_info(0)[0]
    with pytest.raises(vszip_amd.VszipError) as e:
        _alloc(pdev, free0 + (1 << 30))
    assert e.value.code == -4  # VSZIP_ERR_NOMEM
    if ALONE:
        assert free0 - torch.cuda.mem_get_info(0)[0] < (8 << 20)
    # a request of more than a quarter of the free memory is served from one candidate
    big = _alloc(pdev, int(free0 * 0.3))
    assert pdev.arena_info(big)["candidates"] == 1
    pdev.check(pdev.lib.vszip_dev_free(pdev.ctx, big))


def test_options_by_name(pdev):
    import vszip_amd

    assert pdev.get_option("VSZIP_PLACEMENT_MIN_MIB") == 128
    with pdev.options(VSZIP_RT_NO_ICHAIN=1):
        assert pdev.get_option("VSZIP_RT_NO_ICHAIN") == 1
    assert pdev.get_option("VSZIP_RT_NO_ICHAIN") == 0
    with pytest.raises(vszip_amd.VszipError, match="unknown option"):
        pdev.set_option("VSZIP_NO_SUCH_THING", 1)
    # values are range-checked: flags take 0 / 1, integers their documented range; a refused value changes nothing
    for name, bad in (("VSZIP_RT_NO_ICHAIN", 2), ("VSZIP_STAGING", 2), ("VSZIP_SCAN_MODE", 3), ("VSZIP_PLACEMENT_TRIES", 0), ("VSZIP_PLACEMENT_MIN_MIB", 1), ("VSZIP_PLACEMENT", -1)):
        before = pdev.get_option(name)
        with pytest.raises(vszip_amd.VszipError, match="out of range") as e:
            pdev.set_option(name, bad)
        assert e.value.code == -1 and pdev.get_option(name) == before
    pdev.set_option("VSZIP_SCAN_MODE", 2)
    assert pdev.get_option("VSZIP_SCAN_MODE") == 2
    pdev.set_option("VSZIP_SCAN_MODE", 0)
    try:
        pdev.get_option("VSZIP_RT_FUSED")
    except vszip_amd.VszipError as e:  # the default build: a development variant
        assert e.code == -3 and "development variant" in str(e)


def test_environment_values_are_validated_too():
    """the environment is read once per context: an out-of-range or non-numeric value where a number is due keeps the default, VSZIP_STAGING keeps its
    historical spellings ("1", "pinned"; any other text leaves it off)"""
    import subprocess
    import sys

    code = (
        "import sys; sys.path.insert(0, %r); import torch, vszip_amd; d = vszip_amd.Device(0); "
        "print(d.get_option('VSZIP_STAGING'), d.get_option('VSZIP_SCAN_MODE'), d.get_option('VSZIP_PLACEMENT_TRIES'), d.get_option('VSZIP_RT_NO_ICHAIN'), d.get_option('VSZIP_PLACEMENT_MIN_MIB'))"
    ) % str(__import__("pathlib").Path(__file__).resolve().parents[1])
    env = dict(os.environ, VSZIP_STAGING="yes", VSZIP_SCAN_MODE="7", VSZIP_PLACEMENT_TRIES="many", VSZIP_RT_NO_ICHAIN="on", VSZIP_PLACEMENT_MIN_MIB="512")
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    assert r.stdout.split()[-5:] == ["0", "0", "24", "1", "512"], r.stdout
    env["VSZIP_STAGING"] = "pinned"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=300)
    assert r.stdout.split()[-5] == "1", r.stdout
