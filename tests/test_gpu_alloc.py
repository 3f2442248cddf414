"""vszip_dev_alloc / vszip_dev_alloc_probed (include/vszip_hip.h): placed and placement-probed device allocations, context options."""
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


def test_probe_sees_every_candidate_and_the_cheapest_is_kept(dev):
    import torch

    nbytes = 64 << 20
    dev.sync()
    free0 = torch.cuda.mem_get_info(0)[0]
    seen = []
    costs = iter([5.0, 3.0, 9.0, 1.5, 7.0, 1.5, 8.0])

    def probe(ptr):
        seen.append(ptr)
        return next(costs)

    ptr, best, all_costs = dev.alloc_probed(nbytes, 7, probe)
    assert len(seen) == 7 and len(set(seen)) == 7  # seven distinct allocations, all alive at once
    assert all_costs == [5.0, 3.0, 9.0, 1.5, 7.0, 1.5, 8.0] and best == 1.5 and ptr == seen[3]  # the first of equal costs
    held = free0 - torch.cuda.mem_get_info(0)[0]
    if ALONE:
        assert nbytes <= held < 2 * nbytes  # the six losers are back with the device
    # the winner is ordinary device memory
    a = np.arange(1 << 20, dtype=np.uint16).reshape(1024, 1024)
    dev.check(dev.lib.vszip_copy_h2d_2d(dev.ctx, ptr, 2048, a.ctypes.data, 2048, 2048, 1024))
    b = np.empty_like(a)
    dev.check(dev.lib.vszip_copy_d2h_2d(dev.ctx, b.ctypes.data, 2048, ptr, 2048, 2048, 1024))
    dev.sync()
    assert np.array_equal(a, b)
    dev.check(dev.lib.vszip_dev_free(dev.ctx, ptr))
    if ALONE:
        assert free0 - torch.cuda.mem_get_info(0)[0] < (8 << 20)


def test_plain_allocation_without_a_probe(dev):
    p, best = C.c_void_p(), C.c_double(-1.0)
    dev.check(dev.lib.vszip_dev_alloc_probed(dev.ctx, 1 << 20, 1, None, None, C.byref(p), C.byref(best)))
    assert p.value and best.value == 0.0
    dev.check(dev.lib.vszip_dev_free(dev.ctx, p.value))
    dev.check(dev.lib.vszip_dev_alloc_probed(dev.ctx, 1 << 20, 8, None, None, C.byref(p), None))
    dev.check(dev.lib.vszip_dev_free(dev.ctx, p.value))
    assert dev.lib.vszip_dev_alloc_probed(dev.ctx, 1 << 20, 4, None, None, None, None) != 0


# ---- round 4: vszip_dev_alloc places large requests itself (ctx.hip "placed allocations") ------------------------------------------
def _alloc(dev, nbytes):
    p = C.c_void_p()
    dev.check(dev.lib.vszip_dev_alloc(dev.ctx, nbytes, C.byref(p)))
    return p.value


@pytest.fixture()
def pdev():
    """a context of its own (the allocator's state — parked regions, the exhausted flag — is per context), small thresholds"""
    import vszip_amd

    d = vszip_amd.Device(0)
    d.set_option("VSZIP_PLACEMENT", 1)
    d.set_option("VSZIP_PLACEMENT_MIN_MIB", 128)
    d.set_option("VSZIP_PLACEMENT_WALK_GIB", 2)
    d.set_option("VSZIP_PLACEMENT_WALK_MS", 4000)
    yield d
    d.close()


def test_small_requests_are_plain_and_large_ones_are_classified(pdev):
    small = _alloc(pdev, 8 << 20)
    assert pdev.placement_info(small)["bytes_per_second"] == 0.0 and pdev.placement_info()["walks"] == 0
    big = _alloc(pdev, 200 << 20)  # rounded to 256 MiB: a walk of at most 2 GiB / 256 MiB = 8 candidates
    info = pdev.placement_info(big)
    assert info["walks"] == 1 and 1 <= info["probed"] <= 8
    assert 2e12 < info["bytes_per_second"] < 8e12  # the classification copy's rate on the region that was kept
    # it is ordinary device memory
    a = np.arange(1 << 20, dtype=np.uint16).reshape(1024, 1024)
    pdev.check(pdev.lib.vszip_copy_h2d_2d(pdev.ctx, big, 2048, a.ctypes.data, 2048, 2048, 1024))
    b = np.empty_like(a)
    pdev.check(pdev.lib.vszip_copy_d2h_2d(pdev.ctx, b.ctypes.data, 2048, big, 2048, 2048, 1024))
    pdev.sync()
    assert np.array_equal(a, b)
    # the probe entry point on a caller's region, alone and as a pair
    r1 = pdev.probe_region(big, 200 << 20)
    other = _alloc(pdev, 200 << 20)
    r2 = pdev.probe_region(other, 200 << 20, big)
    assert 2e12 < r1 < 8e12 and 2e12 < r2 < 8e12
    for p in (small, big, other):
        pdev.check(pdev.lib.vszip_dev_free(pdev.ctx, p))


def test_freed_regions_are_parked_reused_and_trimmed(pdev):
    import torch

    free0 = torch.cuda.mem_get_info(0)[0]
    a = _alloc(pdev, 256 << 20)
    walks = pdev.placement_info()["walks"]
    pdev.check(pdev.lib.vszip_dev_free(pdev.ctx, a))
    info = pdev.placement_info()
    assert info["parked_regions"] >= 1 and info["parked_bytes"] >= 256 << 20  # parked, not freed
    if ALONE:
        assert free0 - torch.cuda.mem_get_info(0)[0] >= 256 << 20
    b = _alloc(pdev, 250 << 20)  # the same granule: the parked region serves it when it is of the fast class, else it competes in a new walk
    after = pdev.placement_info()
    assert after["walks"] in (walks, walks + 1)
    if after["walks"] == walks:
        assert b == a
    pdev.check(pdev.lib.vszip_dev_free(pdev.ctx, b))
    assert pdev.trim() >= 256 << 20
    assert pdev.placement_info()["parked_regions"] == 0
    pdev.sync()
    if ALONE:
        assert free0 - torch.cuda.mem_get_info(0)[0] < (64 << 20)


def test_parked_memory_is_bounded_and_returns_when_an_allocation_needs_it(pdev):
    pdev.set_option("VSZIP_PLACEMENT_PARK_GIB", 1)
    ptrs = [_alloc(pdev, 512 << 20) for _ in range(4)]
    for p in ptrs:
        pdev.check(pdev.lib.vszip_dev_free(pdev.ctx, p))
    assert pdev.placement_info()["parked_bytes"] <= 1 << 30  # the cap: the slowest parked regions went back to the driver
    # placement off: plain hipMalloc / hipFree, nothing parked
    pdev.trim()
    pdev.set_option("VSZIP_PLACEMENT", 0)
    p = _alloc(pdev, 512 << 20)
    assert pdev.placement_info(p)["bytes_per_second"] == 0.0
    pdev.check(pdev.lib.vszip_dev_free(pdev.ctx, p))
    assert pdev.placement_info()["parked_regions"] == 0


def test_a_walk_without_a_fast_region_is_the_contexts_last(pdev):
    """a budget that admits one candidate: whether that one is fast or not, the next request must not walk again when it was not"""
    pdev.set_option("VSZIP_PLACEMENT_WALK_GIB", 1)
    a = _alloc(pdev, 900 << 20)
    i1 = pdev.placement_info(a)
    assert i1["walks"] == 1 and i1["probed"] == 1
    b = _alloc(pdev, 900 << 20)
    i2 = pdev.placement_info(b)
    if i1["exhausted"]:
        assert i2["walks"] == 1 and i2["bytes_per_second"] == 0.0  # plain allocation, no second search
    else:
        assert i2["walks"] == 2
    for p in (a, b):
        pdev.check(pdev.lib.vszip_dev_free(pdev.ctx, p))


def test_options_by_name(pdev):
    import vszip_amd

    assert pdev.get_option("VSZIP_PLACEMENT_MIN_MIB") == 128
    with pdev.options(VSZIP_RT_NO_ICHAIN=1):
        assert pdev.get_option("VSZIP_RT_NO_ICHAIN") == 1
    assert pdev.get_option("VSZIP_RT_NO_ICHAIN") == 0
    with pytest.raises(vszip_amd.VszipError, match="unknown option"):
        pdev.set_option("VSZIP_NO_SUCH_THING", 1)
    try:
        pdev.get_option("VSZIP_RT_FUSED")
    except vszip_amd.VszipError as e:  # the default build: a development variant
        assert e.code == -3 and "development variant" in str(e)
