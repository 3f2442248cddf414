"""vszip_dev_alloc_probed (include/vszip_hip.h): placement-probed device allocations."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


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
    assert nbytes <= held < 2 * nbytes  # the six losers are back with the device
    # the winner is ordinary device memory
    a = np.arange(1 << 20, dtype=np.uint16).reshape(1024, 1024)
    dev.check(dev.lib.vszip_copy_h2d_2d(dev.ctx, ptr, 2048, a.ctypes.data, 2048, 2048, 1024))
    b = np.empty_like(a)
    dev.check(dev.lib.vszip_copy_d2h_2d(dev.ctx, b.ctypes.data, 2048, ptr, 2048, 2048, 1024))
    dev.sync()
    assert np.array_equal(a, b)
    dev.check(dev.lib.vszip_dev_free(dev.ctx, ptr))
    assert free0 - torch.cuda.mem_get_info(0)[0] < (8 << 20)


def test_plain_allocation_without_a_probe(dev):
    p, best = C.c_void_p(), C.c_double(-1.0)
    dev.check(dev.lib.vszip_dev_alloc_probed(dev.ctx, 1 << 20, 1, None, None, C.byref(p), C.byref(best)))
    assert p.value and best.value == 0.0
    dev.check(dev.lib.vszip_dev_free(dev.ctx, p.value))
    dev.check(dev.lib.vszip_dev_alloc_probed(dev.ctx, 1 << 20, 8, None, None, C.byref(p), None))
    dev.check(dev.lib.vszip_dev_free(dev.ctx, p.value))
    assert dev.lib.vszip_dev_alloc_probed(dev.ctx, 1 << 20, 4, None, None, None, None) != 0
