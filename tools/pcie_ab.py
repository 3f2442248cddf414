#!/usr/bin/env python3
"""Round 3: where do 40 % of the PCIe-fed BoxBlur rate go when the placement search succeeded?
One process: the PCIe-fed leg (bench.pcie_boxblur) fresh, again after the 3 x 64-arena placement search with the chosen arenas
held, again after everything is freed; the whole thing with and without the NUMA binding (child processes).
    python tools/pcie_ab.py            # parent: runs both children
"""
import json
import os
import subprocess
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))


def sysfs_link(local_rank=0):
    try:
        import torch

        pr = torch.cuda.get_device_properties(local_rank)
        bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}.0"
        d = Path(f"/sys/bus/pci/devices/{bdf}")
        rd = lambda n: (d / n).read_text().strip() if (d / n).exists() else None
        return {"bdf": bdf, "numa_node": rd("numa_node"), "current_link_speed": rd("current_link_speed"), "current_link_width": rd("current_link_width"),
                "max_link_speed": rd("max_link_speed"), "max_link_width": rd("max_link_width")}
    except Exception as e:
        return {"error": str(e)}


def child(bind: bool):
    import torch

    import bench
    import vszip_amd

    torch.cuda.set_device(0)
    node = bench.bind_to_gpu_numa(0) if bind else None
    out = {"bind": bind, "numa_node": node, "link": sysfs_link(), "cpus_allowed": len(os.sched_getaffinity(0))}
    dev = vszip_amd.Device(0)
    m = lambda tag: out.__setitem__(tag, round(bench.pcie_boxblur(vszip_amd, 0, 13)["value"], 1))
    m("fresh")
    m("fresh_again")
    t0 = time.perf_counter()
    step, keep = bench.setup_boxblur(dev, 0, 64, 13)
    out["search_s"] = round(time.perf_counter() - t0, 2)
    out["placement"] = {k: keep[2][k] for k in ("first_allocation_us", "destination_candidates_us", "source_candidates_us")}
    for _ in range(50):
        step()
    dev.sync()
    m("after_search_arenas_held")
    del step, keep
    import gc

    gc.collect()
    dev.sync()
    m("after_free")
    # plain allocations of the same size held instead of probed ones: is it the search or the memory held?
    import ctypes as C

    hold = []
    for _ in range(2):
        p = C.c_void_p()
        dev.check(dev.lib.vszip_dev_alloc(dev.ctx, 1700 << 20, C.byref(p)))
        hold.append(p.value)
    m("two_plain_arenas_held")
    for p in hold:
        dev.lib.vszip_dev_free(dev.ctx, p)
    m("end")
    dev.close()
    print("PCIE_AB " + json.dumps(out))


if __name__ == "__main__":
    if len(sys.argv) > 1:
        child(sys.argv[1] == "bind")
    else:
        for mode in ("bind", "nobind"):
            r = subprocess.run([sys.executable, __file__, mode], capture_output=True, text=True)
            for ln in r.stdout.splitlines():
                if ln.startswith("PCIE_AB"):
                    print(ln)
            if r.returncode:
                print("child failed:", r.stderr[-2000:])
