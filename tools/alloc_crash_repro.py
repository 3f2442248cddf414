import ctypes as C, sys
sys.path.insert(0, "/root/repo")
import torch, vszip_amd
dev = vszip_amd.Device(0)
dev.set_option("VSZIP_PLACEMENT_TRIES", int(sys.argv[1]) if len(sys.argv) > 1 else 3)
ps = []
for i in range(6):
    p = C.c_void_p()
    dev.check(dev.lib.vszip_dev_alloc(dev.ctx, 1600 << 20, C.byref(p)))
    print(i, hex(p.value), dev.arena_info(p.value), flush=True)
    ps.append(p.value)
for p in ps:
    dev.check(dev.lib.vszip_dev_free(dev.ctx, p))
print("ok")
