import sys, os
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import bench, vszip_amd
dev = vszip_amd.Device(0)
timed = bench.Timed(dev, lambda: None)
for rnd in range(2):
    for env in ("", "15"):
        if env: os.environ["VSZIP_RT_VRING_MAXR"] = env
        else: os.environ.pop("VSZIP_RT_VRING_MAXR", None)
        o = bench.boxblur_1080p_5pass_leg(dev, timed, True)
        o = {"x": o} if "value" in o else o; k = list(o.keys())[0]
        o2 = bench.boxblur_other_paths_leg(dev, timed)
        print("ring up to r=15" if env else "ring up to r=8 ", round(o[k]["value"]), {a: round(b["value"]) for a, b in o2.items() if "rt_r" in a}, flush=True)
