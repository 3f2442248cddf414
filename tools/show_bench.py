#!/usr/bin/env python3
"""Print the one JSON line of bench.py in readable form.  usage: show_bench.py <file>"""
import json
import sys

d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
print(d["metric"], round(d["value"]), d["unit"], "ms/step", round(d["ms_per_step"], 3), "n_gpus", d["n_gpus"])
print("roofline", {k: (round(v, 4) if isinstance(v, float) else v) for k, v in d["roofline"].items()})
print("cpu_baseline", d.get("cpu_baseline"))
for k, v in (d.get("others") or d["config"].get("others", {})).items():
    r = v.get("roofline") or {}
    print("  %-34s %11.1f %-10s frac %s" % (k, v["value"], v["unit"], round(r.get("frac", 0), 3) if r else None))
print({k: v for k, v in d["config"].items() if k != "others"})
