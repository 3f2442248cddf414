#!/usr/bin/env python3
"""Turn one tools/prof.sh output directory (gpurun_out/prof_<tag>) into the tracked summaries
under profiles/:  <name>_kernel_stats.csv (rocprofv3 --kernel-trace --stats, verbatim),
<name>_pmc.json (per-kernel mean counter values + HBM traffic per launch) and <name>.md.

HBM traffic per launch follows MI355X_MICROARCH.md "HBM / rocprofv3": FETCH_SIZE and WRITE_SIZE
come from separate --pmc passes; rocprofv3 reports both in KB (1024 B). On gfx950 FETCH_SIZE
tallies the 128-B requests of wide coalesced streaming reads at 64 B, so it is doubled;
WRITE_SIZE is exact for 16-B-per-lane streaming stores.
usage: summarize_prof.py gpurun_out/prof_<tag> profiles/<name>
"""
import collections
import csv
import glob
import json
import shutil
import sys
from pathlib import Path


def short(name: str) -> str:
    n = name.replace("void ", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0]


def main():
    src, dst = Path(sys.argv[1]), Path(sys.argv[2])
    dst.parent.mkdir(parents=True, exist_ok=True)
    import os

    # gpurun MERGES into gpurun_out/: an earlier call's files (other process ids) may still lie beside the new ones
    stats = sorted(glob.glob(str(src / "trace" / "*" / "*kernel_stats.csv")), key=os.path.getmtime, reverse=True)
    rows = []
    if stats:
        shutil.copy(stats[0], f"{dst}_kernel_stats.csv")
        rows = list(csv.DictReader(open(stats[0])))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    newest = {}
    for f in glob.glob(str(src / "pmc_*" / "*" / "*counter_collection.csv")):
        d = os.path.dirname(f)
        if d not in newest or os.path.getmtime(f) > os.path.getmtime(newest[d]):
            newest[d] = f
    for f in newest.values():
        for r in csv.DictReader(open(f)):
            agg[short(r["Kernel_Name"])][r["Counter_Name"]].append(float(r["Counter_Value"]))
    pmc = {}
    for k, d in agg.items():
        m = {c: sum(v) / len(v) for c, v in d.items()}
        if "FETCH_SIZE" in m and "WRITE_SIZE" in m:
            m["hbm_read_bytes_per_launch"] = m["FETCH_SIZE"] * 1024 * 2  # gfx950: FETCH_SIZE = half the bytes
            m["hbm_write_bytes_per_launch"] = m["WRITE_SIZE"] * 1024
            m["hbm_traffic_bytes_per_launch"] = m["hbm_read_bytes_per_launch"] + m["hbm_write_bytes_per_launch"]
        m["dispatches_sampled"] = max(len(v) for v in d.values())
        pmc[k] = m
    bench = None
    bj = src / "bench_trace.json"
    if bj.exists():
        lines = []
        for line in bj.read_text().splitlines():
            if line.startswith("{"):
                try:
                    lines.append(json.loads(line))
                except json.JSONDecodeError:
                    import ast

                    lines.append(ast.literal_eval(line))  # tools/prof_legs.py printed dict reprs before it printed JSON
        bench = lines[-1] if len(lines) == 1 else (lines or None)
    # HBM traffic per unit of work: sum over kernels of (traffic per launch x launches per call) / units per call. Launches per
    # call = the trace's Calls / the line's config.timed_calls (--workload runs of bench.py: one workload per process).
    per_unit = None
    if isinstance(bench, dict) and bench.get("config", {}).get("timed_calls") and not bench.get("others"):
        cfg = bench["config"]
        calls = cfg["timed_calls"]
        units = cfg.get("pairs_per_step_per_gpu") or cfg.get("frames_per_step_per_gpu")
        unit_name = "pair" if cfg.get("pairs_per_step_per_gpu") else "frame"
        if units:
            tot, parts = 0.0, {}
            for r in rows:
                k = short(r["Name"])
                if "placement_probe_kernel" in k or "__amd_rocclr" in k:  # the allocator's classification copy, the runtime's copies: not the workload
                    continue
                m = pmc.get(k)
                if m and "hbm_traffic_bytes_per_launch" in m:
                    b = m["hbm_traffic_bytes_per_launch"] * int(r["Calls"]) / calls / units
                    parts[k] = b
                    tot += b
            per_unit = {f"hbm_traffic_bytes_per_{unit_name}": tot, "by_kernel": parts, "timed_calls": calls, f"{unit_name}s_per_call": units}
    json.dump({"source": str(src), "bench_line_under_tracing": bench, "hbm_traffic_per_unit": per_unit, "kernels": pmc}, open(f"{dst}_pmc.json", "w"), indent=1)
    with open(f"{dst}.md", "w") as o:
        o.write(f"# {dst.name}: rocprofv3 summary\n\nCommand: `rocprofv3 --kernel-trace --stats -- python3 bench.py --no-cpu ...` or `... tools/prof_legs.py <legs>` (tools/prof_all.sh), PMC in separate passes.\n\n")
        o.write("| kernel | calls | avg us | min us | max us | % |\n|---|---|---|---|---|---|\n")
        for r in rows:
            o.write(f"| `{short(r['Name'])}` | {r['Calls']} | {float(r['AverageNs'])/1e3:.1f} | {float(r['MinNs'])/1e3:.1f} | {float(r['MaxNs'])/1e3:.1f} | {r['Percentage']} |\n")
        if per_unit:
            key = [k for k in per_unit if k.startswith("hbm_traffic_bytes_per_")][0]
            o.write(f"\n## HBM traffic per {key.rsplit('_', 1)[1]} (FETCH_SIZE x 2 + WRITE_SIZE per launch x launches per call / units per call)\n\n")
            o.write(f"**{per_unit[key] / 1e6:,.1f} MB** ({per_unit['timed_calls']} calls traced)\n\n")
            for k, b in sorted(per_unit["by_kernel"].items(), key=lambda kv: -kv[1]):
                o.write(f"- `{k}`: {b / 1e6:,.1f} MB\n")
        o.write("\n## PMC (mean per dispatch)\n\n")
        for k, m in pmc.items():
            o.write(f"### `{k}`\n\n")
            for c, v in sorted(m.items()):
                o.write(f"- {c}: {v:,.1f}\n")
            o.write("\n")
    print(open(f"{dst}.md").read())


if __name__ == "__main__":
    main()
