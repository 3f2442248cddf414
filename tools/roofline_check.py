#!/usr/bin/env python3
"""Per leg: algorithmic bytes per call / (kernel time per call from the leg's own rocprofv3 --kernel-trace --stats run) / 8 TB/s,
beside the fraction the leg itself reports (tools/roofline_check.sh). basis "dominant kernel": the kernels whose name contains
`kernel_match`, average launch x launches per call; "stream" / "wall": every kernel of the process (the leg ran alone) except the
allocator's classification copy and the runtime's own copy kernels."""
import csv
import json
import sys
from pathlib import Path

d = Path(sys.argv[1])
NOT_THE_LEG = ("placement_probe_kernel", "__amd_rocclr")
TWO_STREAMS = {"eedi3": "the line kernel of the short planes runs beside the vertical-consistency chains (two streams): kernel time sums to more than the wall clock",
               "ssimulacra2_4k": "the small scales run on a second stream beside the large scales' maps kernels: kernel time sums to more than the wall clock"}
print("| leg | basis | algorithmic bytes / call | calls | kernels counted | kernel us / call (profile) | frac from the profile | frac the leg reports | profile / leg | note |")
print("|---|---|---|---|---|---|---|---|---|---|")
for jf in sorted(d.glob("*.json")):
    leg = jf.stem
    rec = None
    for line in jf.read_text().splitlines():
        if line.startswith("{"):
            try:
                j = json.loads(line)
            except json.JSONDecodeError:
                continue
            if j.get("leg") == leg:
                rec = j
    cf = d / f"{leg}.csv"
    if rec is None or not cf.exists():
        print(f"| {leg} | - | - | - | no record / no profile | - | - | - | - | |")
        continue
    rf = rec["record"].get("roofline") or {}
    alg, basis, match, calls = rf.get("alg_bytes_per_call"), rf.get("basis", "?"), rf.get("kernel_match"), rec["calls"]
    rows = [r for r in csv.DictReader(open(cf)) if not any(x in r["Name"] for x in NOT_THE_LEG)]
    dom = bool(match) and basis == "dominant kernel"
    use = [r for r in rows if (match in r["Name"] if dom else True)]
    tot_ns = sum(float(r["TotalDurationNs"]) for r in use)
    n_launch = sum(int(r["Calls"]) for r in use)
    if not alg or not calls or not tot_ns:
        print(f"| {leg} | {basis} | {alg} | {calls} | - | - | - | {rf.get('frac')} | - | |")
        continue
    if dom:
        per_call = max(1, round(n_launch / calls))
        us = tot_ns / n_launch * per_call / 1e3
    else:
        us = tot_ns / calls / 1e3
    fp = alg / (us * 1e-6) / 8e12
    fl = rf.get("frac")
    names = ", ".join(sorted({r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0] for r in use}))[:90]
    note = TWO_STREAMS.get(leg, "")
    if not note and dom and fp / fl > 1.03:
        note = f"the leg's time is HIP events around each launch: +{(us * (fp / fl) - us) / max(1, round(n_launch / calls)):.1f} us per launch over the trace's kernel duration"
    print(f"| {leg} | {basis} | {alg:,.0f} | {calls} | {names} ({n_launch} launches) | {us:.1f} | {fp:.4f} | {fl:.4f} | {fp / fl:.3f} | {note} |")
