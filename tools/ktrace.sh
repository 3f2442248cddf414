#!/bin/bash
# GPU box: the kernel timeline (rocprofv3 --kernel-trace, no --stats) of any python3 command line: start offset, duration and the gap to the
# previous kernel's end, for the last N dispatches.   usage: bash tools/ktrace.sh <N> <script> <args...>
n=$1; shift
R=$GRAFT_REPO_ROOT
script=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ktr
timeout 600 rocprofv3 --kernel-trace --output-format csv -d /tmp/ktr -- python3 $script "$@" > /dev/null 2> /dev/null
python3 - $n <<'PY'
import csv, glob, sys
rows = []
for f in glob.glob("/tmp/ktr/*/*kernel_trace.csv"):
    rows += list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-int(sys.argv[1]):]
t0 = int(rows[0]["Start_Timestamp"]); prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:44]
    print("%-44s start %9.1f us  dur %8.1f us  gap %8.1f us" % (name, (s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0))
    prev = e
PY
