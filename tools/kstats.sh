#!/bin/bash
# GPU box: per-kernel times (rocprofv3 --kernel-trace --stats) of tools/prof_legs.py <legs...>
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ks
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ks -- python3 $GRAFT_REPO_ROOT/tools/prof_legs.py "$@" > /dev/null 2>&1
python3 - <<'PY'
import csv, glob
for f in glob.glob("/tmp/ks/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        print("%-52s calls %4s avg_us %9.1f min %9.1f" % (n, r["Calls"], float(r["AverageNs"]) / 1000, float(r["MinNs"]) / 1000))
PY
