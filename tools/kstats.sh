#!/bin/bash
# GPU box: per-kernel times (rocprofv3 --kernel-trace --stats) of any python3 command line.
# usage: bash tools/kstats.sh <name> <script> <args...>     ->  gpurun_out/<name>_kernel_stats.csv + a table on stdout
name=$1; shift
R=$GRAFT_REPO_ROOT
script=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/ksb
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/ksb -- python3 $script "$@" > $R/gpurun_out/${name}_out.json 2> /dev/null
python3 - "$R/gpurun_out/${name}_kernel_stats.csv" <<'PY'
import csv, glob, sys, shutil
for f in glob.glob("/tmp/ksb/*/*kernel_stats.csv"):
    shutil.copy(f, sys.argv[1])
    for r in csv.DictReader(open(f)):
        n = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        print("%-60s calls %5s avg_us %9.1f min %9.1f max %9.1f  %5s%%" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1000, float(r["MinNs"]) / 1000, float(r["MaxNs"]) / 1000, r["Percentage"]))
PY
