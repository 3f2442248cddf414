#!/bin/bash
# GPU box: PMC passes over any python script. usage: pmc_script.sh <script.py> "<counter set 1>" "<set 2>" ...
scr=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=/tmp/pmc_scr; rm -rf $out; mkdir -p $out
n=0
for set in "$@"; do
  n=$((n+1))
  timeout 300 rocprofv3 --pmc $set --output-format csv -d $out/set$n -- python3 $R/$scr > /dev/null 2> $out/set$n.err
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/set*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        tag = n.replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0][:64]
        agg[tag][r["Counter_Name"]].append(float(r["Counter_Value"]))
for tag, d in agg.items():
    for k, v in sorted(d.items()):
        print(f"{tag:64s} {k:24s} {sum(v)/len(v):16.1f}")
PY
