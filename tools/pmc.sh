#!/bin/bash
# GPU box: PMC passes for the current build. usage: pmc.sh <tag> "<counters set 1>" "<set 2>" ...  (BENCH_ARGS for bench flags)
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/pmc_$tag
rm -rf $out; mkdir -p $out
n=0
for set in "$@"; do
  n=$((n+1))
  rocprofv3 --pmc $set --output-format csv -d $out/set$n -- python3 $R/bench.py --no-cpu --no-others --steps 4 --warmup 1 $BENCH_ARGS > /dev/null 2> $out/set$n.err
done
python3 - <<PY
import csv, glob, collections
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$out/set*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        tag = n.split("(")[0].replace("void ", "").replace("(anonymous namespace)::", "")[:40]
        agg[tag][r["Counter_Name"]].append(float(r["Counter_Value"]))
for tag, d in agg.items():
    for k, v in sorted(d.items()):
        print(f"{tag:40s} {k:28s} {sum(v)/len(v):16.1f}")
PY
