#!/bin/bash
# GPU box: time the EEDI3 kernels under several dev flag sets (ablations) with rocprofv3, and count
# the line kernel's instructions per dispatch. usage: ab_eedi3.sh "<flags>" "<flags>" ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "$@"; do
  (cd $R && VSZIP_DEV_R=13 VSZIP_EXTRA_FLAGS="$cfg" python vapoursynth-zip_amd/build.py > /dev/null 2>&1)
  rm -rf /tmp/e3 /tmp/e3p
  timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/e3 -- python3 $R/tools/prof_legs.py eedi3 > /dev/null 2>&1
  timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --output-format csv -d /tmp/e3p -- python3 $R/tools/prof_legs.py eedi3 > /dev/null 2>&1
  echo "== [$cfg]"
  python3 - <<'PY'
import csv, glob, collections
for f in glob.glob("/tmp/e3/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Name"].replace("void ", "").replace("(anonymous namespace)::", "").split("(")[0]
        if "eedi3" in n: print(f"   {n:24s} calls {r['Calls']:>3s} avg_us {float(r['AverageNs'])/1000:9.1f}")
agg = collections.defaultdict(list)
for f in glob.glob("/tmp/e3p/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "eedi3_line" in r["Kernel_Name"]: agg[r["Counter_Name"]].append(float(r["Counter_Value"]))
print("   line kernel per dispatch:", {k: f"{sum(v)/len(v)/1e9:.3f}G" for k, v in sorted(agg.items())})
PY
done
