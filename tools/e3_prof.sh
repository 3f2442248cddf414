#!/bin/bash
# GPU box: per-kernel times of the EEDI3 leg for library variants (tools/variant.sh), rocprofv3 --kernel-trace --stats
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cp $R/vapoursynth-zip_amd/libvszip_hip.so /tmp/e3_base.so
for n in "$@"; do
  if [ $n = base ]; then cp /tmp/e3_base.so $R/vapoursynth-zip_amd/libvszip_hip.so; else cp $R/tools/ab/$n.so $R/vapoursynth-zip_amd/libvszip_hip.so; fi
  rm -rf /tmp/e3p_$n
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/e3p_$n -- python3 $R/tools/prof_legs.py eedi3 > /tmp/e3p_$n.json 2> /tmp/e3p_$n.err
  f=$(ls -t /tmp/e3p_$n/*/*kernel_stats.csv | head -1)
  echo "== $n: $(grep -o '"value": [0-9.]*' /tmp/e3p_$n.json | head -1)"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.5:
        print(f'   {r["Name"][:90]:90s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"]) / 1e3:9.1f} us  {r["Percentage"]} %')
PY
done
cp /tmp/e3_base.so $R/vapoursynth-zip_amd/libvszip_hip.so
