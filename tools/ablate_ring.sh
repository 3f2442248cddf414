#!/bin/bash
# GPU box: ablation builds of the BoxBlur ring kernel (r=13), per-kernel time via rocprofv3.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for abl in "" "-DVSZIP_ABL_NOSTORE" "-DVSZIP_ABL_MULHI" "-DVSZIP_DIVF" "-DVSZIP_DIV24" "-DVSZIP_ABL_NOSCAN" "-DVSZIP_ABL_NOLDS" "-DVSZIP_ABL_NOLDS -DVSZIP_ABL_MULHI -DVSZIP_ABL_NOSCAN" "-DVSZIP_ABL_NOLDS -DVSZIP_ABL_MULHI -DVSZIP_ABL_NOSCAN -DVSZIP_ABL_NOSTORE"; do
  (cd $R && VSZIP_DEV_R=13 VSZIP_EXTRA_FLAGS="-DVSZIP_RING_D=3 -DVSZIP_RING_WPE=2 $abl" python vapoursynth-zip_amd/build.py > /dev/null 2>&1)
  rm -rf /tmp/abl_out
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/abl_out -- python3 $R/bench.py --no-cpu --no-others --steps 8 --warmup 2 > /dev/null 2>&1
  echo "== [$abl]"
  python3 - <<'PY'
import csv, glob
for f in glob.glob("/tmp/abl_out/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        tag = "krow" if "krow" in n else ("edge" if ", true>" in n else "inner")
        print(f"   {tag:6s} calls {r['Calls']:>3s} avg_us {float(r['AverageNs'])/1000:8.1f}")
PY
done
