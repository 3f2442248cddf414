cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pk
cat > /tmp/rtp.py <<PY
import sys
sys.path.insert(0,'$GRAFT_REPO_ROOT'); sys.path.insert(0,'$GRAFT_REPO_ROOT/tests')
import bench, vszip_amd, torch
dev=vszip_amd.Device(0)
timed=bench.Timed(dev, lambda: dev.sync(), 0.1)
r=bench.boxblur_1080p_5pass_leg(dev,timed,True)
print(r['value'])
PY
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -- python3 /tmp/rtp.py > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
for f in glob.glob("/tmp/pk/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        print(r["Name"][:80], r["Calls"], r["AverageNs"], r["Percentage"])
PY
