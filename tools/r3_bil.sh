python -m pytest tests/test_gpu_bilateral.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
python bench.py --workload bilateral --no-cpu --steps 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bilateral walk 1080p', d['value'])"
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pk; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-others --workload bilateral --steps 6 --warmup 2 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
for f in glob.glob("/tmp/pk/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "bilateral" in r["Name"]: print(r["Name"][:70], r["Calls"], r["AverageNs"])
PY
