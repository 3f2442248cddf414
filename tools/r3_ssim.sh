python -m pytest tests/test_gpu_ssimulacra2.py tests/test_gpu_ssim_prestage.py tests/test_gpu_ssim_yuv.py tests/test_gpu_fullsize.py -x -q 2>&1 | tail -3
python bench.py --workload ssimulacra2 --no-cpu --steps 10 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ssimulacra2 4k pairs/s', d['value'])"
cd /tmp && export TMPDIR=/tmp; rm -rf /tmp/pk; timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pk -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu --no-others --workload ssimulacra2 --steps 4 --warmup 1 > /dev/null 2>&1
python3 - <<'PY'
import csv,glob
for f in glob.glob("/tmp/pk/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "ssim" in r["Name"]: print(r["Name"][:70], r["Calls"], r["AverageNs"], r["Percentage"])
PY
