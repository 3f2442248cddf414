cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pp
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -- python3 $R/bench.py --workload pipeline --steps 5 --warmup 2 > /tmp/pp.json 2> /tmp/pp.err
f=$(ls -t /tmp/pp/*/*kernel_stats.csv | head -1)
cat /tmp/pp.json | head -c 1500; echo
python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 0.3:
        print(f'   {r["Name"][:100]:100s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"]) / 1e3:9.1f} us  {r["Percentage"]} %')
PY
cp $f $R/gpurun_out/r04_pipeline_kernel_stats.csv
