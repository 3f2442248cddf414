cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "1920 1080 64" "3840 2160 16" "1280 720 144"; do
  rm -rf /tmp/cp
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/cp -- python3 $R/tools/ctf_small_prof.py $cfg > /dev/null 2> /tmp/cp.err
  f=$(ls -t /tmp/cp/*/*kernel_stats.csv | head -1)
  echo "== $cfg"
  python3 - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if float(r["Percentage"]) > 1:
        print(f'   {r["Name"][:100]:100s} calls {r["Calls"]:>5s} avg {float(r["AverageNs"]) / 1e3:9.1f} us  {r["Percentage"]} %')
PY
done
