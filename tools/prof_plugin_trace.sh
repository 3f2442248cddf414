#!/bin/bash
# GPU box: kernel + memory-copy timeline statistics of one plugin leg (tools/plugin_bench.py). usage: prof_plugin_trace.sh <threads> <leg filter>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=/tmp/ppt; rm -rf $out; mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $out/trace -- python3 $R/tools/plugin_bench.py $1 "$2" > $out/bench.json 2> $out/trace.err
grep filter $out/bench.json
for f in $(find $out/trace -name '*kernel_stats.csv' -o -name '*memory_copy_stats.csv' -o -name '*domain_stats.csv'); do echo "-- $(basename $f)"; head -12 $f | cut -c1-200; done
python3 - <<'PY'
import csv, glob
for f in glob.glob('/tmp/ppt/trace/*/*memory_copy_trace.csv'):
    rows = list(csv.DictReader(open(f)))
    if not rows: continue
    print("copies:", len(rows), "columns:", list(rows[0].keys()))
    t0 = min(int(r["Start_Timestamp"]) for r in rows); t1 = max(int(r["End_Timestamp"]) for r in rows)
    busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows)
    print(f"span {(t1 - t0) / 1e6:.1f} ms, summed copy time {busy / 1e6:.1f} ms")
    for r in rows[-6:]:
        print({k: r[k] for k in r if k in ("Direction", "Start_Timestamp", "End_Timestamp", "Bytes", "Size")}, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, "us")
PY
