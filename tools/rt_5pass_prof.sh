#!/bin/bash
# GPU box: per-kernel times of the small-radius / multi-pass integer RT legs (tools/rt_small_time.py) under rocprofv3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/rt5
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rt5 -- python3 $R/tools/rt_small_time.py > /tmp/rt5.out 2>&1
tail -1 /tmp/rt5.out
python3 - <<'PY'
import csv, glob
f = glob.glob('/tmp/rt5/*/*kernel_stats.csv')[0]
for r in csv.DictReader(open(f)):
    if 'rt_' in r['Name']:
        print('%-90s calls %5s  avg %9.1f us' % (r['Name'].replace('(anonymous namespace)::', '')[:90], r['Calls'], float(r['AverageNs']) / 1e3))
PY
