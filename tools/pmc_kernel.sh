#!/bin/bash
# GPU box: SQ counters (mean per dispatch, in millions) of kernels whose name contains <pattern>.
# usage: bash tools/pmc_kernel.sh <pattern> <script> <args...>
pat=$1; shift
R=$GRAFT_REPO_ROOT
script=$R/$1; shift
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmck
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/pmck/a -- python3 $script "$@" > /dev/null 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_LDS SQ_INST_CYCLES_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmck/b -- python3 $script "$@" > /dev/null 2>&1
python3 - "$pat" <<'PY'
import csv, glob, collections, sys
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmck/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if sys.argv[1] in n:
            acc[n.split("(")[0].replace("void (anonymous namespace)::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, c in acc.items():
    print(n)
    for k, v in sorted(c.items()):
        print("   %-24s %12.3f M  (n=%d)" % (k, sum(v) / len(v) / 1e6, len(v)))
PY
