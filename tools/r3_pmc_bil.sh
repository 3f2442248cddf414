# SQ counters + LDS-array counters of the bilateral kernels (walk kernel = default)
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/pmck
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --output-format csv -d /tmp/pmck/a -- python3 $R/bench.py --no-cpu --no-others --workload bilateral --steps 6 --warmup 2 > /dev/null 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmck/b -- python3 $R/bench.py --no-cpu --no-others --workload bilateral --steps 6 --warmup 2 > /dev/null 2>&1
timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pmck/t -- python3 $R/bench.py --no-cpu --no-others --workload bilateral --steps 6 --warmup 2 > /dev/null 2>&1
python3 - <<'PY'
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("/tmp/pmck/[ab]/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "bilateral" in n:
            acc[n.split("(")[0].replace("void (anonymous namespace)::", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
for n, c in acc.items():
    print(n)
    for k, v in sorted(c.items()):
        print("   %-24s %12.3f M  (n=%d)" % (k, sum(v) / len(v) / 1e6, len(v)))
for f in glob.glob("/tmp/pmck/t/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        if "bilateral" in r["Name"]: print(r["Name"][:70], r["Calls"], r["AverageNs"])
PY
