#!/bin/bash
# GPU box: kernel trace of single-pair SSIMULACRA2 calls (the plugin's getFrame shape).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_ssim_single
rm -rf $out; mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --output-format csv -d $out/trace -- python3 $R/bench.py --workload ssimulacra2 --frames 4 --no-cpu --steps 3 --warmup 1 > $out/bench_trace.json 2> $out/trace.err
f=$(find $out/trace -name '*kernel_trace.csv' | head -1)
python3 - "$f" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows=[r for r in rows if 'ssim' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
last=rows[-11:]
t0=int(last[0]['Start_Timestamp'])
for r in last:
    print(r['Kernel_Name'][23:45], int(r['Start_Timestamp'])-t0, int(r['End_Timestamp'])-int(r['Start_Timestamp']), r.get('Grid_Size_X'), r.get('Grid_Size_Y'), r.get('Grid_Size_Z'))
P
