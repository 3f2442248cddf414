#!/bin/bash
# GPU box: kernel timeline of EEDI3 through the plugin (16 getFrame threads).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_plugin_eedi3
rm -rf $out; mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d $out/trace -- python3 $R/tools/plugin_bench.py 16 eedi3 > $out/bench.json 2> $out/trace.err
cat $out/bench.json
f=$(find $out/trace -name '*kernel_trace.csv' | head -1)
m=$(find $out/trace -name '*memory_copy_trace.csv' | head -1)
python3 - "$f" "$m" <<'P'
import csv,sys,collections
rows=list(csv.DictReader(open(sys.argv[1])))
ev=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r['Kernel_Name']) for r in rows]
ev.sort()
# last 60% of the trace = the timed pull
t_lo=ev[0][0]+(ev[-1][1]-ev[0][0])*0.5
ev=[e for e in ev if e[0]>=t_lo]
span=ev[-1][1]-ev[0][0]
busy=0;cur_s,cur_e=ev[0][0],ev[0][1]
for s,e,_ in ev[1:]:
    if s>cur_e: busy+=cur_e-cur_s;cur_s,cur_e=s,e
    else: cur_e=max(cur_e,e)
busy+=cur_e-cur_s
print("span ms",span/1e6,"kernel-busy frac",busy/span)
d=collections.defaultdict(list)
for s,e,n in ev: d[n[:70]].append(e-s)
for n,v in sorted(d.items(),key=lambda kv:-sum(kv[1])): print(n, len(v), "avg us %.1f"%(sum(v)/len(v)/1e3), "sum ms %.1f"%(sum(v)/1e6))
try:
    mr=list(csv.DictReader(open(sys.argv[2])))
    mv=[(int(r['Start_Timestamp']),int(r['End_Timestamp']),r.get('Direction','')) for r in mr if int(r['Start_Timestamp'])>=t_lo]
    dd=collections.defaultdict(list)
    for s,e,k in mv: dd[k].append(e-s)
    for k,v in dd.items(): print("copy",k,len(v),"avg us %.1f"%(sum(v)/len(v)/1e3),"sum ms %.1f"%(sum(v)/1e6))
except Exception as ex: print("no copies",ex)
P
