#!/bin/bash
# GPU box: kernel trace of the config-5 chain (bench.py --workload pipeline).
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_pipeline
rm -rf $out; mkdir -p $out
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/bench.py --workload pipeline --steps 3 --warmup 1 > $out/bench_trace.json 2> $out/trace.err
f=$(find $out/trace -name '*kernel_stats.csv' | head -1)
cp $f $out/kernel_stats.csv
head -20 $f | cut -c1-200
