#!/bin/bash
# GPU box: HIP API statistics of one filter through the plugin (16 getFrame threads).
# usage: bash tools/prof_plugin_hip.sh <leg filter, e.g. eedi3>
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_plugin_hip_$1
rm -rf $out; mkdir -p $out
timeout 600 rocprofv3 --hip-runtime-trace --stats --output-format csv -d $out/trace -- python3 $R/tools/plugin_bench.py 16 $1 > $out/bench.json 2> $out/trace.err
cat $out/bench.json
f=$(find $out/trace -name '*hip_api_stats.csv' | head -1)
head -14 $f | cut -c1-160
