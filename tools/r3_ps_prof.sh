#!/bin/bash
# GPU box: per-kernel times of the thresholded PlaneMinMax paths (tools/planestats_ab.py) under rocprofv3
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf /tmp/psprof
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/psprof -- python3 $R/tools/planestats_ab.py > $R/gpurun_out/ps_ab.txt 2>&1
f=$(ls /tmp/psprof/*/*kernel_stats.csv | head -1)
cp $f $R/gpurun_out/ps_ab_kernel_stats.csv
cut -d, -f1-5 $f | head -20
tail -7 $R/gpurun_out/ps_ab.txt
