#!/bin/bash
# GPU box: per-kernel times of the BoxBlur "other paths" leg (RT r=30, RT 3+3 passes r=5, ...) under rocprofv3, with and without the vertical pass's LDS ring
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for mode in ring noring; do
  rm -rf /tmp/rtprof_$mode
  if [ $mode = noring ]; then export VSZIP_RT_NO_VRING=1; else unset VSZIP_RT_NO_VRING; fi
  rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rtprof_$mode -- python3 $R/tools/prof_legs.py boxblur_other > /dev/null 2>&1
  f=$(ls /tmp/rtprof_$mode/*/*kernel_stats.csv | head -1)
  echo "== $mode"; grep -i "rt_" $f | cut -d, -f1-4 | sed 's/(anonymous namespace):://g' | cut -c1-120
done
