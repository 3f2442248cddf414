#!/bin/bash
# GPU box: every `others` leg with a roofline ALONE under rocprofv3 --kernel-trace --stats, so that
# algorithmic bytes / (the profile's kernel time per call) / 8e12 can be set beside the leg's own `frac`.
# usage: bash tools/roofline_check.sh <round tag>  ->  gpurun_out/rc_<tag>/<leg>.{json,csv}, gpurun_out/rc_<tag>/roofline_check.md
rt=${1:-r04}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/rc_$rt
rm -rf $out; mkdir -p $out
legs=${LEGS:-xpsnr_batch plane_average_4k plane_minmax_4k plane_minmax_thr_4k limiter limit_filter boxblur_1080p boxblur_1080p_5pass boxblur_1080p_r1x2_yuv420p8 boxblur_rt_r30_4k boxblur_rt_r5x3_4k boxblur_ct_float_r13_4k boxblur_rt_float_r5x3_4k boxblur_ct_u8_r13_4k bilateral_1080p bilateral_4k ssimulacra2_4k eedi3}
for leg in $legs; do
  d=/tmp/rc_$leg; rm -rf $d
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $d -- python3 $R/tools/prof_legs.py $leg > $out/$leg.json 2> $out/$leg.err
  f=$(ls -t $d/*/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && cp $f $out/$leg.csv
  rm -rf $d
done
python3 $R/tools/roofline_check.py $out > $out/roofline_check.md
cat $out/roofline_check.md
