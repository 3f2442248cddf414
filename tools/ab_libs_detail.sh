#!/bin/bash
# like ab_libs.sh, also printing the dominant kernel's average launch time (roofline.avg_launch_us)
cd $GRAFT_REPO_ROOT
args=$1; shift
cp vapoursynth-zip_amd/libvszip_hip.so /tmp/ab_base.so
for round in $(seq 1 ${ROUNDS:-2}); do
  for n in "$@"; do
    if [ $n = base ]; then cp /tmp/ab_base.so vapoursynth-zip_amd/libvszip_hip.so; else cp tools/ab/$n.so vapoursynth-zip_amd/libvszip_hip.so; fi
    echo -n "[$n] "
    python bench.py --no-cpu --no-others $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print(round(d['value'],1), d['unit'], round(d['ms_per_step']*1e3,1), 'us/step; kernel', round(r.get('avg_launch_us') or 0,1), 'us frac', round(r['frac'],4))"
  done
done
cp /tmp/ab_base.so vapoursynth-zip_amd/libvszip_hip.so
