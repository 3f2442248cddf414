#!/bin/bash
# GPU box: A/B the BoxBlur dev kernel (r=13 only) under several flag sets, interleaved ROUNDS times
# so that box-to-box and run-to-run noise is visible. usage: ab_ring.sh "<flags A>" "<flags B>" ...
cd $GRAFT_REPO_ROOT
n=0
for cfg in "$@"; do
  n=$((n+1))
  VSZIP_DEV_R=13 VSZIP_EXTRA_FLAGS="$cfg" python vapoursynth-zip_amd/build.py > /dev/null 2>&1
  cp vapoursynth-zip_amd/libvszip_hip.so /tmp/ab_$n.so
done
for round in $(seq 1 ${ROUNDS:-3}); do
  n=0
  for cfg in "$@"; do
    n=$((n+1))
    cp /tmp/ab_$n.so vapoursynth-zip_amd/libvszip_hip.so
    echo -n "[$cfg] "
    if [ $round = 1 ]; then python __graft_entry__.py --smoke 2>&1 | tail -1 | cut -c1-9 | tr '\n' ' '; fi
    python bench.py --steps 30 --warmup 5 --no-cpu --no-others ${BENCH_ARGS} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); l=d['roofline'].get('launch_us') or {}; print(round(d['value']), 'fps', round(d['roofline']['avg_launch_us'],1), 'us/launch', round(d['roofline']['frac'],3), '| >=1s sample: min', round(l.get('min',0),1), 'med', round(l.get('median',0),1), 'max', round(l.get('max',0),1), 'frac_med', round(d['roofline'].get('frac_median',0),3))"
  done
done
