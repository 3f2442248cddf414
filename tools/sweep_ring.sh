#!/bin/bash
# GPU box: rebuild the BoxBlur dev kernel (r=13 only) with several ring configs and bench each.
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  D=${cfg%%:*}; W=${cfg##*:}
  VSZIP_DEV_R=13 VSZIP_EXTRA_FLAGS="-DVSZIP_RING_D=$D -DVSZIP_RING_WPE=$W" python vapoursynth-zip_amd/build.py > /dev/null 2>&1
  echo -n "D=$D WPE=$W : "
  python __graft_entry__.py --smoke 2>&1 | tail -1 | cut -c1-60 | tr '\n' ' '
  python -m pytest tests/test_gpu_boxblur.py -q -x -k "natural or batch or stride" 2>&1 | tail -1 | tr '\n' ' '
  python bench.py --steps 30 --warmup 5 --no-cpu --no-others 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']), 'fps', round(d['roofline']['avg_launch_us'],1), 'us/launch', round(d['roofline']['frac'],3))"
done
