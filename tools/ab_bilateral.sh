#!/bin/bash
# GPU box: Bilateral 1080p bench line under several dev flag sets (ablations). usage: ab_bilateral.sh "<flags>" ...
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for cfg in "$@"; do
  (cd $R && VSZIP_DEV_R=13 VSZIP_EXTRA_FLAGS="$cfg" python vapoursynth-zip_amd/build.py > /dev/null 2>&1)
  echo "== [$cfg]"
  (cd $R && timeout 200 python3 bench.py --workload bilateral --no-cpu --no-others --steps 10 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('   fps', round(d['value']), 'kernel frac', round(d['roofline']['frac'],4))")
done
