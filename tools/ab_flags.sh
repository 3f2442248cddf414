#!/bin/bash
# GPU box: one bench workload under several dev flag sets. usage: ab_flags.sh <workload> "<flags>" "<flags>" ...
wl=$1; shift
R=$GRAFT_REPO_ROOT
for cfg in "$@"; do
  (cd $R && VSZIP_DEV_R=13 VSZIP_EXTRA_FLAGS="$cfg" python vapoursynth-zip_amd/build.py > /dev/null 2>&1)
  echo "== [$cfg]"
  (cd $R && timeout 200 python3 bench.py --workload $wl --no-cpu --no-others --steps 10 --warmup 2 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.readline()); print('   value', round(d['value']), d['unit'])")
done
