#!/bin/bash
# GPU box: A/B prebuilt library variants (tools/variant.sh) on one tools/prof_legs.py leg, printing one key.
#   tools/ab_legs.sh <leg> <result key> <name> <name> ...   ("base" = the in-tree build)
cd $GRAFT_REPO_ROOT
leg=$1; key=$2; shift; shift
cp vapoursynth-zip_amd/libvszip_hip.so /tmp/ab_base.so
for round in $(seq 1 ${ROUNDS:-2}); do
  for n in "$@"; do
    if [ $n = base ]; then cp /tmp/ab_base.so vapoursynth-zip_amd/libvszip_hip.so; else cp tools/ab/$n.so vapoursynth-zip_amd/libvszip_hip.so; fi
    echo -n "[$n] "
    python tools/prof_legs.py $leg 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1])['$key']; print(round(d['value'],1), d['unit'], 'frac', round(d['roofline']['frac'],3))"
  done
done
cp /tmp/ab_base.so vapoursynth-zip_amd/libvszip_hip.so
