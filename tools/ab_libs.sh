#!/bin/bash
# GPU box: interleaved A/B of prebuilt library variants (tools/variant.sh) on one bench workload.
#   ROUNDS=3 tools/ab_libs.sh "<bench args>" <name> <name> ...   ("base" = the in-tree build)
cd $GRAFT_REPO_ROOT
args=$1; shift
cp vapoursynth-zip_amd/libvszip_hip.so /tmp/ab_base.so
for round in $(seq 1 ${ROUNDS:-3}); do
  for n in "$@"; do
    if [ $n = base ]; then cp /tmp/ab_base.so vapoursynth-zip_amd/libvszip_hip.so; else cp tools/ab/$n.so vapoursynth-zip_amd/libvszip_hip.so; fi
    echo -n "[$n] "
    python bench.py --no-cpu --no-others $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), d['unit'], round(d['ms_per_step'],3), 'ms/step')"
  done
done
cp /tmp/ab_base.so vapoursynth-zip_amd/libvszip_hip.so
