#!/bin/bash
# GPU box: interleaved A/B of prebuilt library variants (tools/variant.sh) on any python script.
#   ROUNDS=2 tools/ab_script.sh tools/bil_sizes.py <name> <name> ...   ("base" = the in-tree build)
cd $GRAFT_REPO_ROOT
script=$1; shift
cp vapoursynth-zip_amd/libvszip_hip.so /tmp/ab_base.so
for round in $(seq 1 ${ROUNDS:-2}); do
  for n in "$@"; do
    if [ $n = base ]; then cp /tmp/ab_base.so vapoursynth-zip_amd/libvszip_hip.so; else cp tools/ab/$n.so vapoursynth-zip_amd/libvszip_hip.so; fi
    echo -n "[$n] "; python $script 2>/dev/null | tail -1
  done
done
cp /tmp/ab_base.so vapoursynth-zip_amd/libvszip_hip.so
