#!/bin/bash
# GPU box: run one command under each prebuilt library variant.  tools/ab_run.sh "<cmd>" <name> ...
cd $GRAFT_REPO_ROOT
cmd=$1; shift
cp vapoursynth-zip_amd/libvszip_hip.so /tmp/ab_base.so
for n in "$@"; do
  if [ $n = base ]; then cp /tmp/ab_base.so vapoursynth-zip_amd/libvszip_hip.so; else cp tools/ab/$n.so vapoursynth-zip_amd/libvszip_hip.so; fi
  echo "== [$n]"; bash -c "$cmd"
done
cp /tmp/ab_base.so vapoursynth-zip_amd/libvszip_hip.so
