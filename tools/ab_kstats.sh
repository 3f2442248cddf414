#!/bin/bash
# GPU box: per-kernel times (tools/kstats.sh) of one tools/prof_legs.py leg under each prebuilt library variant (tools/variant.sh).
#   tools/ab_kstats.sh <leg> <kernel name pattern> <name> <name> ...   ("base" = the in-tree build)
cd $GRAFT_REPO_ROOT
leg=$1; pat=$2; shift; shift
cp vapoursynth-zip_amd/libvszip_hip.so /tmp/ab_base.so
for n in "$@"; do
  if [ $n = base ]; then cp /tmp/ab_base.so vapoursynth-zip_amd/libvszip_hip.so; else cp tools/ab/$n.so vapoursynth-zip_amd/libvszip_hip.so; fi
  echo "[$n] $(cut -c1-160 gpurun_out/abk_${n}_out.json 2>/dev/null)"
  bash tools/kstats.sh abk_$n tools/prof_legs.py $leg | grep -E "$pat"
  echo "    $(cut -c1-200 gpurun_out/abk_${n}_out.json)"
done
cp /tmp/ab_base.so vapoursynth-zip_amd/libvszip_hip.so
