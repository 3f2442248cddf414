#!/bin/bash
# GPU box: tools/stream_timing.py under several flag sets (full builds: the streaming kernels are small).
cd $GRAFT_REPO_ROOT
for cfg in "$@"; do
  VSZIP_DEV_R=13 VSZIP_EXTRA_FLAGS="$cfg" python vapoursynth-zip_amd/build.py > /dev/null 2>&1
  echo "== [$cfg]"
  python tools/stream_timing.py 2>&1 | grep -v "amdgpu.ids"
done
