#!/bin/bash
# Container: build a variant of libvszip_hip.so with ONE source recompiled under extra flags.
#   tools/variant.sh <name> <source stem> "<extra flags>"   ->  tools/ab/<name>.so   (travels with gpurun, git-ignored)
# (timing-only macros - VSZIP_*_TIMING_*, VSZIP_*DIAG_* - also need -DVSZIP_DEV_VARIANTS in the flags: common.hpp refuses them otherwise)
set -e
cd "$(dirname "$0")/.."
name=$1; stem=$2; extra=$3
python vapoursynth-zip_amd/build.py > /dev/null
mkdir -p tools/ab /tmp/variant_$name
B=vapoursynth-zip_amd/csrc/_build
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math -Wall -Wno-unused-function \
  $(python -c "import sys; sys.path.insert(0, 'vapoursynth-zip_amd'); import build; print(' '.join(build.FILE_FLAGS.get('$stem', [])))") $extra -c vapoursynth-zip_amd/csrc/$stem.hip -o /tmp/variant_$name/$stem.o
objs=$(ls $B/*.o | grep -v "/$stem.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o tools/ab/$name.so $objs /tmp/variant_$name/$stem.o
echo tools/ab/$name.so
