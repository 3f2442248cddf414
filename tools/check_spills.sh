#!/bin/bash
# Compile the BoxBlur CT translation units with -save-temps and list VGPR / spill counts per kernel.
out=${1:-/tmp/vszip_spills}; mkdir -p $out; cd $out
for f in /root/repo/vapoursynth-zip_amd/csrc/boxblur_ct_*.hip; do
  ( /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -ffp-contract=off -fno-fast-math -c $f -o $out/$(basename $f .hip).o -save-temps=obj 2>/dev/null ) &
done; wait
python3 - <<'PY'
import glob, re
for f in sorted(glob.glob("*gfx950.s")):
    txt = open(f).read()
    for m in re.finditer(r"\.name:\s+(\S+)\n(.*?)\.vgpr_count:\s+(\d+)\n\s+\.vgpr_spill_count:\s+(\d+)", txt, re.S):
        name, vg, sp = m.group(1), int(m.group(3)), int(m.group(4))
        if "ring" in name:
            k = re.search(r"ring_kernelI(\w)Li(\d+)ELb(\d)", name)
            print(f"{'u16' if k.group(1)=='t' else 'u8 '} R={int(k.group(2)):2d} general={k.group(3)} vgpr={vg:3d} spill={sp}")
PY
