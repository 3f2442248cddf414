#!/bin/bash
# GPU box: what distinguishes the boxes on which the BoxBlur ring kernel takes 565 us from those on which it takes 665 us?
cd $GRAFT_REPO_ROOT
for f in current_memory_partition current_compute_partition mem_info_vram_total mem_info_vram_used mem_info_vis_vram_total pp_dpm_mclk pp_dpm_sclk pp_dpm_fclk vbios_version; do
  for d in /sys/class/drm/card*/device; do [ -r $d/$f ] && echo "$f: $(tr '\n' ' ' < $d/$f)"; done
done 2>/dev/null | sort -u | head -20
rocm-smi --showmemuse --showuse 2>/dev/null | grep -v "^=\|^$" | head -6
cat /sys/module/amdgpu/parameters/vm_fragment_size /sys/module/amdgpu/parameters/vm_block_size /sys/module/amdgpu/parameters/vm_size 2>/dev/null | tr '\n' ' '; echo
uname -r; cat /sys/module/amdgpu/version 2>/dev/null
python bench.py --no-cpu --no-others --steps 200 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('ring kernel', round(r['avg_launch_us'],1), 'us frac', round(r['frac'],3))"
