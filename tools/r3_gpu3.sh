python -m pytest tests/test_gpu_boxblur.py tests/test_gpu_reference_suite.py tests/test_gpu_random.py tests/test_gpu_depth_parity.py -x -q 2>&1 | tail -4
for f in 0 1; do VSZIP_RT_NO_FUSED=$([ $f = 0 ] && echo 1 || echo "") python - <<PY
import sys, os
if not os.environ.get("VSZIP_RT_NO_FUSED"): os.environ.pop("VSZIP_RT_NO_FUSED", None)
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import bench, vszip_amd, torch
dev=vszip_amd.Device(0)
timed=bench.Timed(dev, lambda: dev.sync(), 0.2)
o=bench.boxblur_other_paths_leg(dev,timed)
r=bench.boxblur_1080p_5pass_leg(dev,timed,True)
print('fused',$f, {k:(round(v['value']),round(v['roofline']['frac'],4)) for k,v in o.items() if 'rt' in k}, '5pass', round(r['value']), round(r['roofline']['frac'],4))
PY
done
bash tools/r3_rtprof.sh
