# first-allocation launch time of the headline batch against the distance between consecutive planes in the arenas
for sp in 0 16 32 48 64 72 80 96 112 128 160 192 256 320; do VSZIP_BENCH_PLACEMENT_TRIES=1 VSZIP_BENCH_PLANE_SPACING_MIB=$sp python bench.py --no-cpu --no-others --steps 300 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('spacing',$sp,'tries1', round(d['roofline']['avg_launch_us'],1), round(d['roofline']['frac'],4), d['roofline']['launch_us']['min'], d['roofline']['launch_us']['max'])"; done
python tools/which_gpu.py 2>/dev/null | tail -2
