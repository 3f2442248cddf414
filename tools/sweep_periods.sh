#!/bin/bash
# GPU box: sweep the ring kernel's band length (periods of NR rows) at several batch sizes.
cd $GRAFT_REPO_ROOT
for F in ${FRAMES:-16}; do
for P in ${PERIODS:-1 2 3 4 6 8}; do
  echo -n "F=$F periods=$P: "
  VSZIP_RING_PERIODS=$P timeout 300 python bench.py --no-cpu --no-others --frames $F --steps 30 --warmup 5 2>/dev/null | python3 -c "import json,sys; j=json.loads(sys.stdin.read()); r=j['roofline']; print(round(j['value']), 'fps  ring', round(r['avg_launch_us'],1), 'us frac', round(r['frac'],3), ' whole', round(r['whole_filter']['frac'],3))"
done; done
