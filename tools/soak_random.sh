#!/bin/bash
# GPU box: the seeded random-geometry parity tests over fresh seeds.  usage: tools/soak_random.sh <first base> <last base>
cd $GRAFT_REPO_ROOT
fail=0
for b in $(seq ${1:-1} ${2:-20}); do
  out=$(VSZIP_TEST_SEED_BASE=$b timeout 600 python -m pytest tests/test_gpu_random.py -x -q 2>&1 | tail -15)
  last=$(echo "$out" | tail -1)
  echo "base $b: $last"
  case "$last" in *failed*|*error*) echo "$out"; fail=1;; esac
done
exit $fail
