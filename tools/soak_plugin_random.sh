#!/bin/bash
# GPU box: the seeded random filter-graph test (fused = unfused through libvszip.so) over fresh seed bases.  usage: tools/soak_plugin_random.sh <first> <last>
cd $GRAFT_REPO_ROOT
fail=0
for b in $(seq ${1:-1} ${2:-50}); do
  out=$(VSZIP_TEST_SEED_BASE=$b timeout 600 python -m pytest tests/test_gpu_plugin_random.py -x -q 2>&1 | tail -25)
  last=$(echo "$out" | tail -1)
  case "$last" in *failed*|*error*) echo "base $b: $last"; echo "$out"; fail=1;; esac
done
echo "bases ${1:-1}..${2:-50} done, fail=$fail"
exit $fail
