#!/bin/bash
# usage (on the GPU box): bash tools/prof.sh <tag> [bench args...]
# kernel-trace + stats, then three PMC passes (own runs: gpurun refuses pmc + tracing in one).
tag=$1; shift
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
out=$R/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 $R/bench.py --no-cpu "$@" > $out/bench_trace.json 2> $out/trace.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 $R/bench.py --no-cpu "$@" > /dev/null 2> $out/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_write -- python3 $R/bench.py --no-cpu "$@" > /dev/null 2> $out/pmc_write.err
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc_sq -- python3 $R/bench.py --no-cpu "$@" > /dev/null 2> $out/pmc_sq.err
cd $out && find . -name "*.csv" | head -30
