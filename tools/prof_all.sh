#!/bin/bash
# GPU box: regenerate every rocprofv3 summary of the round in one call.
# usage: bash tools/prof_all.sh <round tag, e.g. r01>     (ONLY="legs" / "boxblur bilateral": just those workloads)
# Each tag gets a --kernel-trace --stats run and three PMC passes (own runs: gpurun refuses
# pmc + tracing in one). Summaries: python3 tools/summarize_prof.py gpurun_out/prof_<tag> profiles/<tag>
rt=${1:-r01}
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
prof() {
  tag=$1; shift
  out=$R/gpurun_out/prof_$tag
  rm -rf $out; mkdir -p $out
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -- python3 "$@" > $out/bench_trace.json 2> $out/trace.err
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -- python3 "$@" > /dev/null 2> $out/pmc_fetch.err
  timeout 600 rocprofv3 --pmc WRITE_SIZE TCC_HIT_sum TCC_MISS_sum --output-format csv -d $out/pmc_write -- python3 "$@" > /dev/null 2> $out/pmc_write.err
  timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $out/pmc_sq -- python3 "$@" > /dev/null 2> $out/pmc_sq.err
  # round 3: what the non-HBM legs are bound by — LDS-array cycles (with the conflict share) and VALU issue cycles against the clock
  timeout 600 rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_INSTS_VMEM_RD SQ_WAIT_INST_LDS SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $out/pmc_lds -- python3 "$@" > /dev/null 2> $out/pmc_lds.err
}
only=${ONLY:-boxblur bilateral ssimulacra2 legs}
want() { case " $only " in *" $1 "*) return 0;; esac; return 1; }
want boxblur && prof ${rt}_boxblur $R/bench.py --no-cpu --no-others --steps 50 --warmup 3 --min-seconds 0.1
want bilateral && prof ${rt}_bilateral $R/bench.py --no-cpu --no-others --workload bilateral --steps 10 --warmup 2
want ssimulacra2 && prof ${rt}_ssimulacra2 $R/bench.py --no-cpu --no-others --workload ssimulacra2 --steps 5 --warmup 1
want legs && prof ${rt}_legs $R/tools/prof_legs.py eedi3 xpsnr boxblur_other planestats ssim_yuv
# the raw outputs exceed what gpurun copies back (64 MiB): summarise here, keep the summaries and the stderr of each pass
mkdir -p $R/gpurun_out/sum_${rt}
for n in $only; do
  o=$n; [ $n = legs ] && o=other_filters
  python3 $R/tools/summarize_prof.py $R/gpurun_out/prof_${rt}_$n $R/gpurun_out/sum_${rt}/${rt}_$o
  mkdir -p $R/gpurun_out/sum_${rt}/err_$n
  cp $R/gpurun_out/prof_${rt}_$n/*.err $R/gpurun_out/prof_${rt}_$n/bench_trace.json $R/gpurun_out/sum_${rt}/err_$n/ 2>/dev/null
  rm -rf $R/gpurun_out/prof_${rt}_$n
done
ls -la $R/gpurun_out/sum_${rt}
