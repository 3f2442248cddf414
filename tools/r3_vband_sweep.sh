cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for b in 32 48 64 96 128 256; do
  rm -rf /tmp/rtp; VSZIP_RT_VBAND=$b rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/rtp -- python3 $R/tools/prof_legs.py boxblur_other > /dev/null 2>&1
  f=$(ls /tmp/rtp/*/*kernel_stats.csv | head -1)
  echo "band $b: $(grep 'vband_kernel<unsigned short, true' $f | cut -d, -f2-4)"
done
