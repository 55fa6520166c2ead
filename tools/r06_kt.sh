#!/bin/bash
# Round 6: rocprofv3 kernel trace of the term loop for the shipping library and every build/libpovar_hip_exp_*.so -- e0_ck and the
# per-camera kernel behind it, each by itself.  usage: tools/r06_kt.sh <tag>
tag=$1; out=gpurun_out/$tag; mkdir -p $out
export TMPDIR=/tmp
for so in povar_amd/libpovar_hip.so build/libpovar_hip_exp_*.so; do
  [ -f $so ] || continue
  name=$(basename $so .so); name=${name#libpovar_hip_exp_}; [ $name = libpovar_hip ] && name=shipping
  POVAR_LIB=$PWD/$so rocprofv3 --kernel-trace --stats --output-format csv -d $out/kt_$name -- python3 tools/ck_trace.py venice-1778 --variants 1 --solves 6 > /dev/null 2> $out/kt_$name.err
  f=$(ls $out/kt_$name/*/*kernel_stats.csv 2>/dev/null | head -1)
  echo "== $name" | tee -a $out/kt_summary.txt
  [ -n "$f" ] && python3 tools/kernel_stats_table.py $f 40 | grep -E "e0_ck|cam_cold_sum_binv|e0_lpl" | tee -a $out/kt_summary.txt
done
