#!/bin/bash
# Everything the profiles/r03_* files come from, in one GPU call; every step under its own time limit.
set -u
mkdir -p gpurun_out
timeout -k 10 600 bash tools/gpu_r3_profiles.sh || exit 1
timeout -k 10 500 bash tools/gpu_r3_lines.sh > gpurun_out/r3_bench_py_lines.txt 2>&1 || { tail -5 gpurun_out/r3_bench_py_lines.txt; exit 1; }
grep -v amdgpu.ids gpurun_out/r3_bench_py_lines.txt
timeout -k 10 200 bench/exp16384 20 oneshot:0 res2:1:2:0:5 st1:1:2:1:5 st2:1:2:2:5 st3:1:2:3:5 st4:1:2:4:5 raw2:2:2:0:5 raw2s3:2:2:3:5 res1:1:1:0:5 o4k:0:0:0:5:4 o64k:0:0:0:5:64 o1m:0:0:0:5:1024 o17m:0:0:0:5:17408 > gpurun_out/r3_exp16384.txt 2>&1 || { tail -5 gpurun_out/r3_exp16384.txt; exit 1; }
tail -14 gpurun_out/r3_exp16384.txt
C=./bench/thff_conv1d_fullfft4096
{
for w in "5 10" "50 200"; do set -- $w
  for k in 3 33 63; do for rf in "1 1" "1 0" "0 0"; do
    echo "### K=$k B=64 D=768 real_input/real_filter=$rf warmup=$1 runs=$2"; timeout -k 10 120 $C $k 0 64 768 $1 $2 1 0 1 $rf | head -2 || exit 1
  done; done
  echo "### K=33 B=64 D=768 real/real, filter FFT outside the timed region, warmup=$1 runs=$2"; timeout -k 10 120 $C 33 0 64 768 $1 $2 0 0 1 1 1 | head -2 || exit 1
done
} > gpurun_out/r3_conv_bench.txt 2>&1 || { tail -5 gpurun_out/r3_conv_bench.txt; exit 1; }
grep -c "###" gpurun_out/r3_conv_bench.txt
