#!/bin/bash
# The fused convolution's tracked measurements: the binary's lines at the Figure-5 shape (5 + 10 runs as the reference's
# driver makes them, and 50 + 200), the Figure-5 sweep both ways, and the settled plan A/B of tools/gpu_conv_ab.py.
set -u
mkdir -p gpurun_out
C=./bench/thff_conv1d_fullfft4096
{
for w in "5 10" "50 200"; do set -- $w
  for k in 3 33 63; do for rf in "1 1" "1 0" "0 0"; do
    echo "### K=$k B=64 D=768 real_input/real_filter=$rf warmup=$1 runs=$2"; timeout -k 10 120 $C $k 0 64 768 $1 $2 1 0 1 $rf | head -2 || exit 1
  done; done
  echo "### K=33 B=64 D=768 real/real, filter FFT outside the timed region, warmup=$1 runs=$2"; timeout -k 10 120 $C 33 0 64 768 $1 $2 0 0 1 1 1 | head -2 || exit 1
done
} > gpurun_out/r3_conv_bench.txt 2>&1 || { tail -5 gpurun_out/r3_conv_bench.txt; exit 1; }
timeout -k 10 300 python3 tools/figure5_sweep.py --csv gpurun_out/figure5_w5_r10.csv > gpurun_out/figure5_w5_r10.log 2>&1 || { tail -5 gpurun_out/figure5_w5_r10.log; exit 1; }
timeout -k 10 300 python3 tools/figure5_sweep.py --warmup 50 --runs 200 --csv gpurun_out/figure5_w50_r200.csv > gpurun_out/figure5_w50_r200.log 2>&1 || { tail -5 gpurun_out/figure5_w50_r200.log; exit 1; }
for k in 2 0 1; do timeout -k 10 150 python3 tools/gpu_conv_ab.py --kind $k --plans 0,4,5,7,6 --rounds 8 --burst 50 || exit 1; done > gpurun_out/r3_conv_plans.txt 2>&1
grep " ms" gpurun_out/r3_conv_plans.txt
