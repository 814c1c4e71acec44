#!/bin/bash
# 16-bit matrix-core kernels of N < 4096: one-shot vs resident grids of 2..5 workgroups per CU.
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for n in 256 1024 2048; do
  echo "== n=$n mode 6 (one workgroup per tile)"
  timeout -k 10 60 ./bench/tfft_bench --dtype fp16 --n $n --inplace 1 --no-hipfft --mode 6 2>&1 | grep thunderfft || exit 1
  for g in 2 3 4 5; do
    echo "== n=$n resident $g per CU"
    timeout -k 10 60 ./bench/tfft_bench --dtype fp16 --n $n --inplace 1 --no-hipfft --grid $g 2>&1 | grep thunderfft || exit 1
  done
done
