#!/bin/bash
# 16-bit storage sweep, in place and out of place, all single-kernel lengths.
set -e
cd "$(dirname "$0")/.."
mkdir -p gpurun_out
for dt in fp16 bf16; do
  for ip in 1 0; do
    echo "== $dt inplace=$ip" >> gpurun_out/sweep16.log
    timeout -k 10 120 ./bench/tfft_bench --dtype $dt --inplace $ip --no-hipfft "$@" >> gpurun_out/sweep16.log 2>&1
  done
done
grep -E "^==|thunderfft" gpurun_out/sweep16.log
