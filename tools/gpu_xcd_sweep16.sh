#!/bin/bash
# 16-bit kernels: run length of consecutive tiles per XCD (tfft_exec_options.xcd_run_log2), every length, in place and out of place.
set -u
B=./bench/tfft_bench
for rep in 1 2; do for dt in fp16 bf16; do for ip in 1 0; do for x in 2 3 4 5 6; do
echo "### $dt inplace=$ip xcd=$x"
timeout -k 10 120 $B --dtype $dt --inplace $ip --no-hipfft --xcd $x 2>&1 | grep -E "thunderfft\(" || exit 1
done; done; done; done
