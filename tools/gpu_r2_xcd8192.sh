#!/bin/bash
# N = 8192 default kernel: run length of consecutive transforms per XCD, out of place and in place.
set -u
B=./bench/tfft_bench
for i in 1 2; do
for x in 2 3 4 5 6; do
for ip in 0 1; do
echo "### xcd=$x inplace=$ip"; timeout -k 10 120 $B --n 8192 --xcd $x --inplace $ip --no-hipfft 2>&1 | grep -E "thunderfft\(|rror" || exit 1
done; done; done
