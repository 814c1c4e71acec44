#!/bin/bash
# A/B of the N = 8192 / 16384 fp32 plans: default, alternative (mode 3), resident prefetch (mode 4); out of place and in place.
set -u
mkdir -p gpurun_out
B=./bench/tfft_bench
run() { echo "### $*"; timeout -k 10 120 $B "$@" --no-hipfft 2>&1 | grep -E "thunderfft|copy|error|Error" || exit 1; }
for i in 1 2; do
for m in 0 3 4; do
run --n 8192,16384 --mode $m --inplace 0
run --n 8192,16384 --mode $m --inplace 1
done
done
run --n 8192,16384 --mode 1 --inplace 0
