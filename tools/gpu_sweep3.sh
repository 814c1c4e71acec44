#!/bin/bash
set -u
B=./bench/tfft_bench
run() { echo "### $*"; timeout -k 10 120 $B "$@" --no-hipfft 2>&1 | grep -E "thunderfft|error|Error" || exit 1; }
for l in 0 18000 45000 120000; do run --n 4096 --extra-lds $l; run --n 4096 --extra-lds $l --inplace 0; done
for l in 0 18000 45000 120000; do run --n 4096 --extra-lds $l --mode 1; done
for l in 0 45000 120000; do run --n 256,1024,2048 --extra-lds $l; done
