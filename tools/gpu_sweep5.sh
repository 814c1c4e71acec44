#!/bin/bash
set -u
B=./bench/tfft_bench
run() { echo "### $*"; timeout -k 10 120 $B "$@" --no-hipfft 2>&1 | grep -E "thunderfft|error|Error" || exit 1; }
run --n 256,512,1024,2048,4096,8192,16384
run --n 256,512,1024,2048,4096,8192,16384 --inplace 0
run --n 4096,8192,16384 --mode 3
run --n 4096,8192,16384 --mode 3 --inplace 0
run --n 4096 --mode 1
run --n 4096 --mode 1 --inplace 0
run --n 4096
run --n 4096 --inplace 0
