#!/bin/bash
set -u
B=./bench/tfft_bench
run() { echo "### $*"; timeout -k 10 120 $B "$@" --no-hipfft 2>&1 | grep -E "thunderfft|error|Error" || exit 1; }
run --n 4096 --mode 1
run --n 4096 --mode 1 --inplace 0
run --n 4096 --mode 2
run --n 4096 --mode 0 --inplace 0
for g in 1 2 3 4 6 8 16 32 256; do run --n 4096 --grid $g; done
for g in 2 4 8 16 64; do run --n 4096 --mode 1 --grid $g; done
run --n 256,1024,16384 --mode 1
run --n 256,1024,16384 --mode 2
