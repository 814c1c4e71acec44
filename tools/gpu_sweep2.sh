#!/bin/bash
set -u
B=./bench/tfft_bench
run() { echo "### $*"; timeout -k 10 120 $B "$@" --no-hipfft 2>&1 | grep -E "thunderfft|error|Error" || exit 1; }
for g in 0 1 2 4 8 16 64 1024; do run --n 4096 --grid $g; done
for g in 0 4 1024; do run --n 4096 --grid $g --inplace 0; done
for g in 0 4 1024; do run --n 4096 --grid $g --mode 1; done
run --n 256,512,1024,2048,8192,16384 --grid 1024
run --n 256,512,1024,2048,8192,16384 --grid 0
