#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -q --timeout 120 --timeout-method thread -x > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -n 8 gpurun_out/pytest_gpu.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; exit $rc; fi
B=./bench/tfft_bench
run() { echo "### $*"; timeout -k 10 120 $B "$@" --no-hipfft 2>&1 | grep -E "thunderfft|error|Error" || exit 1; }
run --n 256,512,1024,2048,4096,8192,16384
run --n 256,512,1024,2048,4096,8192,16384 --inplace 0
run --n 4096,8192,16384 --mode 3
run --n 4096,8192,16384 --mode 3 --inplace 0
run --n 4096 --mode 3 --grid 2
run --n 4096 --mode 3 --grid 3
run --n 8192,16384 --mode 3 --grid 4
