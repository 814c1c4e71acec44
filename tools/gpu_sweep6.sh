#!/bin/bash
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -q --timeout 120 --timeout-method thread -x -k "alternative or resident" > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -n 5 gpurun_out/pytest_gpu.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; exit $rc; fi
B=./bench/tfft_bench
run() { echo "### $*"; timeout -k 10 120 $B "$@" --no-hipfft 2>&1 | grep -E "thunderfft|error|Error" || exit 1; }
for i in 1 2; do
run --n 64,128,256,512,1024,2048,4096,8192,16384
run --n 64,128,256,512,1024,2048,4096,8192,16384 --mode 3
run --n 64,128,256,512,1024,2048,4096,8192,16384 --inplace 0
run --n 64,128,256,512,1024,2048,4096,8192,16384 --mode 3 --inplace 0
done
