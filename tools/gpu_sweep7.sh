#!/bin/bash
# A/B of default, alternative (mode 3) and second alternative (mode 4) plans, in place and out of place.
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -q --timeout 120 --timeout-method thread -x -k "fp32_vs_oracle or alternative" > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -n 5 gpurun_out/pytest_gpu.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; exit $rc; fi
B=./bench/tfft_bench
NS=${NS:-512,1024,2048,4096}
run() { echo "### $*"; timeout -k 10 120 $B "$@" --no-hipfft 2>&1 | grep -E "thunderfft|error|Error" || exit 1; }
for i in 1 2; do
for m in 0 3 4; do
run --n $NS --mode $m
run --n $NS --mode $m --inplace 0
done
done
