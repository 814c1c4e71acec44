#!/bin/bash
# Runs on the GPU box: GPU parity tests, then the C++ bench sweep. Stops at the first crash.
set -u
mkdir -p gpurun_out
timeout -k 10 900 python -u -m pytest tests -m gpu -q --timeout 120 --timeout-method thread -x > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -n 25 gpurun_out/pytest_gpu.log
if [ $rc -ne 0 ] && [ $rc -ne 1 ]; then echo "pytest crashed rc=$rc"; exit $rc; fi
timeout -k 10 300 ./bench/tfft_bench "$@" > gpurun_out/bench_cpp.log 2>&1
rc2=$?
cat gpurun_out/bench_cpp.log
exit $(( rc != 0 ? rc : rc2 ))
