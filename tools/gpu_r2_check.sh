#!/bin/bash
# Runs on the GPU box: the no-allocation entry point's test (by file name while it is not yet collected), the GPU
# parity suite, bench.py and the C++ sweep. Stops at the first failing step.
set -u
mkdir -p gpurun_out
if [ -f tests/wip_no_alloc.py ]; then
  timeout -k 10 300 python -u -m pytest tests/wip_no_alloc.py -m gpu -q --timeout 120 --timeout-method thread -x > gpurun_out/pytest_noalloc.log 2>&1
  rc=$?; tail -n 15 gpurun_out/pytest_noalloc.log
  if [ $rc -ne 0 ]; then echo "no-alloc test rc=$rc"; exit $rc; fi
fi
timeout -k 10 1000 python -u -m pytest tests -m gpu -q --timeout 180 --timeout-method thread -x > gpurun_out/pytest_gpu.log 2>&1
rc=$?
tail -n 25 gpurun_out/pytest_gpu.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; exit $rc; fi
timeout -k 10 300 python bench.py > gpurun_out/bench_py.log 2> gpurun_out/bench_py.err
rc=$?; cat gpurun_out/bench_py.log; tail -n 5 gpurun_out/bench_py.err
if [ $rc -ne 0 ]; then exit $rc; fi
timeout -k 10 300 ./bench/tfft_bench "$@" > gpurun_out/bench_cpp.log 2>&1
rc=$?
cat gpurun_out/bench_cpp.log
exit $rc
