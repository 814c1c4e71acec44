#!/bin/bash
# GPU box: fused-convolution parity tests, then the Figure-5 style benchmark binary.
set -u
mkdir -p gpurun_out
timeout -k 10 600 python -u -m pytest tests/test_conv_gpu.py -m gpu -q --timeout 120 --timeout-method thread -x > gpurun_out/pytest_conv.log 2>&1
rc=$?
tail -n 15 gpurun_out/pytest_conv.log
if [ $rc -ne 0 ]; then echo "pytest rc=$rc"; exit $rc; fi
C=./bench/thff_conv1d_fullfft4096
{
for args in "9 0 4 8 2 5 1 1 1 0 0" "9 0 4 8 2 5 1 1 0 0 0" "33 0 2 8 2 5 0 1 1 1 0" "63 0 2 8 2 5 1 1 1 1 1"; do
  echo "### $args"; timeout -k 10 120 $C $args || exit 1
done
for k in 3 33 63; do
  for rf in "1 1" "1 0" "0 0"; do
    echo "### K=$k B=64 D=768 real_input/real_filter=$rf"; timeout -k 10 120 $C $k 0 64 768 5 10 1 0 1 $rf || exit 1
  done
done
echo "### K=33 B=64 D=768 filter FFT outside the timed region"; timeout -k 10 120 $C 33 0 64 768 5 10 0 0 1 1 1 || exit 1
} > gpurun_out/conv_bench.log 2>&1
rc=$?
cat gpurun_out/conv_bench.log
exit $rc
