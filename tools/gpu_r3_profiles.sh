#!/bin/bash
# The tracked profiles (profiles/r03_*): rocprofv3 kernel trace + stats and FETCH_SIZE / WRITE_SIZE of bench.py, one
# configuration per run. 16-bit configurations get 50 warm-up and 100 timed launches (the matrix-core kernels
# need tens of milliseconds of sustained work to reach their rate).
set -u
run() { tag=$1; shift; PROF_DIR=prof_$tag BENCH_ARGS="$*" bash tools/gpu_profile.sh > gpurun_out/prof_$tag.log 2>&1 || { tail -5 gpurun_out/prof_$tag.log; exit 1; }; echo "done $tag"; }
mkdir -p gpurun_out
run fp32_4096
run fp32_8192 --fft-len 8192
TRACE_STEPS=100 TRACE_WARMUP=50 run fp32_16384 --fft-len 16384
TRACE_STEPS=100 TRACE_WARMUP=50 run fp16_4096 --dtype fp16
TRACE_STEPS=100 TRACE_WARMUP=50 run bf16_4096 --dtype bf16
run fp32_2p26 --fft-len 67108864 --batch 1
