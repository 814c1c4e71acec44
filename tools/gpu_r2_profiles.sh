#!/bin/bash
# The tracked profiles: rocprofv3 kernel stats + FETCH_SIZE / WRITE_SIZE of bench.py per configuration.
set -u
run() { tag=$1; shift; PROF_DIR=prof_$tag BENCH_ARGS="$*" bash tools/gpu_profile.sh > gpurun_out/prof_$tag.log 2>&1 || { tail -5 gpurun_out/prof_$tag.log; exit 1; }; echo "done $tag"; }
mkdir -p gpurun_out
run fp32_4096
run fp32_8192 --n 8192
run fp32_16384 --n 16384
run fp16_4096 --dtype fp16
run bf16_4096 --dtype bf16
run fp32_2p26 --n 67108864 --batch 1
