set -u
mkdir -p gpurun_out
PROF_DIR=prof_fp32_16384 BENCH_ARGS="--fft-len 16384" TRACE_STEPS=100 TRACE_WARMUP=50 bash tools/gpu_profile.sh > gpurun_out/prof_fp32_16384.log 2>&1 || { tail -5 gpurun_out/prof_fp32_16384.log; exit 1; }
tail -12 gpurun_out/prof_fp32_16384.log
timeout -k 10 200 python bench.py --fft-len 16384 2>/dev/null | tee gpurun_out/bench_16384_line.json
