#!/bin/bash
# The C++ bench's tables with the current kernels (text output of bench/tfft_bench): fp32 sweep with the hipFFT
# and copy rows (e2e protocol and in-kernel repeats), in place, the 16-bit sweeps, the multi-pass lengths, and the SQ
# counters of the fp16 N = 4096 kernel. --warmup / --runs as the binary defaults (the reference's 1 + 10) unless SETTLED=1.
set -u
O=gpurun_out/r03_tables; rm -rf $O; mkdir -p $O
B=./bench/tfft_bench
timeout -k 10 200 $B --inplace 0 > $O/bench_cpp_sweep_e2e.txt 2>&1 || exit 1
timeout -k 10 200 $B --inplace 0 --inside-repeats 200 > $O/bench_cpp_sweep.txt 2>&1 || exit 1
timeout -k 10 200 $B --inplace 1 --no-hipfft > $O/bench_cpp_sweep_inplace.txt 2>&1 || exit 1
for dt in fp16 bf16; do for ip in 1 0; do
  echo "### $dt inplace=$ip" >> $O/16bit_sweep.txt
  timeout -k 10 200 $B --dtype $dt --inplace $ip --no-hipfft --inside-repeats 50 2>&1 | grep -E "type|thunderfft\(|copy" >> $O/16bit_sweep.txt || exit 1
done; done
for n in 32768 65536 262144 1048576 4194304 16777216 67108864 268435456; do
  timeout -k 10 200 $B --n $n --points $(( n > 67108864 ? n : 67108864 )) --inplace 0 2>&1 | grep -E "thunderfft\(|hipfft" >> $O/multipass_sweep.txt || exit 1
done
BENCH_ARGS="--dtype fp16 --n 4096 --inplace 0" bash tools/gpu_pmc_bench.sh > /dev/null && python3 tools/summarize_pmc.py gpurun_out/pmc_bench fft_mfma4096 > $O/sq_counters_fp16_4096.txt
ls -la $O
