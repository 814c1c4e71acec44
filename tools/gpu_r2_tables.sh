#!/bin/bash
# The benchmark tables tracked under profiles/ (text output of bench/tfft_bench, SQ counter summaries).
set -u
O=gpurun_out/r02_tables; rm -rf $O; mkdir -p $O
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
BENCH_ARGS="--n 8192 --inplace 0" bash tools/gpu_pmc_bench.sh > /dev/null && python tools/summarize_pmc.py gpurun_out/pmc_bench fft_dit > $O/sq_counters_8192.txt
BENCH_ARGS="--n 16384 --inplace 0" bash tools/gpu_pmc_bench.sh > /dev/null && python tools/summarize_pmc.py gpurun_out/pmc_bench fft_lds > $O/sq_counters_16384.txt
BENCH_ARGS="--n 4096 --inplace 0" bash tools/gpu_pmc_bench.sh > /dev/null && python tools/summarize_pmc.py gpurun_out/pmc_bench fft_lds > $O/sq_counters_4096.txt
BENCH_ARGS="--n 67108864 --points 67108864 --inplace 0" bash tools/gpu_pmc_bench.sh > /dev/null && python tools/summarize_pmc.py gpurun_out/pmc_bench fft_tile > $O/sq_counters_2p26_tiles.txt
ls -la $O
