#!/bin/bash
# rocprofv3 passes over one tfft_bench configuration (BENCH_ARGS, e.g. "--dtype fp16 --n 4096"):
# kernel trace + stats, FETCH_SIZE, WRITE_SIZE, SQ counters. Output: gpurun_out/prof_cpp/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_cpp
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="${BENCH_ARGS:---n 4096} --no-hipfft --runs 5"
timeout -k 10 200 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o b -- $R/bench/tfft_bench $ARGS > $OUT/trace.log 2>&1 || { tail -5 $OUT/trace.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch -o b -- $R/bench/tfft_bench $ARGS > $OUT/fetch.log 2>&1 || { tail -5 $OUT/fetch.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write -o b -- $R/bench/tfft_bench $ARGS > $OUT/write.log 2>&1 || { tail -5 $OUT/write.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_LDS_BANK_CONFLICT --output-format csv -d $OUT/sq -o b -- $R/bench/tfft_bench $ARGS > $OUT/sq.log 2>&1 || { tail -5 $OUT/sq.log; exit 1; }
head -5 $OUT/trace/b_kernel_stats.csv
