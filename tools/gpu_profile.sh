#!/bin/bash
# rocprofv3 passes over bench.py on the GPU box: kernel trace + stats, then one PMC pass per
# counter group (FETCH_SIZE and WRITE_SIZE do not fit one pass). Outputs land in gpurun_out/.
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/${PROF_DIR:-prof}
rm -rf $OUT
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
EXTRA="${BENCH_ARGS:-}"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o bench -- python3 $R/bench.py --steps ${TRACE_STEPS:-20} --warmup ${TRACE_WARMUP:-3} --no-cpu-baseline $EXTRA > $OUT/trace_stdout.log 2>&1 || { tail -5 $OUT/trace_stdout.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -o bench -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $EXTRA > $OUT/pmc_fetch_stdout.log 2>&1 || { tail -5 $OUT/pmc_fetch_stdout.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -o bench -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline $EXTRA > $OUT/pmc_write_stdout.log 2>&1 || { tail -5 $OUT/pmc_write_stdout.log; exit 1; }
find $OUT -name "*.csv" | head -20
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do head -8 $f; done
