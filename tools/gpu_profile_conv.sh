#!/bin/bash
# rocprofv3 passes over the fused-convolution benchmark (Figure-5 shape): kernel trace + stats,
# then PMC passes (at most 8 SQ counters per pass).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/prof_conv
mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="${CONV_ARGS:-33 0 64 768 2 5 0 0 1 1 1}"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -o conv -- $R/bench/thff_conv1d_fullfft4096 $ARGS > $OUT/trace_stdout.log 2>&1 || { tail -5 $OUT/trace_stdout.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/pmc_a -o conv -- $R/bench/thff_conv1d_fullfft4096 $ARGS > $OUT/pmc_a_stdout.log 2>&1 || { tail -5 $OUT/pmc_a_stdout.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/pmc_b -o conv -- $R/bench/thff_conv1d_fullfft4096 $ARGS > $OUT/pmc_b_stdout.log 2>&1 || { tail -5 $OUT/pmc_b_stdout.log; exit 1; }
for f in $(find $OUT/trace -name "*kernel_stats.csv"); do head -8 $f; done
