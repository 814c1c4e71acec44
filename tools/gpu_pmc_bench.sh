#!/bin/bash
# SQ counter passes over one tfft_bench configuration: BENCH_ARGS="--dtype fp16 --n 4096 --mode 5"
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/pmc_bench
rm -rf $OUT; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
ARGS="${BENCH_ARGS:---n 4096} --no-hipfft --runs 3"
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR --output-format csv -d $OUT/a -o b -- $R/bench/tfft_bench $ARGS > $OUT/a.log 2>&1 || { tail -5 $OUT/a.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --output-format csv -d $OUT/b -o b -- $R/bench/tfft_bench $ARGS > $OUT/b.log 2>&1 || { tail -5 $OUT/b.log; exit 1; }
timeout -s KILL 120 rocprofv3 --pmc SQ_INSTS_VALU_MFMA_MOPS_F16 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_MFMA SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_IFETCH --output-format csv -d $OUT/c -o b -- $R/bench/tfft_bench $ARGS > $OUT/c.log 2>&1 || { tail -5 $OUT/c.log; echo "(pass c failed; continuing)"; }
echo done
