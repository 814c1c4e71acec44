#!/bin/bash
# Interleaved A/B of bench/tfft_bench configurations: every combination of length, plan, XCD run length and
# in-place flag is run in turn, REPS times over, so that box-to-box and minute-to-minute drift hits all alike.
#   DT=fp16 NS="512 1024 2048" MODES="0 10" XCDS="-" INPLACE="1 0" REPS=3 tools/gpu_ab.sh [more tfft_bench arguments]
# MODES: tfft_exec_options.plan values; XCDS: run lengths log2 ("-" = the kernel's default). One line per run:
# the thunderfft row's time and % of 8 TB/s.
set -u
cd "$(dirname "$0")/.."
DT=${DT:-fp32}; NS=${NS:-4096}; MODES=${MODES:-0}; XCDS=${XCDS:--}; INPLACE=${INPLACE:-0}; REPS=${REPS:-3}
for rep in $(seq 1 $REPS); do
  for n in $NS; do for m in $MODES; do for x in $XCDS; do for ip in $INPLACE; do
    xarg=""; [ "$x" != "-" ] && xarg="--xcd $x"
    printf "rep=%s dtype=%s n=%s mode=%s xcd=%s inplace=%s " $rep $DT $n $m $x $ip
    timeout -k 10 120 ./bench/tfft_bench --dtype $DT --n $n --mode $m $xarg --inplace $ip --no-hipfft --runs ${RUNS:-30} "$@" 2>&1 \
      | grep "thunderfft(" | awk '{print $(NF-3), $NF}' || exit 1
  done; done; done; done
done
