#!/bin/bash
# Interleaved A/B of the 16-bit launch forms. Usage: gpu_ab16.sh <dtype> "<modes>" "<lengths>"
cd "$(dirname "$0")/.."
DT=${1:-fp16}
MODES=${2:-"0 10"}
NS=${3:-"512 1024 2048 4096"}
for rep in 1 2 3; do
  for n in $NS; do
    for m in $MODES; do
      for ip in 1 0; do
        printf "rep=%s n=%s mode=%s inplace=%s " $rep $n $m $ip
        timeout -k 10 60 ./bench/tfft_bench --dtype $DT --n $n --inplace $ip --no-hipfft --mode $m --runs 40 2>&1 | grep thunderfft | awk '{print $(NF-3), $NF}' || exit 1
      done
    done
  done
done
