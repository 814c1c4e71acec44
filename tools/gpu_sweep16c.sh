#!/bin/bash
# 16-bit matrix-core kernels: launch forms (resident / one workgroup per tile; table / incremental pass-1 twiddles).
cd "$(dirname "$0")/.."
DT=${1:-fp16}
for n in 256 512 1024 2048 4096; do
  for ip in 1 0; do
    for m in 0 6 10 12; do
      echo "== n=$n inplace=$ip mode $m"
      timeout -k 10 60 ./bench/tfft_bench --dtype $DT --n $n --inplace $ip --no-hipfft --mode $m 2>&1 | grep thunderfft || exit 1
    done
  done
done
