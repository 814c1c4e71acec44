#!/bin/bash
# Runs of 2^L consecutive chunks per XCD (blockIdx.x mod 8 is the XCD of a workgroup): sweep of L, three repeats.
# Usage: gpu_xcd.sh "<L values>" <tfft_bench arguments>
cd "$(dirname "$0")/.."
LS=$1; shift
for rep in 1 2 3; do
for L in $LS; do
  printf "rep=$rep L=$L "
  timeout -k 10 120 ./bench/tfft_bench --no-hipfft --runs 30 --xcd $L "$@" | grep thunderfft | awk '{print $2, $NF}' | tr '\n' ' '; echo
done
done
