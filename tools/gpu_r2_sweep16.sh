#!/bin/bash
# 16-bit storage sweep at 2^28 points: fp16 / bf16, in place and out of place, default kernels, with the copy row.
set -u
for dt in fp16 bf16; do for ip in 1 0; do
echo "### $dt inplace=$ip"
timeout -k 10 120 ./bench/tfft_bench --dtype $dt --inplace $ip --no-hipfft 2>&1 | grep -E "thunderfft\(|copy|rror" || exit 1
done; done
