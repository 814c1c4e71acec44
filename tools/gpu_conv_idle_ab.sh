#!/bin/bash
# The fused-convolution plans under the reference driver's own protocol (one process per call, 5 warm-up + 10 timed runs,
# filter transform inside the timed region): plan 0 = the default (runs of signals with LDS-DMA staging, RUN from the
# problem size), 4 = one workgroup per signal, 5 / 6 = staged with runs of 2 / 8. Three alternating rounds.
C=./bench/thff_conv1d_fullfft4096
for i in 1 2 3; do for m in 0 4 5 6; do
  sleep 0.5
  t=$(TFFT_MODE=$m timeout -k 10 60 $C 33 0 64 768 5 10 1 0 1 1 1 | grep -o "time: [0-9.]* ms") || exit 1
  echo "round $i plan $m  5+10 per process: $t"
done; done
