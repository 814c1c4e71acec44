#!/bin/bash
# bench.py --n 16384 with two builds of the library in turn (ab/libthunderfft_hip_old.so against the tree's; see
# tools/gpu_ab_misc.sh for how to make the old build), at 3 + 20 and at 50 + 100 launches, three rounds.
for i in 1 2 3; do
  for lib in ab/libthunderfft_hip_old.so fft_amd/lib/libthunderfft_hip.so; do
    for sw in "20 3" "100 50"; do set -- $sw
      THUNDERFFT_HIP_LIB=$PWD/$lib timeout -k 10 120 python bench.py --n 16384 --steps $1 --warmup $2 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib'.split('/')[-1][:24].ljust(26), 'steps=$1', 'frac', d['roofline']['frac'], 'kernel_ms', d['roofline']['kernel_ms'], 'ms_per_step', d['ms_per_step'])" || exit 1
    done
  done
done
