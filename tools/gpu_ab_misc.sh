#!/bin/bash
# A/B of two library builds (ab/old/libthunderfft_hip.so against the tree's) on the kernels outside the fp32 batched
# sweep: 16-bit N = 8192 / 16384, multi-pass lengths, the overlap-save convolution binary. Make the old build first:
#   git stash (or git worktree add /tmp/old <commit>), make -j lib there, copy its fft_amd/lib/libthunderfft_hip.so to
#   ab/old/libthunderfft_hip.so and to ab/libthunderfft_hip_old.so (ab/ is git-ignored and travels with gpurun).
set -u
for dt in fp16 bf16; do
  timeout -k 10 300 python tools/gpu_lib_ab.py --dtype $dt --lengths 4096,8192,16384 --groups 10 --burst 10 --passes 2 2>&1 | grep -v amdgpu.ids || exit 1
done
timeout -k 10 400 python tools/gpu_lib_ab.py --lengths 65536,1048576,4194304,16777216,67108864 --groups 10 --burst 5 --passes 2 2>&1 | grep -v amdgpu.ids || exit 1
for i in 1 2; do
  echo "old:"; LD_LIBRARY_PATH=$PWD/ab/old timeout -k 10 60 ./bench/thff_conv1d 16777216 9 0 5 20 3 | grep -i "time\|L2_rel" || exit 1
  echo "new:"; timeout -k 10 60 ./bench/thff_conv1d 16777216 9 0 5 20 3 | grep -i "time\|L2_rel" || exit 1
done
