set -u
# A/B of two builds of the library (ab/libthunderfft_hip_old.so against the tree's; tools/gpu_ab_misc.sh says how to make the old
# one) on the fused convolution at the Figure-5 shape: real pairs three times over, then complex and real-in once each.
for i in 1 2 3; do
  for lib in ab/libthunderfft_hip_old.so fft_amd/lib/libthunderfft_hip.so; do
    echo "### $lib"
    THUNDERFFT_HIP_LIB=$PWD/$lib timeout -k 10 120 python tools/gpu_conv_ab.py --kind 2 --plans 0,4 --rounds 6 --burst 50 | grep "ms" || exit 1
  done
done
for k in 0 1; do for lib in ab/libthunderfft_hip_old.so fft_amd/lib/libthunderfft_hip.so; do
    echo "### $lib"
    THUNDERFFT_HIP_LIB=$PWD/$lib timeout -k 10 120 python tools/gpu_conv_ab.py --kind $k --plans 0,4 --rounds 6 --burst 50 | grep "ms" || exit 1
done; done
