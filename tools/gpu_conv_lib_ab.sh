set -u
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
