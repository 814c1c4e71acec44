// thff_fft for MI355X: the smallest user of the public template API,
//   ThunderFFTInitialize<float>(N); ThunderFFT_global<float, N, true>(in, out, batch, stream); ThunderFFTFinalize<float>();
// on the reference sample's workload and report (thunderfft/sample/fft1d/main.cu:74-163): N = 64 at compile
// time (-DTHFF_FFT_N=<N> changes it), 65536 transforms, transform b holding a tone of amplitude 0.5 + 0.1 b on
// bin 1 + b plus 0.1 on bin 7 (:16-30), compared with the vendor library (hipFFT here, cuFFT there) by
// relative L2 and L-infinity error, plus a weighted checksum and the first bins of transform 0.
#include <hipfft/hipfft.h>

#include <algorithm>
#include <iomanip>
#include <iostream>

#include "harness.hpp"

#ifndef THFF_FFT_N
#define THFF_FFT_N 64
#endif

int main() {
  using harness::cplx;
  constexpr unsigned N = THFF_FFT_N;
  const unsigned batch = 65536;
  const int device = 0;
  harness::hip_ok(hipSetDevice(device), "hipSetDevice");
  std::cout << "ThunderFFT vs hipFFT\n"
            << "  Type   : float/float2\n"
            << "  N      : " << N << " (compile-time)\n"
            << "  batch  : " << batch << "\n"
            << "  device : " << device << "\n";

  // bin-aligned tones, evaluated in fp64 and rounded once
  std::vector<float2> tones((size_t)N * batch);
  for (unsigned b = 0; b < batch; ++b)
    for (unsigned i = 0; i < N; ++i) {
      const double t = 2.0 * M_PI * (double)i / N;
      const cplx v = std::polar(0.5 + 0.1 * b, t * (double)(1 + b)) + std::polar(0.1, t * 7.0);
      tones[(size_t)b * N + i] = make_float2((float)v.real(), (float)v.imag());
    }
  harness::DeviceArray<float2> d_in, d_ours(tones.size()), d_vendor(tones.size());
  d_in.upload(tones);

  thunderfft::ThunderFFTInitialize<float>(N);
  // every call returns after the stream has drained, like the reference's launcher
  const float ms = harness::mean_ms([&] { thunderfft::ThunderFFT_global<float, N, true>(d_in.get(), d_ours.get(), batch, 0); }, 1, 5);

  hipfftHandle plan{};
  if (hipfftPlan1d(&plan, (int)N, HIPFFT_C2C, (int)batch) != HIPFFT_SUCCESS ||
      hipfftExecC2C(plan, reinterpret_cast<hipfftComplex*>(d_in.get()), reinterpret_cast<hipfftComplex*>(d_vendor.get()),
                    HIPFFT_FORWARD) != HIPFFT_SUCCESS)
    harness::die("hipFFT failed");
  harness::hip_ok(hipDeviceSynchronize(), "hipDeviceSynchronize");
  const std::vector<float2> ours = d_ours.download(), vendor = d_vendor.download();

  harness::ErrorSums err;
  long double checksum = 0.0L;
  for (size_t i = 0; i < ours.size(); ++i) {
    err.add(harness::to_cplx(ours[i]), harness::to_cplx(vendor[i]));
    checksum += (long double)ours[i].x * 1.0001L + (long double)ours[i].y * 0.9997L;
  }
  std::cout << std::fixed << std::setprecision(6) << "ThunderFFT avg time : " << ms << " ms\n"
            << "Checksum(ThunderFFT): " << (double)checksum << "\n"
            << "Compare vs hipFFT   : L2_rel=" << std::setprecision(3) << std::scientific << err.l2_rel()
            << "  Linf_rel=" << err.linf_rel() << std::fixed << std::setprecision(6) << "\n";
  const unsigned shown = std::min<unsigned>(8, N);
  std::cout << "First " << shown << " bins (batch 0):\n";
  for (unsigned k = 0; k < shown; ++k)
    std::cout << "  k=" << std::setw(3) << k << "  TF=(" << ours[k].x << ", " << ours[k].y << ")"
              << "  hipFFT=(" << vendor[k].x << ", " << vendor[k].y << ")\n";

  thunderfft::ThunderFFTFinalize<float>();
  hipfftDestroy(plan);
  return 0;
}
