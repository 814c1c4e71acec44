// Minimal forward-transform sample: the MI355X build of the reference's thff_fft
// (thunderfft/sample/fft1d/main.cu:74-163). Same call sequence on the same template API
// (ThunderFFTInitialize<float>(N), ThunderFFT_global<float, N, true>(d_in, d_out, batch, stream),
// ThunderFFTFinalize<float>()), same bin-aligned test tones (:16-30), checksum (:35-43) and report
// lines; hipFFT stands where the reference compares with cuFFT (:119-128).
// N is a compile-time constant as in the reference (:76): build with -DTHFF_FFT_N=<N> to change it.
#include <hipfft/hipfft.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iomanip>
#include <iostream>
#include <vector>

#include <thunderfft/thunderfft.cuh>

#ifndef THFF_FFT_N
#define THFF_FFT_N 64
#endif

int main() {
  constexpr unsigned N = THFF_FFT_N;
  const unsigned batch = 65536;
  const int device = 0;
  CHECK_HIP(hipSetDevice(device));
  std::cout << "ThunderFFT vs hipFFT\n"
            << "  Type   : float/float2\n"
            << "  N      : " << N << " (compile-time)\n"
            << "  batch  : " << batch << "\n"
            << "  device : " << device << "\n";

  const size_t count = size_t(N) * batch;
  const size_t bytes = sizeof(float2) * count;
  std::vector<float2> h_in(count), h_out(count), h_ref(count);
  // transform b: a tone of amplitude 0.5 + 0.1 b on bin 1 + b, plus 0.1 on bin 7
  const float two_pi = 2.0f * float(M_PI);
  for (unsigned b = 0; b < batch; ++b) {
    const float f0 = float(1 + b), amp = 0.5f + 0.1f * float(b);
    for (unsigned i = 0; i < N; ++i) {
      const float t = float(i) / float(N);
      h_in[size_t(b) * N + i] = make_float2(amp * std::cos(two_pi * f0 * t) + 0.1f * std::cos(two_pi * 7.0f * t),
                                            amp * std::sin(two_pi * f0 * t) + 0.1f * std::sin(two_pi * 7.0f * t));
    }
  }

  float2 *d_in = nullptr, *d_out = nullptr, *d_ref = nullptr;
  CHECK_HIP(hipMalloc(&d_in, bytes));
  CHECK_HIP(hipMalloc(&d_out, bytes));
  CHECK_HIP(hipMalloc(&d_ref, bytes));
  CHECK_HIP(hipMemcpy(d_in, h_in.data(), bytes, hipMemcpyHostToDevice));

  thunderfft::ThunderFFTInitialize<float>(N);
  // one warm-up, the mean of five calls (each returns after the stream has drained, like the reference's)
  const unsigned warm_up_runs = 1, runs = 5;
  hipEvent_t e0, e1;
  CHECK_HIP(hipEventCreate(&e0));
  CHECK_HIP(hipEventCreate(&e1));
  for (unsigned i = 0; i < warm_up_runs; ++i) thunderfft::ThunderFFT_global<float, N, true>(d_in, d_out, batch, 0);
  CHECK_HIP(hipDeviceSynchronize());
  CHECK_HIP(hipEventRecord(e0));
  for (unsigned i = 0; i < runs; ++i) thunderfft::ThunderFFT_global<float, N, true>(d_in, d_out, batch, 0);
  CHECK_HIP(hipEventRecord(e1));
  CHECK_HIP(hipDeviceSynchronize());
  float ms_tf = 0.f;
  CHECK_HIP(hipEventElapsedTime(&ms_tf, e0, e1));
  ms_tf /= float(runs);
  CHECK_HIP(hipMemcpy(h_out.data(), d_out, bytes, hipMemcpyDeviceToHost));

  hipfftHandle plan{};
  if (hipfftPlan1d(&plan, int(N), HIPFFT_C2C, int(batch)) != HIPFFT_SUCCESS ||
      hipfftExecC2C(plan, reinterpret_cast<hipfftComplex*>(d_in), reinterpret_cast<hipfftComplex*>(d_ref),
                    HIPFFT_FORWARD) != HIPFFT_SUCCESS) {
    std::fprintf(stderr, "hipFFT failed\n");
    return EXIT_FAILURE;
  }
  CHECK_HIP(hipDeviceSynchronize());
  CHECK_HIP(hipMemcpy(h_ref.data(), d_ref, bytes, hipMemcpyDeviceToHost));

  long double num = 0.0L, den = 0.0L, max_diff = 0.0L, max_ref = 0.0L, acc = 0.0L;
  for (size_t i = 0; i < count; ++i) {
    const long double dx = (long double)h_out[i].x - h_ref[i].x, dy = (long double)h_out[i].y - h_ref[i].y;
    num += dx * dx + dy * dy;
    den += (long double)h_ref[i].x * h_ref[i].x + (long double)h_ref[i].y * h_ref[i].y;
    max_diff = std::max(max_diff, std::hypot(dx, dy));
    max_ref = std::max(max_ref, std::hypot((long double)h_ref[i].x, (long double)h_ref[i].y));
    acc += (long double)h_out[i].x * 1.0001L + (long double)h_out[i].y * 0.9997L;
  }
  const double err_l2 = den > 0 ? (double)std::sqrt((double)(num / den)) : (double)std::sqrt((double)num);
  const double err_linf = max_ref > 0 ? (double)(max_diff / max_ref) : (double)max_diff;

  std::cout << std::fixed << std::setprecision(6);
  std::cout << "ThunderFFT avg time : " << ms_tf << " ms\n";
  std::cout << "Checksum(ThunderFFT): " << (double)acc << "\n";
  std::cout << "Compare vs hipFFT   : L2_rel=" << std::setprecision(3) << std::scientific << err_l2
            << "  Linf_rel=" << err_linf << std::fixed << std::setprecision(6) << "\n";
  const unsigned to_print = std::min<unsigned>(8, N);
  std::cout << "First " << to_print << " bins (batch 0):\n";
  for (unsigned i = 0; i < to_print; ++i)
    std::cout << "  k=" << std::setw(3) << i << "  TF=(" << h_out[i].x << ", " << h_out[i].y << ")"
              << "  hipFFT=(" << h_ref[i].x << ", " << h_ref[i].y << ")\n";

  thunderfft::ThunderFFTFinalize<float>();
  hipfftDestroy(plan);
  CHECK_HIP(hipFree(d_in));
  CHECK_HIP(hipFree(d_out));
  CHECK_HIP(hipFree(d_ref));
  return 0;
}
