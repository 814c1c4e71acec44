// Smallest host program on the public API, in the call order of the reference's fft1d sample
// (thunderfft/sample/fft1d/main.cu:110-118, 155): Initialize, one batched forward transform, Finalize.
// Input: one complex tone per transform, so the spectrum is a single bin of height N.
#include <hip/hip_runtime.h>
#include <thunderfft/thunderfft.cuh>

#include <cmath>
#include <cstdio>
#include <vector>

int main() {
  constexpr int N = 1024;
  const int batch = 64;
  std::vector<float2> h(static_cast<size_t>(N) * batch);
  for (int b = 0; b < batch; ++b)
    for (int n = 0; n < N; ++n) {
      const double a = 2.0 * M_PI * ((b + 1) % N) * n / N;
      h[static_cast<size_t>(b) * N + n] = make_float2(static_cast<float>(std::cos(a)), static_cast<float>(std::sin(a)));
    }
  float2 *d_in = nullptr, *d_out = nullptr;
  if (hipMalloc(&d_in, h.size() * sizeof(float2)) != hipSuccess || hipMalloc(&d_out, h.size() * sizeof(float2)) != hipSuccess) {
    std::fprintf(stderr, "hipMalloc failed\n");
    return 2;
  }
  (void)hipMemcpy(d_in, h.data(), h.size() * sizeof(float2), hipMemcpyHostToDevice);
  thunderfft::ThunderFFTInitialize<float>(N);
  thunderfft::ThunderFFT_global<float, N, true>(d_in, d_out, batch);
  (void)hipDeviceSynchronize();
  (void)hipMemcpy(h.data(), d_out, h.size() * sizeof(float2), hipMemcpyDeviceToHost);
  thunderfft::ThunderFFTFinalize<float>();
  double worst = 0.0;
  for (int b = 0; b < batch; ++b)
    for (int k = 0; k < N; ++k) {
      const float2 v = h[static_cast<size_t>(b) * N + k];
      const double want = k == (b + 1) % N ? N : 0.0;
      worst = std::fmax(worst, std::hypot(v.x - want, static_cast<double>(v.y)));
    }
  std::printf("consumer: N=%d batch=%d max_abs_err=%.3e\n", N, batch, worst);
  (void)hipFree(d_in);
  (void)hipFree(d_out);
  return worst < 1e-2 ? 0 : 1;
}
