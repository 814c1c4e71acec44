// How accurate v_sin_f32 / v_cos_f32 are on exact arguments k / n turns (n a power of two): maximum absolute error of
// (cos, sin) against fp64 over all k < n, for the table sizes the kernels use. The 16-bit matrix-core kernels and the
// fused convolution evaluate part of their twiddles this way (fft_mfma_kernel.hpp, fft_conv_mfma.hpp); the fp32
// kernels read tables rounded once from fp64 (error <= 2^-25 = 3e-8 per component).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

__global__ void probe(float2* out, int n) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  const float turns = (float)k * (1.0f / (float)n);
  out[k] = make_float2(__builtin_amdgcn_cosf(turns), __builtin_amdgcn_sinf(turns));
}

int main() {
  for (int lg : {4, 6, 8, 12, 15, 20, 24}) {
    const int n = 1 << lg;
    float2* d = nullptr;
    if (hipMalloc(&d, sizeof(float2) * n) != hipSuccess) return 1;
    probe<<<(n + 255) / 256, 256>>>(d, n);
    std::vector<float2> h(n);
    if (hipMemcpy(h.data(), d, sizeof(float2) * n, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    double worst = 0, worst_tab = 0;
    int at = 0;
    for (int k = 0; k < n; ++k) {
      const double a = 2.0 * M_PI * (double)k / (double)n;
      const double e = fmax(fabs((double)h[k].x - cos(a)), fabs((double)h[k].y - sin(a)));
      if (e > worst) { worst = e; at = k; }
      worst_tab = fmax(worst_tab, fmax(fabs((double)(float)cos(a) - cos(a)), fabs((double)(float)sin(a) - sin(a))));
    }
    printf("n = 2^%-2d  max |error| evaluated %.3e (at k = %d)   table rounded from fp64 %.3e\n", lg, worst, at, worst_tab);
    (void)hipFree(d);
  }
  return 0;
}
