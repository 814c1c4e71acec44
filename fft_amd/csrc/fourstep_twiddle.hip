// Twiddle step of a four-step transform whose two halves run as separate batched calls (the
// distributed path: local FFTs, all-to-all, local FFTs). Element (r, c) of a row-major block is
// multiplied by W_n^((row0 + r) * (col0 + c)); the angle index is reduced mod n in integers and
// the root evaluated in fp64 (sincospi), so the factor is correctly rounded for any n <= 2^40.
// HBM-bound: one read and one write of every point. No counterpart in the reference, which has
// no transform longer than one block (thunderfft/include/thunderfft/detail/thunderfft_impl.cuh:14-32).
#include <hip/hip_runtime.h>

#include "../../include/thunderfft_c.h"

namespace {

__global__ __launch_bounds__(256) void fourstep_twiddle_f32(float4* data, long long rows, long long cols,
                                                            unsigned long long row0, unsigned long long col0,
                                                            unsigned long long n, int forward) {
  const long long pairs_per_row = cols / 2;
  const long long total = rows * pairs_per_row;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
    const long long r = i / pairs_per_row;
    const long long c = (i - r * pairs_per_row) * 2;
    float4 v = data[i];
    const unsigned long long q = row0 + (unsigned long long)r;
    float out[4];
    const float in[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
    for (int k = 0; k < 2; ++k) {
      const unsigned long long j = (q * (col0 + (unsigned long long)c + k)) % n;
      double s, co;
      sincospi(-2.0 * (double)j / (double)n, &s, &co);
      if (!forward) s = -s;
      const float wr = (float)co, wi = (float)s;
      out[2 * k] = in[2 * k] * wr - in[2 * k + 1] * wi;
      out[2 * k + 1] = in[2 * k] * wi + in[2 * k + 1] * wr;
    }
    data[i] = make_float4(out[0], out[1], out[2], out[3]);
  }
}

}  // namespace

extern "C" int tfft_fourstep_twiddle_launch(void* d_data, long long rows, long long cols, long long row0,
                                            long long col0, long long n, int forward, void* stream) {
  const long long total = rows * (cols / 2);
  long long grid = (total + 255) / 256;
  if (grid > 256 * 64) grid = 256 * 64;
  if (grid < 1) grid = 1;
  fourstep_twiddle_f32<<<dim3((unsigned)grid), dim3(256), 0, static_cast<hipStream_t>(stream)>>>(
      static_cast<float4*>(d_data), rows, cols, (unsigned long long)row0, (unsigned long long)col0,
      (unsigned long long)n, forward);
  return (int)hipGetLastError();
}
