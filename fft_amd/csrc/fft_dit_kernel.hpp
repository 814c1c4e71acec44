// N = 8192 (fp32) as two interleaved 4096-point transforms in one workgroup.
//
// The one-pass plan of this length keeps 32 points per thread and needs 64 KiB of LDS for one exchange, which
// leaves two workgroups per CU: too few independent load / compute / store phases per CU to keep its memory
// requests flowing (72 % of 8 TB/s against 78 % for N = 4096). Here the transform is cut by decimation in time,
//   x[2 m + r] -> F_r[k'] (4096 points each),   X[k' + 4096 k0] = F_0[k'] + (-1)^k0 W_N^k' F_1[k'],
// F_0 and F_1 run one after the other through the 4096-point block transform (fft_block.hpp) and its 34 KiB
// exchange buffer, and the last step is a radix-2 butterfly between registers of one thread. A workgroup keeps
// the 256 threads and the LDS footprint of the N = 4096 kernel, so four workgroups share a CU: 76-77 %, and
// 78-79 % with the scheduler's max-ILP strategy (set for this translation unit in the Makefile).
//   loads  : x[2 m] and x[2 m + 1] are 16 contiguous bytes, one load of the lane that owns column m:
//            de-interleaving is register naming, and the 64 lanes of a wave read one contiguous kilobyte
//   stores : block k0 of the result is 4096 contiguous points in the register order the 4096-point transform
//            leaves, so each block goes out through that transform's own store path
// The same cut does not help N = 16384: four subsequences need 64 points per thread (two 256-thread workgroups
// per CU: 50 %), and a 512-thread form with 32 points per thread is again two workgroups per CU (68 %, level with
// the one-pass plan): at 128 KiB per transform the register file of a CU holds two transforms either way.
// The reference has no transform beyond 4096 (thunderfft/include/thunderfft/detail/thunderfft_impl.cuh:14-32).
#pragma once
#include "fft_block.hpp"

namespace tfft {

// tw: W_N^j, j < N, followed by W_4096^j, j < 4096 (what tfft_initialize uploads for this length).
template <class P, bool FWD, int MINW>
__global__ __launch_bounds__(P::WG, MINW) void fft_dit_kernel(const float4* in, float4* out,
                                                             const float2* __restrict__ tw, long long batch,
                                                             int xcd_log2) {
  static_assert(P::N == 4096 && P::TPB == 1 && P::NB1 == 1 && P::NBL == 1 && P::R1 == P::RL,
                "built on the one-butterfly-per-pass 4096-point plan");
  constexpr int M = P::N, N = 2 * M, E = P::E, C1 = P::C1;
  using IO = StoreF32;
  __shared__ __attribute__((aligned(16))) char lds_raw[P::LDS_BYTES];

  if (blockIdx.x >= batch) return;
  const long long tr = xcd_chunk(blockIdx.x, batch, xcd_log2);
  const BlockFFT<P> f(tw + N, lds_raw);
  const int c = f.b1_c(0);  // this thread's column of both subsequences; also its output column q

  // ---- all rows first: row n1 of column c holds x[2 (c + C1 n1)], x[2 (c + C1 n1) + 1]
  f32x4_t raw[E];
  {
    const f32x4_t* src = reinterpret_cast<const f32x4_t*>(in + tr * (N / 2)) + c;
#pragma unroll
    for (int n1 = 0; n1 < E; ++n1) raw[n1] = __builtin_nontemporal_load(src + n1 * C1);
  }
  const c32 wq = tw[c];  // W_N^q, the base of the final twiddles

  // ---- the subsequences, one after the other through the same exchange buffer
  c32 F0[E], F1[E];
#pragma unroll
  for (int n1 = 0; n1 < E; ++n1) {
    // inverse = forward on swapped parts
    F0[n1] = FWD ? make_float2(raw[n1].x, raw[n1].y) : make_float2(raw[n1].y, raw[n1].x);
    F1[n1] = FWD ? make_float2(raw[n1].z, raw[n1].w) : make_float2(raw[n1].w, raw[n1].z);
  }
  f.template transform<true>(F0, false);
  f.template transform<true>(F1, true);

  // ---- X[q + C1 e + M k0] = F_0 + (-1)^k0 W_N^q W_(N / C1)^e F_1 at k' = q + C1 e
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const c32 t = mul_w<N / C1>(cmul(F1[e], wq), e);
    const c32 x0 = cadd(F0[e], t), x1 = csub(F0[e], t);
    F0[e] = FWD ? x0 : make_float2(x0.y, x0.x);
    F1[e] = FWD ? x1 : make_float2(x1.y, x1.x);
  }

  // ---- both blocks leave like the output of a 4096-point transform
  IO::raw* dst = out + tr * (N / 2);
  f.template store<IO>(dst, 1, F0);
  f.template store<IO>(dst + (M / 2), 1, F1);
}

}  // namespace tfft
