// N = 8192 and 16384 (fp32) as R0 = 2 or 4 interleaved 4096-point transforms in one workgroup.
//
// The one-pass plans of these lengths keep 32 points per thread and need 64 KiB (or twice 32 KiB) of LDS for
// one exchange, which leaves two workgroups per CU: every wave then spends its life issuing its own
// instructions one behind the other, and the CU has too few waves to cover that with another wave's memory
// phase. Here the transform is cut by decimation in time,
//   x[R0 m + r] -> F_r[k'] (4096 points each),   X[k' + 4096 k0] = sum_r W_R0^(r k0) W_N^(r k') F_r[k'],
// the F_r run one after the other through the 4096-point block transform (fft_block.hpp) and its 34 KiB
// exchange buffer, and the last step is a radix-R0 butterfly between registers of one thread. A workgroup keeps
// the 256 threads and the LDS footprint of the N = 4096 kernel, so four (R0 = 2) workgroups share a CU.
//   loads  : the R0 points x[R0 m .. R0 m + R0 - 1] of column m are 8 R0 contiguous bytes, one or two 16-byte
//            loads of the lane that owns column m: de-interleaving is register naming, and the 64 lanes of a
//            wave read one contiguous kilobyte per instruction (R0 = 2)
//   stores : block k0 of the result is 4096 contiguous points in the register order the 4096-point transform
//            leaves, so each block goes out through that transform's own store path
// The reference has no transform beyond 4096 (thunderfft/include/thunderfft/detail/thunderfft_impl.cuh:14-32).
#pragma once
#include "fft_block.hpp"

namespace tfft {

// tw: W_N^j, j < N, followed by W_4096^j, j < 4096 (what tfft_initialize uploads for these lengths).
template <class P, int R0, bool FWD, int MINW>
__global__ __launch_bounds__(P::WG, MINW) void fft_dit_kernel(const float4* in, float4* out,
                                                             const float2* __restrict__ tw, long long batch,
                                                             int xcd_log2) {
  static_assert(P::N == 4096 && P::TPB == 1 && P::NB1 == 1 && P::NBL == 1 && P::R1 == P::RL,
                "built on the one-butterfly-per-pass 4096-point plan");
  static_assert(R0 == 2 || R0 == 4, "N = 8192 or 16384");
  constexpr int M = P::N, N = M * R0, E = P::E, C1 = P::C1;
  using IO = StoreF32;
  __shared__ __attribute__((aligned(16))) char lds_raw[P::LDS_BYTES];

  if (blockIdx.x >= batch) return;
  const long long tr = xcd_chunk(blockIdx.x, batch, xcd_log2);
  const BlockFFT<P> f(tw + N, lds_raw);
  const int c = f.b1_c(0);  // this thread's column of every subsequence; also its output column q

  // ---- all rows first: row n1 of column c holds x[R0 (c + C1 n1) + r], r < R0
  constexpr int LPR = R0 / 2;  // 16-byte loads per row
  f32x4_t raw_a[E], raw_b[LPR == 2 ? E : 1];  // subsequences 0, 1 and 2, 3
  {
    const f32x4_t* src = reinterpret_cast<const f32x4_t*>(in + tr * (N / 2)) + c * LPR;
#pragma unroll
    for (int n1 = 0; n1 < E; ++n1) {
      raw_a[n1] = __builtin_nontemporal_load(src + (n1 * C1) * LPR);
      if constexpr (LPR == 2) raw_b[n1] = __builtin_nontemporal_load(src + (n1 * C1) * LPR + 1);
    }
  }
  const c32 wq = tw[c];  // W_N^q, the base of the final twiddles

  // ---- the subsequences, one after the other through the same exchange buffer. (One array per subsequence and
  // straight-line calls: a two-dimensional array indexed by a loop over r is left in scratch memory for R0 = 4.)
  c32 F0[E], F1[E], F2[R0 == 4 ? E : 1], F3[R0 == 4 ? E : 1];
#define TFFT_DIT_SUBSEQUENCE(r, Fr, lead)                                                         \
  {                                                                                              \
    _Pragma("unroll") for (int n1 = 0; n1 < E; ++n1) {                                           \
      const f32x4_t t = ((r) >> 1) ? raw_b[LPR == 2 ? n1 : 0] : raw_a[n1];                        \
      const c32 v = ((r) & 1) ? make_float2(t.z, t.w) : make_float2(t.x, t.y);                    \
      Fr[n1] = FWD ? v : make_float2(v.y, v.x); /* inverse = forward on swapped parts */         \
    }                                                                                            \
    f.template transform<true>(Fr, lead);                                                        \
  }
  TFFT_DIT_SUBSEQUENCE(0, F0, false)
  TFFT_DIT_SUBSEQUENCE(1, F1, true)
  if constexpr (R0 == 4) {
    TFFT_DIT_SUBSEQUENCE(2, F2, true)
    TFFT_DIT_SUBSEQUENCE(3, F3, true)
  }
#undef TFFT_DIT_SUBSEQUENCE

  // ---- X[q + C1 e + M k0] = sum_r W_R0^(r k0) (W_N^q)^r W_(N / C1)^(r e) F_r[q + C1 e]
  c32 wr[R0];
  wr[1] = wq;
  if constexpr (R0 == 4) {
    wr[2] = csqr(wq);
    wr[3] = cmul(wr[2], wq);
  }
  auto finish = [](c32 v) { return FWD ? v : make_float2(v.y, v.x); };
#pragma unroll
  for (int e = 0; e < E; ++e) {
    c32 x[R0];
    x[0] = F0[e];
    x[1] = mul_w<N / C1>(cmul(F1[e], wr[1]), e);
    if constexpr (R0 == 4) {
      x[2] = mul_w<N / C1>(cmul(F2[e], wr[2]), 2 * e);
      x[3] = mul_w<N / C1>(cmul(F3[e], wr[3]), 3 * e);
    }
    dft<R0>(x);
    F0[e] = finish(x[0]);
    F1[e] = finish(x[1]);
    if constexpr (R0 == 4) {
      F2[e] = finish(x[2]);
      F3[e] = finish(x[3]);
    }
  }

  // ---- every block k0 leaves like the output of a 4096-point transform
  IO::raw* dst = out + tr * (N / 2);
  f.template store<IO>(dst, 1, F0);
  f.template store<IO>(dst + (M / 2), 1, F1);
  if constexpr (R0 == 4) {
    f.template store<IO>(dst + 2 * (M / 2), 1, F2);
    f.template store<IO>(dst + 3 * (M / 2), 1, F3);
  }
}

}  // namespace tfft
