// In-register radix butterflies for gfx950 (wave64, fp32 VALU).
//
// dft<R>(x) replaces x[0..R-1] by its forward DFT, X[k] = sum_n x[n] e^{-2 pi i n k / R},
// natural order in and out. Every index below is a compile-time constant once the loops
// are unrolled, so the arrays live in VGPRs and the small twiddles fold to immediates.
// The sign convention matches the reference's rotate() (thunderfft/include/thunderfft/
// detail/util_arith.h:85-91) and its forward "-i" butterflies
// (detail/kernel/shared_kernel_n256_b16.cuh:69-85).
#pragma once
#include <hip/hip_runtime.h>

#include "tw_consts.hpp"

namespace tfft {

using c32 = float2;

__device__ __forceinline__ c32 cadd(c32 a, c32 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ c32 csub(c32 a, c32 b) { return make_float2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ c32 cmul(c32 a, c32 b) {
  return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ c32 csqr(c32 a) {
  return make_float2(a.x * a.x - a.y * a.y, 2.0f * a.x * a.y);
}

// a * W_R^j with W_R = e^{-2 pi i / R}; j is a constant after unrolling, so the chain of
// tests below folds to one branch and the multiples of pi/4 cost no general multiply.
template <int R>
__device__ __forceinline__ c32 mul_w(c32 a, int j) {
  static_assert(64 % R == 0, "radix must divide 64");
  j &= (R - 1);
  if (j == 0) return a;
  if (4 * j == R) return make_float2(a.y, -a.x);       // * (-i)
  if (2 * j == R) return make_float2(-a.x, -a.y);      // * (-1)
  if (4 * j == 3 * R) return make_float2(-a.y, a.x);   // * (+i)
  constexpr float h = 0.70710678118654752f;
  if (8 * j == R) return make_float2((a.x + a.y) * h, (a.y - a.x) * h);
  if (8 * j == 3 * R) return make_float2((a.y - a.x) * h, -(a.x + a.y) * h);
  if (8 * j == 5 * R) return make_float2(-(a.x + a.y) * h, (a.x - a.y) * h);
  if (8 * j == 7 * R) return make_float2((a.x - a.y) * h, (a.x + a.y) * h);
  const float c = tw_cos64(j * (64 / R));
  const float s = tw_sin64(j * (64 / R));
  return make_float2(a.x * c + a.y * s, a.y * c - a.x * s);
}

template <int R>
__device__ __forceinline__ void dft(c32 (&x)[R]);

template <>
__device__ __forceinline__ void dft<2>(c32 (&x)[2]) {
  const c32 a = x[0], b = x[1];
  x[0] = cadd(a, b);
  x[1] = csub(a, b);
}

template <>
__device__ __forceinline__ void dft<4>(c32 (&x)[4]) {
  const c32 t0 = cadd(x[0], x[2]);
  const c32 t1 = csub(x[0], x[2]);
  const c32 t2 = cadd(x[1], x[3]);
  const c32 t3 = csub(x[1], x[3]);
  const c32 t3r = make_float2(t3.y, -t3.x);  // -i * t3
  x[0] = cadd(t0, t2);
  x[1] = cadd(t1, t3r);
  x[2] = csub(t0, t2);
  x[3] = csub(t1, t3r);
}

// Cooley-Tukey split R = RA * RB inside one thread: RB inner DFTs of size RA over the
// stride-RB subsequences, twiddle, RA outer DFTs of size RB.
template <int R>
__device__ __forceinline__ void dft(c32 (&x)[R]) {
  constexpr int RA = (R == 8 || R == 32) ? 2 : 4;
  constexpr int RB = R / RA;
  c32 t[RA][RB];
#pragma unroll
  for (int n2 = 0; n2 < RB; ++n2) {
    c32 a[RA];
#pragma unroll
    for (int n1 = 0; n1 < RA; ++n1) a[n1] = x[n1 * RB + n2];
    dft<RA>(a);
#pragma unroll
    for (int k1 = 0; k1 < RA; ++k1) t[k1][n2] = mul_w<R>(a[k1], n2 * k1);
  }
#pragma unroll
  for (int k1 = 0; k1 < RA; ++k1) {
    c32 b[RB];
#pragma unroll
    for (int n2 = 0; n2 < RB; ++n2) b[n2] = t[k1][n2];
    dft<RB>(b);
#pragma unroll
    for (int k2 = 0; k2 < RB; ++k2) x[k1 + RA * k2] = b[k2];
  }
}

// pw[k] = w^k for k = 0..R-1 from w alone, by squarings and one-step products so that no
// power is more than ceil(log2 R) multiplications away from w.
template <int R>
__device__ __forceinline__ void twiddle_powers(c32 w, c32 (&pw)[R]) {
  pw[0] = make_float2(1.0f, 0.0f);
  pw[1] = w;
#pragma unroll
  for (int k = 2; k < R; ++k) {
    if ((k & 1) == 0)
      pw[k] = csqr(pw[k / 2]);
    else
      pw[k] = cmul(pw[k - 1], w);
  }
}

// x[j] *= w^j for j = 1..R-1. Up to radix 16 the powers come from one chain; above that a full
// chain would keep R/2 powers alive at once, so j = 8a + b uses two short chains,
// w^j = (w^8)^a * w^b: 16 live powers instead of 32 or 64 for the same number of multiplies.
template <int R>
__device__ __forceinline__ void apply_twiddles(c32 (&x)[R], c32 w) {
  if constexpr (R <= 16) {
    c32 pw[R];
    twiddle_powers<R>(w, pw);
#pragma unroll
    for (int j = 1; j < R; ++j) x[j] = cmul(x[j], pw[j]);
  } else {
    constexpr int RH = R / 8;
    c32 lo[8], hi[RH];
    twiddle_powers<8>(w, lo);
    twiddle_powers<RH>(csqr(lo[4]), hi);
#pragma unroll
    for (int a = 0; a < RH; ++a) {
#pragma unroll
      for (int b = 0; b < 8; ++b) {
        if (a == 0 && b == 0) continue;
        const c32 f = (a == 0) ? lo[b] : (b == 0) ? hi[a] : cmul(hi[a], lo[b]);
        x[8 * a + b] = cmul(x[8 * a + b], f);
      }
    }
  }
}

}  // namespace tfft
