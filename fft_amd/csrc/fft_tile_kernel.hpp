// Strided tile FFT: the building block of the multi-pass ("four-step") path for transforms that
// do not fit one workgroup's LDS. The reference stops at N = 4096 (one block per transform,
// thunderfft/include/thunderfft/detail/thunderfft_impl.cuh:14-32) and has no such path.
//
// A large transform N = n1*n2[*n3] is evaluated as 2 or 3 sweeps over HBM. Every sweep runs
// `TPB` short transforms of length n = R1*R2 per workgroup on a TILE of the data:
//
//   column mode : the TPB transforms are adjacent columns of a strided matrix,
//                 address(t, e) = base + e*stride + t   (t fastest: 8*TPB contiguous bytes per row)
//   row mode    : each transform is a contiguous run,
//                 address(t, e) = base + t*stride + e
//
// chosen independently for input and output, so the last sweep can read contiguous rows and
// write its results transposed into natural order. A sweep may multiply its outputs by the
// four-step twiddle W_M^(q*k) (q = the transform's column index in the enclosing length-M
// transform, k = output index); W_M^j comes from a two-level table W_M^j = hi[j >> s] * lo[j & mask].
//
// Inside the workgroup the structure is that of fft_lds_kernel.hpp: E points per thread, one
// radix-R1 pass, one LDS exchange (layout [element][transform], padded), one radix-R2 pass;
// 16-byte global accesses by pairing lanes l / l+32 on adjacent transforms (column mode) or
// adjacent elements (row mode) and trading halves with v_permlane32_swap.
#pragma once
#include "fft_lds_kernel.hpp"

namespace tfft {

struct TileArgs {
  const float4* in;
  float4* out;
  const float2* tw;     // W_n^j, j < n: inter-pass twiddles of the short transform
  const float2* tw_hi;  // four-step twiddle, high part: W_M^(j << tw_shift)
  const float2* tw_lo;  // low part: W_M^j, j < 2^tw_shift
  int tw_shift;
  unsigned long long tw_mmask;  // M - 1
  long long ntiles;
  int na, nb;  // tile id = (bi * na + a) * nb + b
  long long in_bi, in_a, in_b, in_stride;      // element units
  long long out_bi, out_a, out_b, out_stride;
  long long q_a, q_b;  // column index of transform 0 of the tile: a*q_a + b*q_b
  int xcd_log2;        // runs of 2^xcd_log2 consecutive tiles per XCD (xcd_chunk, fft_lds_kernel.hpp)
};

enum : int { kTileInCol = 1, kTileOutCol = 2, kTileTwiddle = 4 };

template <int N_, int E_, int R1_, int R2_, int TPB_, int PS_, int PW_, int MINW_ = 1, int FLAGS_ = 0>
struct TilePlan {
  static constexpr int N = N_, E = E_, R1 = R1_, R2 = R2_, TPB = TPB_, PS = PS_, PW = PW_, MINW = MINW_;
  static constexpr bool SPLIT = (FLAGS_ & kFlagSplit) != 0;
  static constexpr int WG = TPB * N / E;
  static constexpr int C1 = N / R1, CL = N / R2;  // C1 = R2, CL = R1
  static constexpr int NB1 = E / R1, NBL = E / R2;
  static constexpr int NRAW = E / 2;
  static constexpr int TOTAL = N * TPB;
  static constexpr int pad(int i) { return PW ? i + (i >> PS) * PW : i; }
  static constexpr int LDS_ELEMS = pad(TOTAL - 1) + 1;
  static constexpr int LDS_BYTES = LDS_ELEMS * (SPLIT ? 4 : 8);
  static_assert(R1 * R2 == N, "two-pass plan");
  static_assert(WG % 64 == 0 && WG <= 1024, "workgroup must be whole waves");
  static_assert(TPB % 2 == 0, "lanes pair up adjacent transforms");
  static_assert(NB1 == 1 || NB1 % 2 == 0, "first pass needs 1 or an even number of butterflies");
  static_assert(NBL == 1 || NBL % 2 == 0, "last pass needs 1 or an even number of butterflies");
};

template <class P, bool FWD, int IOMODE>
__global__ __launch_bounds__(P::WG, P::MINW) void fft_tile_kernel(const TileArgs a) {
  using IO = StoreF32;
  using raw = IO::raw;
  constexpr int N = P::N, E = P::E, R1 = P::R1, R2 = P::R2, TPB = P::TPB, WG = P::WG;
  constexpr int C1 = P::C1, CL = P::CL, NB1 = P::NB1, NBL = P::NBL, NRAW = P::NRAW;
  constexpr bool IN_COL = (IOMODE & kTileInCol) != 0, OUT_COL = (IOMODE & kTileOutCol) != 0;
  constexpr bool TWIDDLE = (IOMODE & kTileTwiddle) != 0;
  constexpr bool SPLIT = P::SPLIT;

  __shared__ __attribute__((aligned(16))) char lds_raw[P::LDS_BYTES];
  float2* const lds2 = reinterpret_cast<float2*>(lds_raw);
  float* const lds1 = reinterpret_cast<float*>(lds_raw);

  const int tid = threadIdx.x;
  const int hi = (tid >> 5) & 1;

  // ---- tile coordinates
  if (blockIdx.x >= a.ntiles) return;
  const long long tile = xcd_chunk(blockIdx.x, a.ntiles, a.xcd_log2);
  const long long bi = tile / ((long long)a.na * a.nb);
  const int rem = (int)(tile - bi * ((long long)a.na * a.nb));
  const int ta = rem / a.nb, tb = rem - ta * a.nb;
  const long long in_base = bi * a.in_bi + ta * a.in_a + tb * a.in_b;
  const long long out_base = bi * a.out_bi + ta * a.out_a + tb * a.out_b;

  // ---- first pass butterflies (tr, c) and their loads. Lanes run along the contiguous axis.
  constexpr int NG1 = (NB1 == 1) ? 1 : NB1 / 2;
  constexpr int LPG1 = NRAW / NG1;
  int b1_tr[NB1], b1_c[NB1];
  // Offsets inside a tile stay below 2^31 points (the longest enclosing transform is 2^30), so
  // they are kept in 32 bits; only the tile base is 64-bit.
  unsigned ld_base[NG1];  // point offset of load 0 of the group (both points of the pair)
  const unsigned in_stride = (unsigned)a.in_stride, out_stride = (unsigned)a.out_stride;
  auto decode1 = [&](int g, int& tr, int& c) {
    if constexpr (IN_COL) { tr = g % TPB; c = g / TPB; } else { tr = g / C1; c = g % C1; }
  };
  auto in_addr = [&](int tr, unsigned e) -> unsigned {
    return IN_COL ? e * in_stride + (unsigned)tr : (unsigned)tr * in_stride + e;
  };
  if constexpr (NB1 == 1) {
    decode1(pair_swizzle(tid), b1_tr[0], b1_c[0]);
    const int tr0 = IN_COL ? (b1_tr[0] & ~1) : b1_tr[0];
    const int c0 = IN_COL ? b1_c[0] : (b1_c[0] & ~1);
    ld_base[0] = in_addr(tr0, (unsigned)(hi * (R1 / 2) * C1 + c0));
  } else {
#pragma unroll
    for (int pi = 0; pi < NB1 / 2; ++pi) {
      const int g = 2 * (tid + WG * pi);
      decode1(g, b1_tr[2 * pi], b1_c[2 * pi]);
      decode1(g + 1, b1_tr[2 * pi + 1], b1_c[2 * pi + 1]);
      ld_base[pi] = in_addr(b1_tr[2 * pi], (unsigned)b1_c[2 * pi]);
    }
  }
  const unsigned row_step = IN_COL ? (unsigned)C1 * in_stride : (unsigned)C1;  // next row of the same column
  raw pre[NRAW];
  {
    const raw* src = a.in + (in_base >> 1);
#pragma unroll
    for (int g = 0; g < NG1; ++g)
#pragma unroll
      for (int i = 0; i < LPG1; ++i) pre[g * LPG1 + i] = IO::load(src + ((ld_base[g] + (unsigned)i * row_step) >> 1));
  }

  // ---- last pass butterflies (tr, q) and their stores
  constexpr int NGL = (NBL == 1) ? 1 : NBL / 2;
  constexpr int SPGL = NRAW / NGL;
  int bl_tr[NBL], bl_q[NBL];
  unsigned st_base[NGL];
  auto decodeL = [&](int g, int& tr, int& q) {
    if constexpr (OUT_COL) { tr = g % TPB; q = g / TPB; } else { tr = g / CL; q = g % CL; }
  };
  auto out_addr = [&](int tr, unsigned e) -> unsigned {
    return OUT_COL ? e * out_stride + (unsigned)tr : (unsigned)tr * out_stride + e;
  };
  if constexpr (NBL == 1) {
    decodeL(pair_swizzle(tid), bl_tr[0], bl_q[0]);
    const int tr0 = OUT_COL ? (bl_tr[0] & ~1) : bl_tr[0];
    const int q0 = OUT_COL ? bl_q[0] : (bl_q[0] & ~1);
    st_base[0] = out_addr(tr0, (unsigned)(hi * (R2 / 2) * CL + q0));
  } else {
#pragma unroll
    for (int pi = 0; pi < NBL / 2; ++pi) {
      const int g = 2 * (tid + WG * pi);
      decodeL(g, bl_tr[2 * pi], bl_q[2 * pi]);
      decodeL(g + 1, bl_tr[2 * pi + 1], bl_q[2 * pi + 1]);
      st_base[pi] = out_addr(bl_tr[2 * pi], (unsigned)bl_q[2 * pi]);
    }
  }
  const unsigned out_row_step = OUT_COL ? (unsigned)CL * out_stride : (unsigned)CL;

  // ---- inter-pass twiddle bases W_n^c
  c32 w1[NB1];
#pragma unroll
  for (int b = 0; b < NB1; ++b) w1[b] = a.tw[b1_c[b]];

  // ---- four-step twiddles of the last pass: output k_out = q + CL*k of column qq gets
  // W_M^(qq*k_out) = W_M^(qq*q) * (W_M^(qq*CL))^k
  c32 fs_base[NBL], fs_step[NBL];
  if constexpr (TWIDDLE) {
    auto big_w = [&](unsigned long long j) -> c32 {
      j &= a.tw_mmask;
      return cmul(a.tw_hi[j >> a.tw_shift], a.tw_lo[j & ((1ull << a.tw_shift) - 1)]);
    };
    const unsigned long long q0 = (unsigned long long)(ta * a.q_a + tb * a.q_b);
#pragma unroll
    for (int b = 0; b < NBL; ++b) {
      const unsigned long long qq = q0 + (unsigned long long)bl_tr[b];
      fs_base[b] = big_w(qq * (unsigned long long)bl_q[b]);
      fs_step[b] = big_w(qq * (unsigned long long)CL);
    }
  }

  // ---- LDS slots: logical index (element i, transform tr) -> i*TPB + tr, padded
  auto slot = [&](int tr, int i) { return pad_idx<P::PS, P::PW>(i * TPB + tr); };

  // ---- unpack
  c32 v[E];
  if constexpr (NB1 == 1) {
#pragma unroll
    for (int i = 0; i < NRAW; ++i) {
      raw r = pre[i];
      IO::swap_halves(r);
      IO::unpack(r, v[i], v[i + R1 / 2]);
    }
  } else {
#pragma unroll
    for (int pi = 0; pi < NB1 / 2; ++pi)
#pragma unroll
      for (int n1 = 0; n1 < R1; ++n1) IO::unpack(pre[pi * R1 + n1], v[(2 * pi) * R1 + n1], v[(2 * pi + 1) * R1 + n1]);
  }
  if constexpr (!FWD) {
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = make_float2(v[e].y, v[e].x);
  }

  // ---- first pass
#pragma unroll
  for (int b = 0; b < NB1; ++b) {
    c32 x[R1];
#pragma unroll
    for (int j = 0; j < R1; ++j) x[j] = v[b * R1 + j];
    dft<R1>(x);
    c32 pw[R1];
    twiddle_powers<R1>(w1[b], pw);
#pragma unroll
    for (int j = 1; j < R1; ++j) x[j] = cmul(x[j], pw[j]);
#pragma unroll
    for (int j = 0; j < R1; ++j) v[b * R1 + j] = x[j];
  }

  // ---- exchange: writer (tr, c) element k1 -> i = k1*C1 + c; reader (tr, q) element n2 -> i = q*C1 + n2
  auto wslot = [&](int e) { return slot(b1_tr[e / R1], (e % R1) * C1 + b1_c[e / R1]); };
  auto rslot = [&](int e) { return slot(bl_tr[e / R2], bl_q[e / R2] * C1 + (e % R2)); };
  if constexpr (!SPLIT) {
#pragma unroll
    for (int e = 0; e < E; ++e) lds2[wslot(e)] = v[e];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = lds2[rslot(e)];
  } else {
    float re[E];
#pragma unroll
    for (int e = 0; e < E; ++e) lds1[wslot(e)] = v[e].x;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e) re[e] = lds1[rslot(e)];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e) lds1[wslot(e)] = v[e].y;
    __syncthreads();
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = make_float2(re[e], lds1[rslot(e)]);
  }

  // ---- last pass (+ four-step twiddle)
#pragma unroll
  for (int b = 0; b < NBL; ++b) {
    c32 x[R2];
#pragma unroll
    for (int j = 0; j < R2; ++j) x[j] = v[b * R2 + j];
    dft<R2>(x);
    if constexpr (TWIDDLE) {
      c32 pw[R2];
      twiddle_powers<R2>(fs_step[b], pw);
#pragma unroll
      for (int j = 0; j < R2; ++j) x[j] = cmul(x[j], j == 0 ? fs_base[b] : cmul(fs_base[b], pw[j]));
    }
#pragma unroll
    for (int j = 0; j < R2; ++j) v[b * R2 + j] = x[j];
  }
  if constexpr (!FWD) {
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = make_float2(v[e].y, v[e].x);
  }

  // ---- store
  {
    raw* dst = a.out + (out_base >> 1);
#pragma unroll
    for (int g = 0; g < NGL; ++g)
#pragma unroll
      for (int i = 0; i < SPGL; ++i) {
        raw r;
        if constexpr (NBL == 1) {
          r = IO::pack(v[i], v[i + R2 / 2]);
          IO::swap_halves(r);
        } else {
          r = IO::pack(v[(2 * g) * R2 + i], v[(2 * g + 1) * R2 + i]);
        }
        IO::store(dst + ((st_base[g] + (unsigned)i * out_row_step) >> 1), r);
      }
  }
}

}  // namespace tfft
