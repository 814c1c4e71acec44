// Fused circular convolution, N = 4096, fp16 storage, on the matrix cores:
//   y = IDFT( DFT(x) * H ),  both transforms as three v_mfma_f32_16x16x32_f16 radix-16 passes
// (fragment layouts, LDS images and their granule rotation: fft_mfma_kernel.hpp).
//
// Counterpart of the reference's thunder_conv4096_block_kernel family
// (thunderfft/sample/conv1d/conv1d_fullfft_l4096_thunder.cu:148-292), which likewise runs
// ThunderFFT_kernel_reg forward, multiplies by the permuted filter spectrum in registers and runs
// the inverse without leaving the register file (:178-189). Here the third forward pass leaves lane
// (g, i) of wave w with X[k = i + 16 (4w + t) + 256 (4g + r)], r = 0..3 for t = 0..3: four consecutive
// values along the inverse transform's first axis (stride 256) for column 16 (4w + t) + i, which is
// exactly the B fragment of the inverse's first pass. The product with H (stored in that lane
// order) is rounded to fp16 and fed straight back into the matrix cores.
//
// Pass-1 twiddles W_4096^(k1 c): 16 packed fp16 values per lane and column map, evaluated once per workgroup with
// v_sin_f32 / v_cos_f32 (conv_mfma4096_load_tw1), kept in registers and applied with the packed fp16 product the
// pass-2 twiddles use (3 instructions per value, rounding included). The fp32 form they replace -- 4 start and 4 step
// values per lane, W^(k1 (c0 + s)) = W^(k1 c0) (W^(k1 q))^s -- cost 4 + 1 instructions per value plus 12 complex
// products per pass for the chains: about 120 of the 490 VALU instructions per wave and signal, in a kernel whose
// on-chip work alone takes longer than its HBM traffic (with the loads or the stores compiled out it ran at 87 % and
// 82 % of its full time, with neither at 78 %). Rounding the twiddles to fp16 moves the result's error against an
// fp64 convolution from 5.5e-4 to 5.6e-4 (relative L2).
#pragma once
#include "fft_mfma_kernel.hpp"

namespace tfft {

struct ConvMfma4096 {
  // float2-sized entries appended after Mfma4096's tables: 2 maps x 4 fragments x 256 lanes x 4 packed fp16 pairs
  static constexpr int TABLE = 2 * 4 * 256 * 4 / 2;
  static constexpr int OFFSET = Mfma4096::N + Mfma4096::TABLES;
};

// Host side: the same values as a table, for the fp16 kernels of N = 8192 / 16384 (fft_mfma_big_kernel reads map 1
// once per transform). 16-byte entry [m][s][tid] = W^(k1 c) as fp16 (cos, -sin) for k1 = 4g + r, r = 0..3, with
// tid = 64w + 16g + i and column c = 64w + 4i + s (map m = 0: fragments straight from the signal's rows) or
// c = 16 (4w + s) + i (map m = 1: fragments that are the previous transform's output registers).
inline void conv_mfma4096_tables(float2* dst) {
  unsigned* out = reinterpret_cast<unsigned*>(dst);
  for (int m = 0; m < 2; ++m)
    for (int s = 0; s < 4; ++s)
      for (int tid = 0; tid < 256; ++tid) {
        const int wv = tid >> 6, g = (tid >> 4) & 3, i = tid & 15;
        const int c = m == 0 ? 64 * wv + 4 * i + s : 16 * (4 * wv + s) + i;
        for (int r = 0; r < 4; ++r) {
          const double a = -2.0 * M_PI * (double)(((4 * g + r) * c) & 4095) / 4096.0;
          out[((m * 4 + s) * 256 + tid) * 4 + r] =
              (unsigned)mfma_host_half((float)cos(a)) | ((unsigned)mfma_host_half((float)sin(a)) << 16);
        }
      }
}

// This lane's pass-1 twiddles of column map `map` (see conv_mfma4096_tables), evaluated rather than fetched:
// v_sin_f32 / v_cos_f32 take their argument in turns and (k1 c mod 4096) / 4096 is exact in fp32, so the 32
// transcendental instructions per map run under the wait for the first signal and the 16 KiB of table per map and
// workgroup stay out of L2 and the vector cache. (The table itself is still built: the fp16 kernels of
// N = 8192 / 16384 read map 1 of it.)
__device__ __forceinline__ void conv_mfma4096_load_tw1(int map, int tid, u32x4_t (&tw1)[4]) {
  const int w = tid >> 6, g = (tid >> 4) & 3, i = tid & 15;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const int c = map == 0 ? 64 * w + 4 * i + s : 16 * (4 * w + s) + i;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const float turns = (float)(((4 * g + r) * c) & 4095) * (1.0f / 4096);
      tw1[s][r] = __builtin_bit_cast(unsigned, pack_h2(__builtin_amdgcn_cosf(turns), -__builtin_amdgcn_sinf(turns)));
    }
  }
}

enum : int { kConvMfmaComplex = 0, kConvMfmaRealIn = 1, kConvMfmaRealPair = 2 };

// One direction's constants: the DFT-16 matrix fragments and the pass-2 twiddles of this lane.
template <bool FWD>
struct MfmaDir {
  mfma_h8 a_re, a_im;
  __device__ __forceinline__ void init(const float2* tw, int tid) {
    const u32x4_t* ftab = reinterpret_cast<const u32x4_t*>(tw + Mfma4096::N + Mfma4096::TW_LANE_ORDER) +
                          (FWD ? 0 : 256) + (tid & 63);
    a_re = __builtin_bit_cast(mfma_h8, ftab[0]);
    a_im = __builtin_bit_cast(mfma_h8, ftab[64]);
  }
  // (re + i im) * t forward, * conj(t) inverse; t = (cos, -sin)
  static __device__ __forceinline__ c32 twiddle(float re, float im, c32 t) {
    if (FWD) return make_float2(re * t.x - im * t.y, re * t.y + im * t.x);
    return make_float2(re * t.x + im * t.y, im * t.x - re * t.y);
  }
};

// Pass 1 on four fragments. CHAIN = false: fragment s holds column 64w + 4i + s (rows straight from
// global memory); CHAIN = true: column 16 (4w + s) + i (the previous transform's output registers).
template <bool FWD, bool CHAIN>
__device__ __forceinline__ void mfma_pass1(const MfmaDir<FWD>& d, const u32x4_t (&packed)[4], const u32x4_t (&tw1)[4],
                                           unsigned* img, int w, int g, int i) {
  const mfma_f4 zero = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const mfma_h8 b = __builtin_bit_cast(mfma_h8, packed[s]);
    const mfma_f4 cr = __builtin_amdgcn_mfma_f32_16x16x32_f16(d.a_re, b, zero, 0, 0, 0);
    const mfma_f4 ci = __builtin_amdgcn_mfma_f32_16x16x32_f16(d.a_im, b, zero, 0, 0, 0);
    // image row = (k1, n3), point n2 (granules rotated by ((row >> 1) + (row >> 6)) & 3)
    const int n2 = CHAIN ? 4 * w + s : 4 * w + (i >> 2);
    const int n3 = CHAIN ? i : 4 * (i & 3) + s;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = (4 * g + r) * 16 + n3;
      const int rot = ((n3 >> 1) + g) & 3;
      img[row * 16 + ((n2 + 4 * rot) & 15)] =
          cmul_h2<!FWD>(pack_h2(cr[r], ci[r]), __builtin_bit_cast(h2_t, (unsigned)tw1[s][r]));
    }
  }
}

// Pass 2: image `src` (rows (k1, n3), points n2) to image `dst` (rows (k2, k1), points n3).
template <bool FWD>
__device__ __forceinline__ void mfma_pass2(const MfmaDir<FWD>& d, const h2_t (&t2)[4], const unsigned* src, unsigned* dst,
                                           int w, int g, int i) {
  const mfma_f4 zero = {0.f, 0.f, 0.f, 0.f};
  u32x4_t frag[4];
  const int gran = (g + (i >> 1) + w) & 3;
#pragma unroll
  for (int t = 0; t < 4; ++t)
    frag[t] = *reinterpret_cast<const u32x4_t*>(&src[((4 * w + t) * 16 + i) * 16 + 4 * gran]);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const mfma_h8 b = __builtin_bit_cast(mfma_h8, frag[t]);
    const mfma_f4 cr = __builtin_amdgcn_mfma_f32_16x16x32_f16(d.a_re, b, zero, 0, 0, 0);
    const mfma_f4 ci = __builtin_amdgcn_mfma_f32_16x16x32_f16(d.a_im, b, zero, 0, 0, 0);
    const int rot = (t + 2 * w + (t >> 1)) & 3;
    const int pos = (i + 4 * rot) & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r)
      dst[((4 * g + r) * 16 + 4 * w + t) * 16 + pos] = cmul_h2<!FWD>(pack_h2(cr[r], ci[r]), t2[r]);
  }
}

// Pass 3: image `src` (rows (k2, k1), points n3) to registers: out[t][r] = X[i + 16 (4w + t) + 256 (4g + r)].
template <bool FWD>
__device__ __forceinline__ void mfma_pass3(const MfmaDir<FWD>& d, const unsigned* src, mfma_f4 (&xr)[4], mfma_f4 (&xi)[4],
                                           int w, int g, int i) {
  const mfma_f4 zero = {0.f, 0.f, 0.f, 0.f};
  u32x4_t frag[4];
  const int gran = (g + i + (i >> 1)) & 3;
#pragma unroll
  for (int t = 0; t < 4; ++t)
    frag[t] = *reinterpret_cast<const u32x4_t*>(&src[((4 * w + t) * 16 + i) * 16 + 4 * gran]);
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const mfma_h8 b = __builtin_bit_cast(mfma_h8, frag[t]);
    xr[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(d.a_re, b, zero, 0, 0, 0);
    xi[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(d.a_im, b, zero, 0, 0, 0);
  }
}

// Filter spectra in the order the kernel's lanes consume them: 16-byte entry [d][t][tid] holds
// H[d][k = i + 16 (4w + t) + 256 (4g + r)], r = 0..3 (tid = 64w + 16g + i). Counterpart of
// permute_filter_layout4096 (conv1d_fullfft_l4096_thunder.cu:120-146).
template <int UNUSED = 0>  // a template so that the header can be included from several translation units
__global__ __launch_bounds__(256) void conv_mfma4096_filter_layout_kernel(const unsigned* __restrict__ h,
                                                                         u32x4_t* __restrict__ hp, int D) {
  const int tid = threadIdx.x, w = tid >> 6, g = (tid >> 4) & 3, i = tid & 15;
  const long long d = blockIdx.x;
  const unsigned* src = h + d * 4096;
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    const int k = i + 16 * (4 * w + t) + 1024 * g;
    hp[d * 1024 + t * 256 + tid] = u32x4_t{src[k], src[k + 256], src[k + 512], src[k + 768]};
  }
}

// Signals [B][D][4096]; workgroups that share a filter are adjacent (block = d * nb + b').
//   KIND complex  : x, y interleaved fp16 complex
//   KIND real in  : x real fp16, y complex
//   KIND real pair: x, y real fp16, real taps; signals (2b', 2b'+1) of filter d share one transform
// PERSIST: a resident grid; a workgroup walks work items blockIdx.x, blockIdx.x + gridDim.x, ... and
// requests the rows of its next item as soon as the current rows are packed into fragments.
template <int KIND, bool PERSIST>
__global__ __launch_bounds__(256, PERSIST ? 2 : 3) void fft_conv_mfma4096_kernel(const void* xv, const u32x4_t* __restrict__ hp,
                                                                  void* yv,  // xv and yv may alias
                                                                  const float2* __restrict__ tw, int B,
                                                                  int D) {
  __shared__ __attribute__((aligned(16))) unsigned lds[Mfma4096::OUT_IMG + Mfma4096::IMG];
  unsigned* const img_a = lds;
  unsigned* const img_b = lds + Mfma4096::OUT_IMG;
  const int tid = threadIdx.x;
  const int w = tid >> 6, g = (tid >> 4) & 3, i = tid & 15;

  // work item -> (filter d, signal b0 [and b0 + 1]); items that share a filter are adjacent
  const int nb = (KIND == kConvMfmaRealPair) ? (B + 1) >> 1 : B;
  const long long items = (long long)nb * D;
  struct Item {
    int d;
    bool two;
    long long s0, s1;
  };
  auto locate = [&](long long idx) {
    Item it;
    it.d = (int)(idx / nb);
    const int bq = (int)(idx - (long long)it.d * nb);
    const int b0 = (KIND == kConvMfmaRealPair) ? 2 * bq : bq;
    it.two = (KIND == kConvMfmaRealPair) && (b0 + 1 < B);
    it.s0 = (long long)b0 * D + it.d;
    it.s1 = it.s0 + D;
    return it;
  };

  // ---- rows of a signal: lane (g, i) of wave w takes rows 4G + g, columns 64w + 4i .. 4i+3
  u32x4_t rows[4];
  auto request_rows = [&](const Item& it) {
    if constexpr (KIND == kConvMfmaComplex) {
      const u32x4_t* src = static_cast<const u32x4_t*>(xv) + it.s0 * 1024 + 16 * w + i;
#pragma unroll
      for (int G = 0; G < 4; ++G) rows[G] = __builtin_nontemporal_load(src + (4 * G + g) * 64);
    } else {
      const u32x2_t* src0 = static_cast<const u32x2_t*>(xv) + it.s0 * 1024 + 16 * w + i;
      const u32x2_t* src1 = static_cast<const u32x2_t*>(xv) + it.s1 * 1024 + 16 * w + i;
#pragma unroll
      for (int G = 0; G < 4; ++G) {
        const u32x2_t a = __builtin_nontemporal_load(src0 + (4 * G + g) * 64);
        u32x2_t c = {0u, 0u};
        if (it.two) c = __builtin_nontemporal_load(src1 + (4 * G + g) * 64);
        // (re, im) = (signal b0, signal b0 + 1 or zero), one dword per point
        rows[G] = u32x4_t{(a.x & 0xffffu) | (c.x << 16), (a.x >> 16) | (c.x & 0xffff0000u),
                          (a.y & 0xffffu) | (c.y << 16), (a.y >> 16) | (c.y & 0xffff0000u)};
      }
    }
  };

  long long idx = blockIdx.x;
  if (idx >= items) return;
  Item cur = locate(idx);
  request_rows(cur);

  // ---- the pass-1 twiddles of both column maps, the matrix fragments
  u32x4_t tw1_g[4], tw1_c[4];
  conv_mfma4096_load_tw1(0, tid, tw1_g);
  conv_mfma4096_load_tw1(1, tid, tw1_c);
  h2_t t2[4];
  {
    const u32x4_t* ftab = reinterpret_cast<const u32x4_t*>(tw + Mfma4096::N + Mfma4096::TW_LANE_ORDER) + (tid & 63);
    const u32x4_t p = ftab[128], q = ftab[192];
    t2[0] = pack_h2(__uint_as_float(p.x), __uint_as_float(p.y));
    t2[1] = pack_h2(__uint_as_float(p.z), __uint_as_float(p.w));
    t2[2] = pack_h2(__uint_as_float(q.x), __uint_as_float(q.y));
    t2[3] = pack_h2(__uint_as_float(q.z), __uint_as_float(q.w));
  }
  MfmaDir<true> fwd;
  MfmaDir<false> inv;
  fwd.init(tw, tid);
  inv.init(tw, tid);

  for (;;) {
    // ---- this item's filter in lane order: it has the forward transform's time to arrive
    u32x4_t hreg[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) hreg[t] = hp[(long long)cur.d * 1024 + t * 256 + tid];

    // ---- forward transform
    u32x4_t packed[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned a0 = rows[0][j], a1 = rows[1][j], a2 = rows[2][j], a3 = rows[3][j];
      auto s = __builtin_amdgcn_permlane32_swap(a0, a2, false, false);
      a0 = s[0]; a2 = s[1];
      s = __builtin_amdgcn_permlane32_swap(a1, a3, false, false);
      a1 = s[0]; a3 = s[1];
      s = __builtin_amdgcn_permlane16_swap(a0, a1, false, false);
      a0 = s[0]; a1 = s[1];
      s = __builtin_amdgcn_permlane16_swap(a2, a3, false, false);
      a2 = s[0]; a3 = s[1];
      packed[j] = u32x4_t{a0, a1, a2, a3};
    }
    const long long next = idx + gridDim.x;
    const bool more = PERSIST && next < items;
    Item nxt = cur;
    if (more) {
      nxt = locate(next);
      request_rows(nxt);
    }
    mfma_pass1<true, false>(fwd, packed, tw1_g, img_a, w, g, i);
    __syncthreads();
    mfma_pass2<true>(fwd, t2, img_a, img_b, w, g, i);
    __syncthreads();
    mfma_f4 xr[4], xi[4];
    mfma_pass3<true>(fwd, img_b, xr, xi, w, g, i);

    // ---- product with the filter spectrum, rounded to fp16: the inverse's first-pass fragments
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        packed[t][r] = cmul_h2<false>(pack_h2(xr[t][r], xi[t][r]), __builtin_bit_cast(h2_t, (unsigned)hreg[t][r]));
    }

    // ---- inverse transform (image a is free: every wave is past the barrier that followed its reads)
    mfma_pass1<false, true>(inv, packed, tw1_c, img_a, w, g, i);
    __syncthreads();
    mfma_pass2<false>(inv, t2, img_a, img_b, w, g, i);
    __syncthreads();
    mfma_pass3<false>(inv, img_b, xr, xi, w, g, i);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k3 = 4 * g + r;
        img_a[i + 16 * (4 * w + t) + 256 * k3 + 4 * k3] = pack_f16(xr[t][r], xi[t][r]);
      }
    }
    __syncthreads();

    // ---- natural order out
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const u32x4_t v = *reinterpret_cast<const u32x4_t*>(&img_a[4 * (tid + 256 * q) + 4 * (w + 4 * q)]);
      if constexpr (KIND == kConvMfmaRealPair) {
        const u32x2_t y0 = {(v.x & 0xffffu) | (v.y << 16), (v.z & 0xffffu) | (v.w << 16)};
        const u32x2_t y1 = {(v.x >> 16) | (v.y & 0xffff0000u), (v.z >> 16) | (v.w & 0xffff0000u)};
        __builtin_nontemporal_store(y0, static_cast<u32x2_t*>(yv) + cur.s0 * 1024 + tid + 256 * q);
        if (cur.two) __builtin_nontemporal_store(y1, static_cast<u32x2_t*>(yv) + cur.s1 * 1024 + tid + 256 * q);
      } else {
        __builtin_nontemporal_store(v, static_cast<u32x4_t*>(yv) + cur.s0 * 1024 + tid + 256 * q);
      }
    }
    if (!more) break;
    __syncthreads();  // the output image is read before the next item's first pass overwrites it
    idx = next;
    cur = nxt;
  }
}

// 16 bytes per lane from global memory straight into LDS (no VGPR in between): lane l of the wave lands at
// lds_byte_address + 16 l. The instruction takes its LDS base from M0, which the compiler owns: saved and restored
// inside the statement. The load is invisible to the compiler's s_waitcnt bookkeeping; the kernel below waits for it
// itself (s_waitcnt vmcnt(0) at a point where nothing else of its own is outstanding).
__device__ __forceinline__ void lds_dma_16(const void* gsrc, unsigned lds_byte_address) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_byte_address)
      : "memory");
}

// The convolution with the NEXT work item's signal travelling into LDS while the current one is in its six
// passes. The one-shot kernel measures close to the SUM of its HBM time and its on-chip time: a workgroup waits
// for its rows, computes, stores, and only the other three workgroups of the CU fill those waits. Here a
// workgroup takes RUN consecutive items of one filter; right after it has read item k's signal out of a 16 KiB
// staging area it starts LDS-DMA loads of item k + 1's signal into that area (no registers held for it, no wait
// until just before item k's stores), so that every workgroup keeps a load in flight under its own arithmetic.
// LDS: two pass images + the staging area = 49 KiB, three workgroups per CU. The filter spectrum and all twiddle
// constants stay in registers for the run.
//
// PRIV: every wave stages exactly the bytes its own lanes consume (its 64 columns of all 16 rows: 16 segments of
// 256 bytes of a complex signal, 128 bytes of a real one), lane l fetching the 16 bytes lane l will read, so the
// staging area needs no workgroup barrier on either side: a wave waits for its own loads, reads, and issues the next
// item's. The two pass images trade places from item to item, so that an item's first pass writes the image whose
// last readers (the previous item's third inverse pass) are already behind a barrier, and the barrier that kept the
// output image's readers ahead of the next first pass goes as well: five barriers per item instead of seven, and a
// wave may start the next item while the others still store this one.
template <int KIND, int RUN, bool PRIV = false>
__global__ __launch_bounds__(256, 3) void fft_conv_mfma4096_dma_kernel(const void* xv, const u32x4_t* __restrict__ hp,
                                                                       void* yv,  // xv and yv may alias
                                                                       const float2* __restrict__ tw, int B, int D) {
  constexpr int kImgB = PRIV ? Mfma4096::OUT_IMG : Mfma4096::IMG;  // PRIV: either image may hold the output
  __shared__ __attribute__((aligned(16))) unsigned lds[Mfma4096::OUT_IMG + kImgB];
  __shared__ __attribute__((aligned(16))) unsigned stage[4096];  // one item's signal(s): as they lie in memory, or
                                                                 // (PRIV) 4 KiB per wave in its lanes' order
  unsigned* const img_a = lds;
  unsigned* const img_b = lds + Mfma4096::OUT_IMG;
  const int tid = threadIdx.x;
  const int w = tid >> 6, g = (tid >> 4) & 3, i = tid & 15;

  const int nb = (KIND == kConvMfmaRealPair) ? (B + 1) >> 1 : B;
  const int runs = (nb + RUN - 1) / RUN;
  const int d = blockIdx.x / runs;
  if (d >= D) return;
  const int bq0 = (blockIdx.x - d * runs) * RUN;
  const int bq1 = min(nb, bq0 + RUN);

  // first signal index of item bq, and whether a second signal rides along (real pairs)
  auto first_signal = [&](int bq) { return (long long)((KIND == kConvMfmaRealPair) ? 2 * bq : bq) * D + d; };
  auto has_second = [&](int bq) { return (KIND == kConvMfmaRealPair) && (2 * bq + 1 < B); };

  // wave w copies 1 KiB pieces: piece (round, w) of the staging area = bytes [4096 round + 1024 w, + 1024);
  // PRIV: piece r of the wave's own 4 KiB = bytes [4096 w + 1024 r, + 1024)
  const unsigned stage_lds = (unsigned)(uintptr_t)stage;
  const unsigned wave_piece = __builtin_amdgcn_readfirstlane(stage_lds + (PRIV ? 4096u : 1024u) * (unsigned)w);
  const int lane = tid & 63;
  auto start_dma = [&](int bq) {
    const long long s0 = first_signal(bq);
    if constexpr (PRIV && KIND == kConvMfmaComplex) {
      // instruction r: rows 4r .. 4r + 3, 16 lanes (256 bytes) per row: lane (g, i) fetches what it reads as rows[r]
      const u32x4_t* src = static_cast<const u32x4_t*>(xv) + s0 * 1024 + 16 * w + (lane >> 4) * 64 + (lane & 15);
#pragma unroll
      for (int r = 0; r < 4; ++r) lds_dma_16(src + 256 * r, wave_piece + 1024u * r);
    } else if constexpr (PRIV) {
      // instruction r: rows 8r .. 8r + 7 of one real signal, 8 lanes (128 bytes) per row
      const u32x4_t* src0 = static_cast<const u32x4_t*>(xv) + s0 * 512 + 8 * w + (lane >> 3) * 32 + (lane & 7);
#pragma unroll
      for (int r = 0; r < 2; ++r) lds_dma_16(src0 + 256 * r, wave_piece + 1024u * r);
      if (has_second(bq)) {  // uniform over the workgroup
        const u32x4_t* src1 = src0 + (long long)D * 512;
#pragma unroll
        for (int r = 0; r < 2; ++r) lds_dma_16(src1 + 256 * r, wave_piece + 2048u + 1024u * r);
      }
    } else if constexpr (KIND == kConvMfmaComplex) {
      const u32x4_t* src = static_cast<const u32x4_t*>(xv) + s0 * 1024 + tid;  // 16 KiB, linear
#pragma unroll
      for (int r = 0; r < 4; ++r) lds_dma_16(src + 256 * r, wave_piece + 4096u * r);
    } else {
      const u32x4_t* src0 = static_cast<const u32x4_t*>(xv) + s0 * 512 + tid;  // 8 KiB per real signal
#pragma unroll
      for (int r = 0; r < 2; ++r) lds_dma_16(src0 + 256 * r, wave_piece + 4096u * r);
      if (has_second(bq)) {  // uniform over the workgroup
        const u32x4_t* src1 = src0 + (long long)D * 512;
#pragma unroll
        for (int r = 0; r < 2; ++r) lds_dma_16(src1 + 256 * r, wave_piece + 8192u + 4096u * r);
      }
    }
  };
  start_dma(bq0);

  // ---- constants of the run, all in registers
  u32x4_t hreg[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) hreg[t] = hp[(long long)d * 1024 + t * 256 + tid];
  u32x4_t tw1_g[4], tw1_c[4];
  conv_mfma4096_load_tw1(0, tid, tw1_g);
  conv_mfma4096_load_tw1(1, tid, tw1_c);
  h2_t t2[4];
  {
    const u32x4_t* ftab = reinterpret_cast<const u32x4_t*>(tw + Mfma4096::N + Mfma4096::TW_LANE_ORDER) + (tid & 63);
    const u32x4_t p = ftab[128], q = ftab[192];
    t2[0] = pack_h2(__uint_as_float(p.x), __uint_as_float(p.y));
    t2[1] = pack_h2(__uint_as_float(p.z), __uint_as_float(p.w));
    t2[2] = pack_h2(__uint_as_float(q.x), __uint_as_float(q.y));
    t2[3] = pack_h2(__uint_as_float(q.z), __uint_as_float(q.w));
  }
  MfmaDir<true> fwd;
  MfmaDir<false> inv;
  fwd.init(tw, tid);
  inv.init(tw, tid);

  // the first item's signal: every wave waits for its own pieces (and for the table loads above), then all meet
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  auto item = [&](int bq, unsigned* const img_a, unsigned* const img_b) __attribute__((always_inline)) {
    const long long s0 = first_signal(bq), s1 = s0 + D;
    const bool two = has_second(bq);
    // an opaque copy per item: without it the optimiser derives per-use forms of the 32 twiddle registers (halves
    // splatted, signs flipped) outside the loop and keeps those alive for the whole run
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(tw1_g[s][r]), "+v"(tw1_c[s][r]));
    // the staging area is complete (each wave waited for its pieces before arriving here); PRIV: a wave reads
    // nothing but its own pieces
    if constexpr (!PRIV) __syncthreads();

    // ---- this lane's rows 4G + g, columns 64w + 4i .. 4i + 3, out of the staging area
    u32x4_t rows[4];
#pragma unroll
    for (int G = 0; G < 4; ++G) {
      if constexpr (KIND == kConvMfmaComplex) {
        rows[G] = *reinterpret_cast<const u32x4_t*>(
            &stage[PRIV ? 1024 * w + 256 * G + 4 * lane : 4 * ((4 * G + g) * 64 + 16 * w + i)]);
      } else {
        // dword index of four real points; the second signal of a pair lies kSecond dwords further
        constexpr int kSecond = PRIV ? 512 : 2048;
        const int at = PRIV ? 1024 * w + 128 * G + 2 * lane : 2 * ((4 * G + g) * 64 + 16 * w + i);
        const u32x2_t a = *reinterpret_cast<const u32x2_t*>(&stage[at]);
        u32x2_t c = {0u, 0u};
        if (two) c = *reinterpret_cast<const u32x2_t*>(&stage[kSecond + at]);
        rows[G] = u32x4_t{(a.x & 0xffffu) | (c.x << 16), (a.x >> 16) | (c.x & 0xffff0000u),
                          (a.y & 0xffffu) | (c.y << 16), (a.y >> 16) | (c.y & 0xffff0000u)};
      }
    }
    u32x4_t packed[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned a0 = rows[0][j], a1 = rows[1][j], a2 = rows[2][j], a3 = rows[3][j];
      auto s = __builtin_amdgcn_permlane32_swap(a0, a2, false, false);
      a0 = s[0]; a2 = s[1];
      s = __builtin_amdgcn_permlane32_swap(a1, a3, false, false);
      a1 = s[0]; a3 = s[1];
      s = __builtin_amdgcn_permlane16_swap(a0, a1, false, false);
      a0 = s[0]; a1 = s[1];
      s = __builtin_amdgcn_permlane16_swap(a2, a3, false, false);
      a2 = s[0]; a3 = s[1];
      packed[j] = u32x4_t{a0, a1, a2, a3};
    }
    // every wave has its rows in registers: the staging area is free (PRIV: this wave's reads of its own pieces
    // have returned, which is all that its next loads can overwrite)
    if constexpr (PRIV) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    else __syncthreads();
    if (bq + 1 < bq1) start_dma(bq + 1);

    // ---- forward transform, product, inverse transform
    mfma_pass1<true, false>(fwd, packed, tw1_g, img_a, w, g, i);
    __syncthreads();
    mfma_pass2<true>(fwd, t2, img_a, img_b, w, g, i);
    __syncthreads();
    mfma_f4 xr[4], xi[4];
    mfma_pass3<true>(fwd, img_b, xr, xi, w, g, i);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
        packed[t][r] = cmul_h2<false>(pack_h2(xr[t][r], xi[t][r]), __builtin_bit_cast(h2_t, (unsigned)hreg[t][r]));
    }
    mfma_pass1<false, true>(inv, packed, tw1_c, img_a, w, g, i);
    __syncthreads();
    mfma_pass2<false>(inv, t2, img_a, img_b, w, g, i);
    __syncthreads();
    mfma_pass3<false>(inv, img_b, xr, xi, w, g, i);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int k3 = 4 * g + r;
        img_a[i + 16 * (4 * w + t) + 256 * k3 + 4 * k3] = pack_f16(xr[t][r], xi[t][r]);
      }
    }
    __syncthreads();

    // ---- natural order out. The next signal has had the six passes to arrive: waiting for it here, before this
    // item's stores are issued, leaves nothing else outstanding to wait for.
    u32x4_t v[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) v[q] = *reinterpret_cast<const u32x4_t*>(&img_a[4 * (tid + 256 * q) + 4 * (w + 4 * q)]);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if constexpr (KIND == kConvMfmaRealPair) {
        const u32x2_t y0 = {(v[q].x & 0xffffu) | (v[q].y << 16), (v[q].z & 0xffffu) | (v[q].w << 16)};
        const u32x2_t y1 = {(v[q].x >> 16) | (v[q].y & 0xffff0000u), (v[q].z >> 16) | (v[q].w & 0xffff0000u)};
        __builtin_nontemporal_store(y0, static_cast<u32x2_t*>(yv) + s0 * 1024 + tid + 256 * q);
        if (two) __builtin_nontemporal_store(y1, static_cast<u32x2_t*>(yv) + s1 * 1024 + tid + 256 * q);
      } else {
        __builtin_nontemporal_store(v[q], static_cast<u32x4_t*>(yv) + s0 * 1024 + tid + 256 * q);
      }
    }
  };
  if constexpr (!PRIV) {
#pragma unroll 1
    for (int bq = bq0; bq < bq1; ++bq) item(bq, img_a, img_b);
  } else {
#pragma unroll 1
    for (int bq = bq0; bq < bq1; bq += 2) {
      item(bq, img_a, img_b);
      if (bq + 1 >= bq1) break;
      item(bq + 1, img_b, img_a);
    }
  }
}

}  // namespace tfft
