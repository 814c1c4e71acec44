// 16-bit storage, N = 256 ... 16384: the passes of the transform on the matrix cores
// (fft_mfma4096_kernel: N = 256 ... 4096; fft_mfma_big_kernel at the end of the file: N = 8192, 16384).
//
// The reference computes its transforms as tensor-core products of small DFT matrices with
// fragments of the data (thunderfft/include/thunderfft/detail/kernel/unit_kernel_fp16.cuh:97-238,
// mma.m16n8k8 with fp16 accumulation). This is the wave64 counterpart for gfx950: one
// v_mfma_f32_16x16x32_f16 multiplies the real-ified 16-point DFT matrix (16 x 32: rows = output
// index, K = 16 points x (re, im)) by 16 columns of fp16 data, fp32 accumulation; two of them (real
// and imaginary output rows) transform 16 columns. 4096 = 16 x 16 x 16, n = 256 n1 + 16 n2 + n3,
// k = k1 + 16 k2 + 256 k3, decimation in frequency like the VALU kernels:
//
//   pass 1 over n1, twiddle W_4096^(k1 (16 n2 + n3)); pass 2 over n2, twiddle W_256^(k2 n3); pass 3 over n3.
//
// Fragment layouts (lane l = 16 g + i): the B operand of lane (g, i) is points 4g .. 4g+3 of column
// i -- four (re, im) fp16 pairs, 16 contiguous bytes when the contracted axis is fastest; the
// result D gives the same lane outputs 4g .. 4g+3 of column i. So every pass needs its input with
// the contracted axis fastest:
//   * pass 1 takes it from registers: a lane loads 16 bytes (four adjacent columns of one row) for
//     four rows, and a 4 x 4 transpose across lanes l, l+16, l+32, l+48 (two v_permlane32_swap and
//     two v_permlane16_swap per column) leaves it with four consecutive rows of each column;
//   * passes 2 and 3 read 16-byte granules (ds_read_b128) from an LDS image [column][point] that the
//     previous pass wrote dword by dword; the image has no padding, its four granules per row are
//     rotated by a function of the row chosen by tools/lds_mfma_layout.py so that every read is
//     conflict-free and every write at most 2-way (free for ds_write_b32);
//   * the output goes through a natural-order image so that global stores are 16 bytes per lane.
// Data is rounded to fp16 between passes (the storage type's own precision); accumulation and the
// pass-1 twiddles are fp32, the pass-2 twiddles of fp16 storage are applied on packed fp16. Intermediate values must fit fp16, i.e. N * max|x| < 65504 -- the bound
// that also keeps the fp16 result finite.
//
// N = 256 R1 with R1 = 1, 2, 4, 8: a workgroup takes 16 / R1 whole transforms (the same 4096 points,
// 16 rows of 256) and the first pass multiplies by the block-diagonal matrix I_(16/R1) (x) DFT_R1 -- a
// radix-R1 pass over the rows of each transform, at the matrix-core cost of the radix-16 one -- followed
// by the twiddle W_N^(k1 (16 n2 + n3)); passes 2 and 3 are the same 256-point transforms. Row s of the
// pass-1 result is sub-transform (transform s / R1, k1 = s mod R1) and output k = k1 + R1 (k2 + 16 k3).
// R1 = 1 needs no first pass: the transposed rows are the pass-2 image.
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <type_traits>

#include "fft_lds_kernel.hpp"

namespace tfft {

typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

struct Mfma4096 {
  static constexpr int N = 4096, WG = 256;
  static constexpr int IMG = 4096;            // dwords of a pass image
  static constexpr int OUT_IMG = 4096 + 64;   // natural-order output image, 4 dwords of pad per 256
  static constexpr int TW_LANE_ORDER = 4096;  // float2 entries appended to the twiddle table: pass-1 twiddles
  static constexpr int FRAG_TABLE = 1024;     // then, per direction: matrix fragments and pass-2 twiddles per lane
  static constexpr int BF16_FRAG_TABLE = 1024;  // then, per direction: the matrix as bf16 high and low parts per lane
  static constexpr int P1_TABLE = 512;  // then, per storage type and direction: the pass-1 matrix of N < 4096 per lane
  static constexpr int INCR_TABLE = 2 * (1024 + 16);  // then start and step values of the incrementally built pass-1 twiddles, two column maps
  static constexpr int TABLES = TW_LANE_ORDER + FRAG_TABLE + BF16_FRAG_TABLE + 2 * P1_TABLE + INCR_TABLE;
};

// Host side: the pass-1 twiddles in the order the kernel's lanes consume them. Entry
// [(j*2 + rp) * 256 + tid] holds the float4 { W^(k1 c), W^(k1' c) } for rows s = 4 g + 2 rp and s + 1,
// k1 = s mod R1, with tid = 64 w + 16 g + i, c = 64 w + 4 i + j, W = exp(-2 pi i / n), R1 = n / 256.
inline unsigned short mfma_host_half(float f) {  // round to nearest even; |f| <= 1, no subnormal results here
  unsigned u;
  memcpy(&u, &f, 4);
  const unsigned sign = (u >> 16) & 0x8000u;
  const unsigned mag = u & 0x7fffffffu;
  if (mag < 0x38800000u) return (unsigned short)sign;  // below the smallest normal half: the tables only hold 0 there
  const unsigned rounded = mag + 0xfffu + ((mag >> 13) & 1u);
  return (unsigned short)(sign | ((rounded - 0x38000000u) >> 13));
}

// n < 256 (64, 128): the tile's rows hold 256 / n whole transforms; pass 2 multiplies by I (x) DFT_q, q = n / 16, and
// its twiddles are W_n^((k2 mod q) n3). The tables keep their places: the block matrix sits where the pass-1 matrix
// of the longer lengths does, the twiddles where the W_256 ones do.
inline void mfma_lane_tables(float2* dst, int n) {
  const int r1 = n < 256 ? 1 : n / 256;
  const int blk = n < 256 ? n / 16 : r1;  // block size of the block-diagonal matrix
  for (int j = 0; j < 4; ++j)
    for (int rp = 0; rp < 2; ++rp)
      for (int tid = 0; tid < 256; ++tid) {
        const int w = tid >> 6, g = (tid >> 4) & 3, i = tid & 15;
        const int c = 64 * w + 4 * i + j;
        for (int h = 0; h < 2; ++h) {
          const int k1 = (4 * g + 2 * rp + h) % r1;
          const double a = -2.0 * M_PI * (double)((k1 * c) % n) / (double)n;
          dst[(((j * 2 + rp) * 256) + tid) * 2 + h] = make_float2((float)cos(a), (float)sin(a));
        }
      }
  // Per direction d (0 forward, 1 inverse), 16-byte entry [d*256 + q*64 + l] of lane l = 16 g + i:
  //   q = 0: row i of the real-output matrix, K = 8g .. 8g+7 as fp16: (cos, +-sin) per point m = 4g + jj
  //   q = 1: row i of the imaginary-output matrix: (-+sin, cos)
  //   q = 2, 3: the pass-2 twiddles W_256^((4g + r) i), r = 0..3, as fp32 (cos, -sin)
  unsigned char* base = reinterpret_cast<unsigned char*>(dst + Mfma4096::TW_LANE_ORDER);
  for (int d = 0; d < 2; ++d)
    for (int l = 0; l < 64; ++l) {
      const int g = l >> 4, i = l & 15;
      unsigned short are[8], aim[8];
      for (int jj = 0; jj < 4; ++jj) {
        const double a = 2.0 * M_PI * (double)((i * (4 * g + jj)) & 15) / 16.0;
        const float c = (float)cos(a), sn = (float)(d == 0 ? sin(a) : -sin(a));
        are[2 * jj] = mfma_host_half(c);
        are[2 * jj + 1] = mfma_host_half(sn);
        aim[2 * jj] = mfma_host_half(-sn);
        aim[2 * jj + 1] = mfma_host_half(c);
      }
      float t2[8];
      for (int r = 0; r < 4; ++r) {
        const double a = n < 256 ? -2.0 * M_PI * (double)(((4 * g + r) % blk) * i) / (double)n
                                 : -2.0 * M_PI * (double)((4 * g + r) * i) / 256.0;
        t2[2 * r] = (float)cos(a);
        t2[2 * r + 1] = (float)sin(a);
      }
      memcpy(base + 16 * (d * 256 + 0 * 64 + l), are, 16);
      memcpy(base + 16 * (d * 256 + 1 * 64 + l), aim, 16);
      memcpy(base + 16 * (d * 256 + 2 * 64 + l), t2, 16);
      memcpy(base + 16 * (d * 256 + 3 * 64 + l), t2 + 4, 16);
    }
  // bf16 storage: 16-byte entry [d*256 + q*64 + l]: q = 0, 1 the two matrices rounded to bf16, q = 2, 3 the
  // rounding remainders (value - high part) as bf16, so that high + low carries 16 mantissa bits
  auto bf16_rne = [](float f) {
    unsigned u;
    memcpy(&u, &f, 4);
    return (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
  };
  auto bf16_val = [](unsigned short b) {
    const unsigned u = (unsigned)b << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
  };
  unsigned char* bbase = base + 8 * Mfma4096::FRAG_TABLE;
  for (int d = 0; d < 2; ++d)
    for (int l = 0; l < 64; ++l) {
      const int g = l >> 4, i = l & 15;
      unsigned short hi[2][8], lo[2][8];
      for (int jj = 0; jj < 4; ++jj) {
        const double a = 2.0 * M_PI * (double)((i * (4 * g + jj)) & 15) / 16.0;
        const float c = (float)cos(a), sn = (float)(d == 0 ? sin(a) : -sin(a));
        const float v[2][2] = {{c, sn}, {-sn, c}};
        for (int m = 0; m < 2; ++m)
          for (int part = 0; part < 2; ++part) {
            hi[m][2 * jj + part] = bf16_rne(v[m][part]);
            lo[m][2 * jj + part] = bf16_rne(v[m][part] - bf16_val(hi[m][2 * jj + part]));
          }
      }
      memcpy(bbase + 16 * (d * 256 + 0 * 64 + l), hi[0], 16);
      memcpy(bbase + 16 * (d * 256 + 1 * 64 + l), hi[1], 16);
      memcpy(bbase + 16 * (d * 256 + 2 * 64 + l), lo[0], 16);
      memcpy(bbase + 16 * (d * 256 + 3 * 64 + l), lo[1], 16);
    }
  // Pass-1 matrix I (x) DFT_r1 per lane, 16-byte entry [st*256 + d*128 + q*64 + l], st = 0 fp16, 1 bf16;
  // q = 0 real-output rows, q = 1 imaginary-output rows; row s = i, K = points rho = 4g .. 4g+3:
  // [s / r1 == rho / r1] exp(-+2 pi i (s mod r1)(rho mod r1) / r1).
  unsigned char* pbase = bbase + 8 * Mfma4096::BF16_FRAG_TABLE;
  for (int st = 0; st < 2; ++st)
    for (int d = 0; d < 2; ++d)
      for (int l = 0; l < 64; ++l) {
        const int g = l >> 4, i = l & 15;
        unsigned short m[2][8];
        for (int jj = 0; jj < 4; ++jj) {
          const int rho = 4 * g + jj;
          float c = 0.f, sn = 0.f;
          if (i / blk == rho / blk) {
            const double a = 2.0 * M_PI * (double)(((i % blk) * (rho % blk)) % blk) / (double)blk;
            c = (float)cos(a);
            sn = (float)(d == 0 ? sin(a) : -sin(a));
          }
          const float v[2][2] = {{c, sn}, {-sn, c}};
          for (int q = 0; q < 2; ++q)
            for (int part = 0; part < 2; ++part)
              m[q][2 * jj + part] = st == 0 ? mfma_host_half(v[q][part]) : bf16_rne(v[q][part]);
        }
        memcpy(pbase + 16 * (st * 256 + d * 128 + 0 * 64 + l), m[0], 16);
        memcpy(pbase + 16 * (st * 256 + d * 128 + 1 * 64 + l), m[1], 16);
      }
  // Incremental pass-1 twiddles: float4 start [h][tid] = { W^(k1 c0), W^(k1' c0) } for rows s = 4g + 2h and s + 1,
  // c0 = 64w + 4i, then float4 step [h][g] = { W^k1, W^k1' }; k1 = s mod r1.
  float2* inc = reinterpret_cast<float2*>(pbase + 16 * 512);
  auto wn = [n](long long e) {
    const double a = -2.0 * M_PI * (double)(e % n) / (double)n;
    return make_float2((float)cos(a), (float)sin(a));
  };
  for (int h = 0; h < 2; ++h) {
    for (int tid = 0; tid < 256; ++tid) {
      const int wv = tid >> 6, g = (tid >> 4) & 3, i = tid & 15;
      for (int u = 0; u < 2; ++u) inc[2 * (h * 256 + tid) + u] = wn((long long)((4 * g + 2 * h + u) % r1) * (64 * wv + 4 * i));
    }
    for (int g = 0; g < 4; ++g)
      for (int u = 0; u < 2; ++u) inc[2 * (512 + h * 4 + g) + u] = wn((4 * g + 2 * h + u) % r1);
  }
  // The same for lanes that hold columns 64w + 16j + i, j = 0..3: start at c0 = 64w + i, step W^(16 k1).
  float2* inc2 = inc + 1024 + 16;
  for (int h = 0; h < 2; ++h) {
    for (int tid = 0; tid < 256; ++tid) {
      const int wv = tid >> 6, g = (tid >> 4) & 3, i = tid & 15;
      for (int u = 0; u < 2; ++u) inc2[2 * (h * 256 + tid) + u] = wn((long long)((4 * g + 2 * h + u) % r1) * (64 * wv + i));
    }
    for (int g = 0; g < 4; ++g)
      for (int u = 0; u < 2; ++u) inc2[2 * (512 + h * 4 + g) + u] = wn(16LL * ((4 * g + 2 * h + u) % r1));
  }
}

__device__ __forceinline__ unsigned pack_f16(float re, float im) {
  const __half2 h = __float22half2_rn(make_float2(re, im));
  return *reinterpret_cast<const unsigned*>(&h);
}

// Complex products on packed fp16 (v_pk_mul_f16 + v_pk_fma_f16, half swaps and signs as operand
// modifiers): the pass-2 twiddles and the filter product take 3 instructions per value, rounding
// included, where fp32 arithmetic takes 4 to 7. The values are rounded to fp16 right afterwards anyway;
// the reference multiplies in half2 throughout (thunderfft/include/thunderfft/detail/util_arith.h:65-73).
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
__device__ __forceinline__ h2_t pack_h2(float re, float im) {
  return __builtin_convertvector(f32x2_t{re, im}, h2_t);  // one v_cvt_pk_f16_f32, round to nearest even
}
template <bool CONJ>
__device__ __forceinline__ unsigned cmul_h2(h2_t x, h2_t t) {  // x * t, or x * conj(t)
  const h2_t a = x * t.xx;
  const h2_t xs = CONJ ? h2_t{x.y, -x.x} : h2_t{-x.y, x.x};
  return __builtin_bit_cast(unsigned, __builtin_elementwise_fma(xs, t.yy, a));
}

typedef __bf16 mfma_bf8 __attribute__((ext_vector_type(8)));

// The 16-point DFT of 16 columns on the matrix cores for one 16-bit storage type: this lane's rows of
// the real-ified matrix, and the product with a B fragment (four (re, im) pairs of one column).
//   fp16: operands fp16, one v_mfma_f32_16x16x32_f16 per output part.
//   bf16: operands bf16; the matrix is split into a bf16 high part and a bf16 remainder (two
//         v_mfma_f32_16x16x32_bf16 per output part), which takes its rounding error from 2^-9 to 2^-17
//         and leaves the data's own bf16 rounding as the only error of a pass. Intermediates in bf16
//         keep the fp32 exponent range, so there is no overflow condition beyond the result's own.
struct MfmaF16 {
  u32x4_t re, im;
  // tabs: the lane-ordered tables behind the n natural-order twiddles
  __device__ __forceinline__ void init(const float2* tabs, bool fwd, int lane) {
    const u32x4_t* t = reinterpret_cast<const u32x4_t*>(tabs + Mfma4096::TW_LANE_ORDER) + (fwd ? 0 : 256) + lane;
    re = t[0];
    im = t[64];
  }
  __device__ __forceinline__ void init_pass1(const float2* tabs, bool fwd, int lane) {
    const u32x4_t* t = reinterpret_cast<const u32x4_t*>(tabs + Mfma4096::TW_LANE_ORDER + Mfma4096::FRAG_TABLE +
                                                        Mfma4096::BF16_FRAG_TABLE) + (fwd ? 0 : 128) + lane;
    re = t[0];
    im = t[64];
  }
  __device__ __forceinline__ void dft16(u32x4_t b, mfma_f4& cr, mfma_f4& ci) const {
    const mfma_f4 zero = {0.f, 0.f, 0.f, 0.f};
    const mfma_h8 bv = __builtin_bit_cast(mfma_h8, b);
    cr = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mfma_h8, re), bv, zero, 0, 0, 0);
    ci = __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(mfma_h8, im), bv, zero, 0, 0, 0);
  }
  static __device__ __forceinline__ unsigned pack(float re_, float im_) { return pack_f16(re_, im_); }
  static __device__ __forceinline__ c32 unpack(unsigned v) { return __half22float2(*reinterpret_cast<const __half2*>(&v)); }
  // pass-2 twiddles: kept and applied as packed fp16
  using tw2_t = h2_t;
  static __device__ __forceinline__ tw2_t make_tw2(c32 t) { return pack_h2(t.x, t.y); }
  template <bool FWD>
  static __device__ __forceinline__ unsigned twiddle_pack(float re_, float im_, tw2_t t) { return cmul_h2<!FWD>(pack_h2(re_, im_), t); }
};

template <bool SPLIT>
struct MfmaBF16T {
  u32x4_t re_hi, im_hi, re_lo, im_lo;
  __device__ __forceinline__ void init(const float2* tabs, bool fwd, int lane) {
    const u32x4_t* t = reinterpret_cast<const u32x4_t*>(tabs + Mfma4096::TW_LANE_ORDER + Mfma4096::FRAG_TABLE) +
                       (fwd ? 0 : 256) + lane;
    re_hi = t[0];
    im_hi = t[64];
    re_lo = t[128];
    im_lo = t[192];
  }
  // the pass-1 matrices of N < 4096 are stored unsplit (entries 0, +-1 and +-sqrt(1/2) only)
  __device__ __forceinline__ void init_pass1(const float2* tabs, bool fwd, int lane) {
    static_assert(!SPLIT, "the pass-1 matrix table holds no remainders");
    const u32x4_t* t = reinterpret_cast<const u32x4_t*>(tabs + Mfma4096::TW_LANE_ORDER + Mfma4096::FRAG_TABLE +
                                                        Mfma4096::BF16_FRAG_TABLE) + 256 + (fwd ? 0 : 128) + lane;
    re_hi = t[0];
    im_hi = t[64];
    re_lo = im_lo = u32x4_t{0u, 0u, 0u, 0u};
  }
  __device__ __forceinline__ void dft16(u32x4_t b, mfma_f4& cr, mfma_f4& ci) const {
    const mfma_f4 zero = {0.f, 0.f, 0.f, 0.f};
    const mfma_bf8 bv = __builtin_bit_cast(mfma_bf8, b);
    cr = ci = zero;
    if (SPLIT) {
      cr = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf8, re_lo), bv, zero, 0, 0, 0);
      ci = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf8, im_lo), bv, zero, 0, 0, 0);
    }
    cr = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf8, re_hi), bv, cr, 0, 0, 0);
    ci = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(mfma_bf8, im_hi), bv, ci, 0, 0, 0);
  }
  static __device__ __forceinline__ unsigned pack(float re_, float im_) { return StoreBF16::cvt(make_float2(re_, im_)); }
  static __device__ __forceinline__ c32 unpack(unsigned v) { return make_float2(__uint_as_float(v << 16), __uint_as_float(v & 0xffff0000u)); }
  // pass-2 twiddles: fp32 (8 mantissa bits would not carry a twiddle)
  using tw2_t = c32;
  static __device__ __forceinline__ tw2_t make_tw2(c32 t) { return t; }
  template <bool FWD>
  static __device__ __forceinline__ unsigned twiddle_pack(float re_, float im_, tw2_t t) {
    const c32 y = FWD ? make_float2(re_ * t.x - im_ * t.y, re_ * t.y + im_ * t.x) : make_float2(re_ * t.x + im_ * t.y, im_ * t.x - re_ * t.y);
    return pack(y.x, y.y);
  }
};
using MfmaBF16 = MfmaBF16T<true>;
using MfmaBF16Single = MfmaBF16T<false>;

// Where point p of the natural-order output image lives: padding that grows with the higher index
// bits, chosen per R1 so that the pass-3 writers (lanes = 16 sub-transforms x 4 row groups) spread
// over the banks while the offsets between a lane's 16 values stay compile-time constants; always a
// multiple of four dwords, so the 16-byte readers find their granules whole.
template <int R1>
__device__ __forceinline__ int mfma_out_slot(int p) {
  if (R1 == 8) return p + 16 * (p >> 9) + 8 * (p >> 11);
  if (R1 == 4) return p + 16 * (p >> 8) + 4 * (p >> 10);
  if (R1 == 2) return p + 4 * (p >> 9);
  return p + 4 * (p >> 8);  // R1 = 1, 16
}
template <int R1>
constexpr int mfma_out_words() {
  return 4096 + (R1 == 8 ? 128 : R1 == 4 ? 256 : 64);
}

// DOUBLE: two LDS images (33 KiB, three barriers) instead of one (16.6 KiB, five barriers).
// PERSIST: a resident grid; a workgroup walks tiles blockIdx.x, blockIdx.x + gridDim.x, ... and
// requests the rows of its next tile as soon as the current rows are packed into fragments, so
// that its HBM requests stay in flight under the three passes (16 more live VGPRs).
// R1 = N / 256; a tile is 4096 points = 16 / R1 transforms, `batch` counts transforms.
// INCR: the pass-1 twiddles W^(k1 c) of the four columns c0 .. c0 + 3 of a lane are built from W^(k1 c0)
// and the step W^k1 (32 + 32 bytes of tables per lane instead of 128; 12 more complex multiplications per tile).
// REPEAT (measurement builds only): the three passes run `repeats` times on the tile, the natural-order output
// image of one round being read back as the rows of the next, so that the launch prices the on-chip work
// (registers -> three passes -> LDS image -> registers) without more HBM traffic. Values overflow; timing only.
// CONVOUT (N = 4096, one workgroup per transform): the result leaves in the order the fused convolution reads its
// filter spectra -- 16-byte entry [transform][t][tid] = X[i + 16 (4w + t) + 256 (4g + r)], r = 0..3, which is exactly
// what a lane holds after pass 3 -- instead of natural order: the filter transform and its re-ordering in one kernel.
// Q < 16 (with R1 = 1): N = 16 Q = 64 or 128. A row of 256 points holds 16 / Q whole transforms; pass 2 is the
// block-diagonal I (x) DFT_Q over each transform's Q groups of 16 points, its twiddles are W_N^((k2 mod Q) n3), pass 3
// is the same DFT_16, and a lane's 16 results are Q-strided points of one transform. The batch need not fill the
// last row: loads beyond it re-read the last valid 16 bytes, stores beyond it are dropped.
template <class T16, bool FWD, bool DOUBLE, int MINW, bool PERSIST = false, int R1 = 16, bool INCR = false,
          bool REPEAT = false, bool CONVOUT = false, int Q = 16>
__global__ __launch_bounds__(256, MINW) void fft_mfma4096_kernel(const u32x4_t* in, u32x4_t* out,  // in and out may alias (in-place calls)
                                                                     const float2* __restrict__ tw, long long batch,
                                                                     int xcd_log2, unsigned repeats = 1,
                                                                     unsigned io_mask = 3) {
  static_assert(!REPEAT || !PERSIST, "the repeat build is a one-shot kernel");
  static_assert(!CONVOUT || (R1 == 16 && !PERSIST && !REPEAT), "the filter order is that of one 4096-point transform");
  // measurement build only: bit 0 = load the tile's rows (else start from constants), bit 1 = store the result;
  // run-time flags, so that the rest of the kernel is compiled as it stands
  const bool do_loads = !REPEAT || (io_mask & 1u), do_stores = !REPEAT || (io_mask & 2u);
  static_assert(R1 == 1 || R1 == 2 || R1 == 4 || R1 == 8 || R1 == 16, "N = 256 R1");
  static_assert(!INCR || R1 > 1, "N = 256 has no first pass");
  static_assert(Q == 16 || ((Q == 4 || Q == 8) && R1 == 1 && !REPEAT && !PERSIST), "N = 64, 128: the one-shot N = 256 form");
  constexpr int OUT_WORDS = mfma_out_words<R1>();
  static_assert(mfma_out_words<16>() == Mfma4096::OUT_IMG, "the convolution kernels share the N = 4096 image");
  __shared__ __attribute__((aligned(16))) unsigned lds[OUT_WORDS + (DOUBLE ? Mfma4096::IMG : 0)];
  unsigned* const img_a = lds;                              // pass-2 input, later the output image
  unsigned* const img_b = DOUBLE ? lds + OUT_WORDS : lds;   // pass-3 input

  const int tid = threadIdx.x;
  const int w = tid >> 6, g = (tid >> 4) & 3, i = tid & 15;
  const long long units = batch * (4 * Q * R1);     // 16-byte units (four points) of the whole batch
  const long long rows_total = (units + 63) >> 6;   // rows of 256 points
  const long long tiles = (rows_total + 15) >> 4;
  long long tr = blockIdx.x;
  if (tr >= tiles) return;
  const float2* const tabs = tw + 16 * Q * R1;

  // ---- rows first: lane (g, i) of wave w loads rows 4G + g, columns 64w + 4i .. 4i+3
  // Rows past the batch (last tile when N < 4096) re-read the tile's last valid row: their results are never
  // stored, and the matrix of pass 1 has zeros between different transforms.
  u32x4_t rows[4];
  auto last_row = [&](long long t) { return R1 == 16 ? 15 : (int)min(15LL, rows_total - 16 * t - 1); };
  auto request_rows = [&](long long t) {  // t: tile index
    const u32x4_t* src = in + t * 1024 + 16 * w + i;
    const int last = last_row(t);
#pragma unroll
    for (int G = 0; G < 4; ++G) {
      if constexpr (Q == 16) {
        rows[G] = __builtin_nontemporal_load(src + min(4 * G + g, last) * 64);
      } else {  // the last row may be partly filled
        const long long at = t * 1024 + 16 * w + i + min(4 * G + g, last) * 64;
        rows[G] = __builtin_nontemporal_load(in + min(at, units - 1));
      }
    }
  };
  // tr is the position in the launch order, tile the 4096 points it stands for (xcd_chunk, fft_lds_kernel.hpp)
  long long tile = xcd_chunk(tr, tiles, xcd_log2);
  if (do_loads) {
    request_rows(tile);
  } else {
#pragma unroll
    for (int G = 0; G < 4; ++G) rows[G] = u32x4_t{0x3c003c00u, 0x38003800u, (unsigned)threadIdx.x, 0u};
  }

  // ---- pass-1 twiddles in lane order, the DFT-16 matrix fragments, the pass-2 twiddles
  const float4* tw1 = reinterpret_cast<const float4*>(tabs) + tid;
  float4 t1[R1 == 1 || INCR ? 1 : 8];
  c32 start[4], step[4];
  if constexpr (INCR) {
    // start [h][tid] = { W^(k1 c0), W^(k1' c0) } for rows s = 4g + 2h and s + 1, c0 = 64w + 4i; step [h][g] = { W^k1, W^k1' }
    const float4* ctab = reinterpret_cast<const float4*>(tabs + Mfma4096::TABLES - Mfma4096::INCR_TABLE);
    const float4 q0 = ctab[512 + g], q1 = ctab[516 + g];
    // The start values W^(k1 c0) are evaluated, not fetched: v_sin_f32 / v_cos_f32 take their argument in turns, and
    // (k1 c0 mod n) / n is exact in fp32, so eight transcendental instructions under the wait for the rows replace the
    // one table that differs from lane to lane (8 KiB per tile through L2 and the vector cache).
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int k1 = (4 * g + r) % R1;
      const float turns = (float)((k1 * (64 * w + 4 * i)) & (256 * R1 - 1)) * (1.0f / (256 * R1));
      start[r] = make_float2(__builtin_amdgcn_cosf(turns), -__builtin_amdgcn_sinf(turns));
    }
    step[0] = make_float2(q0.x, q0.y); step[1] = make_float2(q0.z, q0.w);
    step[2] = make_float2(q1.x, q1.y); step[3] = make_float2(q1.z, q1.w);
  } else if constexpr (R1 > 1) {
#pragma unroll
    for (int e = 0; e < 8; ++e) t1[e] = tw1[e * 256];
  }
  const u32x4_t* ftab = reinterpret_cast<const u32x4_t*>(tabs + Mfma4096::TW_LANE_ORDER) + (tid & 63);
  T16 mat;
  mat.init(tabs, FWD, tid & 63);
  T16 mat1 = mat;  // pass 1: DFT_16 when R1 = 16, else I (x) DFT_R1
  if constexpr (R1 > 1 && R1 < 16) mat1.init_pass1(tabs, FWD, tid & 63);
  T16 mat2 = mat;  // pass 2: DFT_16, or I (x) DFT_Q for N = 64, 128
  if constexpr (Q < 16) mat2.init_pass1(tabs, FWD, tid & 63);
  typename T16::tw2_t t2[4];
  {
    // (evaluating these like the start values pays nowhere: N = 256, which has no start values, gains a point x -> y
    // and loses 3.5 in place; at N = 4096 sixteen transcendentals per lane cost 2 points)
    const u32x4_t p = ftab[128], q = ftab[192];
    t2[0] = T16::make_tw2(make_float2(__uint_as_float(p.x), __uint_as_float(p.y)));
    t2[1] = T16::make_tw2(make_float2(__uint_as_float(p.z), __uint_as_float(p.w)));
    t2[2] = T16::make_tw2(make_float2(__uint_as_float(q.x), __uint_as_float(q.y)));
    t2[3] = T16::make_tw2(make_float2(__uint_as_float(q.z), __uint_as_float(q.w)));
  }

  auto twiddle = [](float re, float im, c32 t) {
    // (re + i im) * t forward, * conj(t) inverse; t = (cos, -sin)
    if (FWD) return make_float2(re * t.x - im * t.y, re * t.y + im * t.x);
    return make_float2(re * t.x + im * t.y, im * t.x - re * t.y);
  };
  for (;;) {
  const long long next = tr + gridDim.x;
  const long long next_tile = xcd_chunk(next, tiles, xcd_log2);
  unsigned round = 0;
  for (;;) {  // one trip, or `repeats` trips in the measurement build
  // ---- pass 1: transpose rows <-> lane groups, then 16 columns {64w + 4i + j} per j
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
    packed[j] = u32x4_t{a0, a1, a2, a3};  // rows 4g .. 4g+3 of column 64w + 4i + j
  }
  if (PERSIST && next < tiles) request_rows(next_tile);
  c32 run[4];
  if constexpr (INCR) {
#pragma unroll
    for (int r = 0; r < 4; ++r) run[r] = start[r];
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    mfma_f4 cr, ci;
    if constexpr (R1 > 1) mat1.dft16(packed[j], cr, ci);
    // lane holds Y1[s = 4g + r][n2 = 4w + (i >> 2)][n3 = 4 (i & 3) + j]; image 2 row = (s, n3), point n2
    const int n2 = 4 * w + (i >> 2);
    const int rot = (2 * (i & 3) + (j >> 1) + g) & 3;
    const int pos = (n2 + 4 * rot) & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = (4 * g + r) * 16 + 4 * (i & 3) + j;
      if constexpr (INCR) {
        const c32 y = twiddle(cr[r], ci[r], run[r]);
        img_a[row * 16 + pos] = T16::pack(y.x, y.y);
        if (j < 3) run[r] = cmul(run[r], step[r]);
      } else if constexpr (R1 > 1) {
        const float4 tt = t1[j * 2 + (r >> 1)];
        const c32 t = (r & 1) ? make_float2(tt.z, tt.w) : make_float2(tt.x, tt.y);
        const c32 y = twiddle(cr[r], ci[r], t);
        img_a[row * 16 + pos] = T16::pack(y.x, y.y);
      } else {
        img_a[row * 16 + pos] = packed[j][r];
      }
    }
  }
  __syncthreads();

  // ---- pass 2: wave w owns s = 4w + t; column n3 = i, points n2 = 4g .. 4g+3
  u32x4_t frag[4];
  {
    const int rot = ((i >> 1) + w) & 3;
    const int gran = (g + rot) & 3;
#pragma unroll
    for (int t = 0; t < 4; ++t)
      frag[t] = *reinterpret_cast<const u32x4_t*>(&img_a[((4 * w + t) * 16 + i) * 16 + 4 * gran]);
  }
  if (!DOUBLE) __syncthreads();
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    mfma_f4 cr, ci;
    mat2.dft16(frag[t], cr, ci);
    // lane holds Y2[s = 4w + t][k2 = 4g + r][n3 = i]; image 3 row = (k2, s), point n3
    const int rot = (t + 2 * w + (t >> 1)) & 3;
    const int pos = (i + 4 * rot) & 15;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int row = (4 * g + r) * 16 + 4 * w + t;
      img_b[row * 16 + pos] = T16::template twiddle_pack<FWD>(cr[r], ci[r], t2[r]);
    }
  }
  __syncthreads();

  // ---- pass 3: wave w owns k2 = 4w + t; column s = i, points n3 = 4g .. 4g+3
  {
    const int rot = (i + (i >> 1)) & 3;
    const int gran = (g + rot) & 3;
#pragma unroll
    for (int t = 0; t < 4; ++t)
      frag[t] = *reinterpret_cast<const u32x4_t*>(&img_b[((4 * w + t) * 16 + i) * 16 + 4 * gran]);
  }
  if (!DOUBLE) __syncthreads();
  // lane holds X of sub-transform s = i at k2 = 4w + t, k3 = 4g + r: point
  // p = (i / R1) N + (i mod R1) + R1 (k2 + 16 k3) of the tile
  // (the padding of mfma_out_slot is the same for a lane's 16 values, or grows by 4 per r when R1 = 16)
  if constexpr (CONVOUT) {
    u32x4_t* dst = out + tile * 1024 + tid;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      mfma_f4 cr, ci;
      mat.dft16(frag[t], cr, ci);
      __builtin_nontemporal_store(
          u32x4_t{T16::pack(cr[0], ci[0]), T16::pack(cr[1], ci[1]), T16::pack(cr[2], ci[2]), T16::pack(cr[3], ci[3])},
          dst + 256 * t);
    }
    return;
  }
  // N = 64, 128: k2 = 4w + t is output ka = k2 mod Q of transform k2 / Q of row i: point 16 Q (k2 / Q) + ka + Q k3
  const int slot0 = Q < 16 ? mfma_out_slot<1>(i * 256 + 16 * Q * ((4 * w) / Q) + (4 * w) % Q + Q * (4 * g))
                           : mfma_out_slot<R1>((i / R1) * (256 * R1) + (i % R1) + R1 * (4 * w) + 16 * R1 * (4 * g));
  constexpr int R_STEP = Q < 16 ? Q : 16 * R1 + (R1 == 16 ? 4 : 0);
  unsigned last[4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t) {
    mfma_f4 cr, ci;
    mat.dft16(frag[t], cr, ci);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const unsigned v = T16::pack(cr[r], ci[r]);
      if constexpr (R1 == 1) last[r][t] = v;  // t = 0..3 are four adjacent points: one 16-byte write per r below
      else img_a[slot0 + R1 * t + R_STEP * r] = v;
    }
  }
  if constexpr (R1 == 1) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      *reinterpret_cast<u32x4_t*>(&img_a[slot0 + R_STEP * r]) = u32x4_t{last[r][0], last[r][1], last[r][2], last[r][3]};
  }
  __syncthreads();
  if (!REPEAT || ++round >= repeats) break;
  // next round: this lane's rows 4G + g, columns 64w + 4i .. 4i + 3, back from the natural-order image
#pragma unroll
  for (int G = 0; G < 4; ++G)
    rows[G] = *reinterpret_cast<const u32x4_t*>(&img_a[mfma_out_slot<R1>(4 * ((4 * G + g) * 64 + 16 * w + i))]);
  __syncthreads();
  }

  // ---- natural order out, 16 bytes per lane
  const int last_cur = last_row(tile);
  u32x4_t* dst = out + tile * 1024 + tid;
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const u32x4_t v = *reinterpret_cast<const u32x4_t*>(&img_a[mfma_out_slot<R1>(4 * (tid + 256 * q))]);
    if constexpr (Q < 16) {
      if (tile * 1024 + tid + 256 * q < units) __builtin_nontemporal_store(v, dst + 256 * q);
    } else {
      if (do_stores && (R1 == 16 || 4 * q + w <= last_cur)) __builtin_nontemporal_store(v, dst + 256 * q);
    }
  }
  if (!PERSIST || next >= tiles) break;
  __syncthreads();  // the output image is read before the next tile's pass 1 overwrites it
  tr = next;
  tile = next_tile;
  }
}

// N = 4096 R0, R0 = 2 or 4, 16-bit storage: decimation in time over R0 around the 4096-point passes.
// The R0 interleaved subsequences x[R0 n' + r0] are transformed one after the other (F_r0, the three
// matrix-core passes above through the same two LDS images), and
//   X[k' + 4096 k0] = sum_r0 W_R0^(r0 k0) W_N^(r0 k') F_r0[k']
// is a radix-R0 butterfly across results that sit in the same lane and register slot, in fp32. The R0
// points n' = c of the subsequences are 4 R0 contiguous bytes, one load of a lane: de-interleaving is
// register naming, and a lane takes columns c = 64w + 16j + i, j = 0..3, so that the 16 lanes of a row read
// contiguous memory. Every output block k0 is 4096 contiguous points and leaves through the
// natural-order image like a 4096-point transform. One workgroup per transform. The final twiddles are
// W^(r0 (i + 64w + 1024g)) (one gather per lane from the natural-order table) times the wave-uniform
// W^(16 r0 t) and W^(256 r0 r). `tw` holds the N natural-order twiddles followed by the lane-ordered
// tables of N = 4096.
// H2 (fp16 storage): the pass-1 twiddles of the R0 sub-transforms come rounded to fp16 from a lane-order table
// (16 KiB per transform, read once for all R0 of them) and are applied with the packed product of the pass-2
// twiddles, instead of being built R0 times over from start and step values in fp32.
template <class T16, bool FWD, int R0, int MINW, bool H2 = false>
__global__ __launch_bounds__(256, MINW) void fft_mfma_big_kernel(const u32x4_t* in, u32x4_t* out,  // in and out may alias
                                                                     const float2* __restrict__ tw, long long batch,
                                                                     int xcd_log2) {
  static_assert(R0 == 2 || R0 == 4, "N = 8192 or 16384");
  constexpr int N = 4096 * R0;
  constexpr int OUT_WORDS = mfma_out_words<16>();
  __shared__ __attribute__((aligned(16))) unsigned lds[2 * OUT_WORDS];
  unsigned* const img_a = lds;              // pass-2 input; at the end the output image of even k0
  unsigned* const img_b = lds + OUT_WORDS;  // pass-3 input; at the end the output image of odd k0

  const int tid = threadIdx.x;
  const int w = tid >> 6, g = (tid >> 4) & 3, i = tid & 15;
  if (blockIdx.x >= batch) return;
  const long long tr = xcd_chunk(blockIdx.x, batch, xcd_log2);
  const float2* const tabs = tw + N;

  // ---- all rows first: lane (g, i) of wave w, row rho = 4G + g, columns c = 64w + 16j + i: points R0 (256 rho + c) + r0
  typedef unsigned vec_t __attribute__((ext_vector_type(R0)));
  vec_t L[4][4];
  {
    const vec_t* src = reinterpret_cast<const vec_t*>(in) + tr * 4096 + 64 * w + i;
#pragma unroll
    for (int G = 0; G < 4; ++G)
#pragma unroll
      for (int j = 0; j < 4; ++j) L[G][j] = __builtin_nontemporal_load(src + 256 * (4 * G + g) + 16 * j);
  }
  // N = 8192: nothing moves across this line, so that every row load is out before the table loads below (fp16 x -> y
  // 72.4-73.4 -> 74.4-74.7 % of 8 TB/s in an A/B of the two builds, bf16 level; the kernel also loses its 8 bytes of
  // scratch). Not for N = 16384: there the scheduler, left alone, puts the tables between the rows and holds the
  // last four rows back until the first tables have arrived, and that order measured 1-1.5 points FASTER for fp16
  // (level for bf16) than all sixteen rows first.
  if constexpr (R0 == 2) __builtin_amdgcn_sched_barrier(0);

  // ---- constants of the 4096-point passes
  T16 mat;
  mat.init(tabs, FWD, tid & 63);
  c32 start[4], step[4];
  u32x4_t t1h[4];
  typename T16::tw2_t t2[4];
  {
    if constexpr (H2) {
      // behind the lane-ordered tables: packed fp16 W^(k1 c), 16-byte entry [map][j][tid], map 1 = columns 64w + 16j + i
      // (evaluating the sixteen values instead measured level at N = 8192 and 0.4 points slower at N = 16384)
      const u32x4_t* htab = reinterpret_cast<const u32x4_t*>(tabs + Mfma4096::TABLES) + 1024 + tid;
#pragma unroll
      for (int j = 0; j < 4; ++j) t1h[j] = htab[256 * j];
    } else {
      const float4* ctab = reinterpret_cast<const float4*>(tabs + Mfma4096::TABLES - Mfma4096::INCR_TABLE / 2);
      const float4 q0 = ctab[512 + g], q1 = ctab[516 + g];
      if constexpr (std::is_same<T16, MfmaF16>::value && R0 == 4) {
        // (the fp16 A/B form of N = 16384 keeps its table: with the evaluation it spills 36 bytes per lane)
        const float4 s0 = ctab[tid], s1 = ctab[256 + tid];
        start[0] = make_float2(s0.x, s0.y); start[1] = make_float2(s0.z, s0.w);
        start[2] = make_float2(s1.x, s1.y); start[3] = make_float2(s1.z, s1.w);
      } else {
        // start values W^(k1 (64w + i)): evaluated under the wait for the rows (see fft_mfma4096_kernel; bf16
        // N = 8192 +0.4 ... 0.8 points, N = 16384 +0.2 ... 0.4)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float turns = (float)(((4 * g + r) * (64 * w + i)) & 4095) * (1.0f / 4096);
          start[r] = make_float2(__builtin_amdgcn_cosf(turns), -__builtin_amdgcn_sinf(turns));
        }
      }
      step[0] = make_float2(q0.x, q0.y); step[1] = make_float2(q0.z, q0.w);
      step[2] = make_float2(q1.x, q1.y); step[3] = make_float2(q1.z, q1.w);
    }
    const u32x4_t* ftab = reinterpret_cast<const u32x4_t*>(tabs + Mfma4096::TW_LANE_ORDER) + (tid & 63);
    const u32x4_t p = ftab[128], q = ftab[192];
    t2[0] = T16::make_tw2(make_float2(__uint_as_float(p.x), __uint_as_float(p.y)));
    t2[1] = T16::make_tw2(make_float2(__uint_as_float(p.z), __uint_as_float(p.w)));
    t2[2] = T16::make_tw2(make_float2(__uint_as_float(q.x), __uint_as_float(q.y)));
    t2[3] = T16::make_tw2(make_float2(__uint_as_float(q.z), __uint_as_float(q.w)));
  }
  auto twiddle = [](float re, float im, c32 t) {
    if (FWD) return make_float2(re * t.x - im * t.y, re * t.y + im * t.x);
    return make_float2(re * t.x + im * t.y, im * t.x - re * t.y);
  };

  unsigned res[R0][4][4];  // [r0][t][r]: F_r0[k' = i + 16 (4w + t) + 256 (4g + r)], later X[k' + 4096 k0] at [k0]
#pragma unroll
  for (int r0 = 0; r0 < R0; ++r0) {
    // ---- pass 1: this subsequence's dwords, rows <-> lane groups, 16 columns {64w + 16j + i} per j
    c32 run[4];
    if constexpr (!H2) {
#pragma unroll
      for (int r = 0; r < 4; ++r) run[r] = start[r];
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      unsigned a[4];
#pragma unroll
      for (int G = 0; G < 4; ++G) a[G] = L[G][j][r0];
      auto s = __builtin_amdgcn_permlane32_swap(a[0], a[2], false, false);
      a[0] = s[0]; a[2] = s[1];
      s = __builtin_amdgcn_permlane32_swap(a[1], a[3], false, false);
      a[1] = s[0]; a[3] = s[1];
      s = __builtin_amdgcn_permlane16_swap(a[0], a[1], false, false);
      a[0] = s[0]; a[1] = s[1];
      s = __builtin_amdgcn_permlane16_swap(a[2], a[3], false, false);
      a[2] = s[0]; a[3] = s[1];
      mfma_f4 cr, ci;
      mat.dft16(u32x4_t{a[0], a[1], a[2], a[3]}, cr, ci);
      // column 16 n2 + n3 with n2 = 4w + j, n3 = i
      const int rot = ((i >> 1) + g) & 3;
      const int pos = (4 * w + j + 4 * rot) & 15;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if constexpr (H2) {
          img_a[((4 * g + r) * 16 + i) * 16 + pos] =
              cmul_h2<!FWD>(pack_h2(cr[r], ci[r]), __builtin_bit_cast(h2_t, (unsigned)t1h[j][r]));
        } else {
          const c32 y = twiddle(cr[r], ci[r], run[r]);
          img_a[((4 * g + r) * 16 + i) * 16 + pos] = T16::pack(y.x, y.y);
          if (j < 3) run[r] = cmul(run[r], step[r]);
        }
      }
    }
    __syncthreads();
    // ---- pass 2
    u32x4_t frag[4];
    {
      const int gran = (g + (i >> 1) + w) & 3;
#pragma unroll
      for (int t = 0; t < 4; ++t)
        frag[t] = *reinterpret_cast<const u32x4_t*>(&img_a[((4 * w + t) * 16 + i) * 16 + 4 * gran]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      mfma_f4 cr, ci;
      mat.dft16(frag[t], cr, ci);
      const int rot = (t + 2 * w + (t >> 1)) & 3;
      const int pos = (i + 4 * rot) & 15;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        img_b[((4 * g + r) * 16 + 4 * w + t) * 16 + pos] = T16::template twiddle_pack<FWD>(cr[r], ci[r], t2[r]);
    }
    __syncthreads();
    // ---- pass 3
    {
      const int gran = (g + i + (i >> 1)) & 3;
#pragma unroll
      for (int t = 0; t < 4; ++t)
        frag[t] = *reinterpret_cast<const u32x4_t*>(&img_b[((4 * w + t) * 16 + i) * 16 + 4 * gran]);
    }
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      mfma_f4 cr, ci;
      mat.dft16(frag[t], cr, ci);
#pragma unroll
      for (int r = 0; r < 4; ++r) res[r0][t][r] = T16::pack(cr[r], ci[r]);
    }
  }

  // ---- across the subsequences: twiddle W_N^(r0 k'), radix-R0 butterfly
  c32 bu[R0][4];  // [r0][t] = W^(r0 (i + 64w + 1024g + 16t))
  c32 v4[R0][4];  // [r0][r] = W^(256 r0 r)
#pragma unroll
  for (int r0 = 1; r0 < R0; ++r0) {
    const c32 base = tw[(r0 * (i + 64 * w + 1024 * g)) & (N - 1)];
#pragma unroll
    for (int t = 0; t < 4; ++t) bu[r0][t] = t == 0 ? base : cmul(base, tw[16 * r0 * t]);
#pragma unroll
    for (int r = 1; r < 4; ++r) v4[r0][r] = tw[256 * r0 * r];
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      c32 v[R0];
      v[0] = T16::unpack(res[0][t][r]);
#pragma unroll
      for (int r0 = 1; r0 < R0; ++r0) {
        const c32 f = T16::unpack(res[r0][t][r]);
        const c32 tt = r == 0 ? bu[r0][t] : cmul(bu[r0][t], v4[r0][r]);
        v[r0] = twiddle(f.x, f.y, tt);
      }
      if constexpr (R0 == 2) {
        res[0][t][r] = T16::pack(v[0].x + v[1].x, v[0].y + v[1].y);
        res[1][t][r] = T16::pack(v[0].x - v[1].x, v[0].y - v[1].y);
      } else {
        const c32 a = make_float2(v[0].x + v[2].x, v[0].y + v[2].y), b = make_float2(v[0].x - v[2].x, v[0].y - v[2].y);
        const c32 c = make_float2(v[1].x + v[3].x, v[1].y + v[3].y), d = make_float2(v[1].x - v[3].x, v[1].y - v[3].y);
        // forward: y1 = b - i d, y3 = b + i d; inverse the other way round
        const c32 ym = make_float2(b.x + d.y, b.y - d.x), yp = make_float2(b.x - d.y, b.y + d.x);
        res[0][t][r] = T16::pack(a.x + c.x, a.y + c.y);
        res[2][t][r] = T16::pack(a.x - c.x, a.y - c.y);
        res[1][t][r] = FWD ? T16::pack(ym.x, ym.y) : T16::pack(yp.x, yp.y);
        res[3][t][r] = FWD ? T16::pack(yp.x, yp.y) : T16::pack(ym.x, ym.y);
      }
    }

  // ---- blocks k0 out in natural order, two at a time through the two images
  const int slot0 = mfma_out_slot<16>(i + 64 * w + 1024 * g);
#pragma unroll
  for (int k0 = 0; k0 < R0; k0 += 2) {
    __syncthreads();  // the images' previous readers (pass 3, or the previous pair's stores) are done
#pragma unroll
    for (int h = 0; h < 2; ++h)
#pragma unroll
      for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) (h ? img_b : img_a)[slot0 + 16 * t + 260 * r] = res[k0 + h][t][r];
    __syncthreads();
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      u32x4_t* dst = out + tr * (N / 4) + (k0 + h) * 1024 + tid;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const u32x4_t v = *reinterpret_cast<const u32x4_t*>(&(h ? img_b : img_a)[mfma_out_slot<16>(4 * (tid + 256 * q))]);
        __builtin_nontemporal_store(v, dst + 256 * q);
      }
    }
  }
}

}  // namespace tfft
