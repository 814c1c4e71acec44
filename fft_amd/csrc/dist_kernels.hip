// Local passes of the distributed four-step transform (fft_amd/distributed.py): between the all-to-all
// exchanges every rank reorders its shard and combines the chunks it received. Both kernels are HBM-bound,
// one read and one write of every point, 16-byte accesses. The reference has no distributed or multi-GPU
// code (single cudaSetDevice per sample, thunderfft/sample/fft1d/main.cu:80); there is nothing to cite.
//
//   tfft_dist_permute : batched 2-D copy through a 64 x 64 LDS tile, transposing or not, optionally times the
//                       four-step twiddle W_n^(p q) with p, q affine in the element's indices (two-level table)
//   tfft_dist_radix   : radix-P butterfly across the P chunks a rank holds after an exchange, times the
//                       twiddle W_M^(t k) -- the step that turns "P chunks of rows of length T" into P batches
//                       of rows whose length-T transforms finish a length-M = P T transform
//
// Inverse transforms run the forward arithmetic on data with real and imaginary parts swapped at load and
// store (multiplying a swapped value by w equals swapping the product with conj(w)).
#include <hip/hip_runtime.h>

#include <cstdint>

#include "../../include/thunderfft_c.h"
#include "butterfly.hpp"

namespace {

using tfft::c32;
typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int kTile = 64;

template <bool TRANSPOSE, bool TWIDDLE, bool FWD>
__global__ __launch_bounds__(256) void dist_permute_kernel(const float2* __restrict__ in, float2* __restrict__ out,
                                                           const tfft_dist_permute_args a, const float2* __restrict__ tw_hi,
                                                           const float2* __restrict__ tw_lo) {
  __shared__ float2 tile[kTile][kTile + 1];
  const int tiles_j = (int)((a.nj + kTile - 1) / kTile), tiles_i = (int)((a.ni + kTile - 1) / kTile);
  long long t = blockIdx.x;
  const int tj = (int)(t % tiles_j);
  t /= tiles_j;
  const int ti = (int)(t % tiles_i);
  t /= tiles_i;
  const long long b2 = t % a.nb2, b1 = t / a.nb2;
  const long long i0 = (long long)ti * kTile, j0 = (long long)tj * kTile;
  const float2* src = in + b1 * a.in_b1 + b2 * a.in_b2;
  float2* dst = out + b1 * a.out_b1 + b2 * a.out_b2;

  const int lane2 = (threadIdx.x & 31) * 2, row0 = threadIdx.x >> 5;  // two adjacent points per thread, 8 rows per step
  auto load2 = [&](long long i, long long j, c32& p, c32& q) {
    const f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(src + i * a.in_i + j));
    p = FWD ? make_float2(v.x, v.y) : make_float2(v.y, v.x);
    q = FWD ? make_float2(v.z, v.w) : make_float2(v.w, v.z);
  };
  // Four-step twiddles of this thread's two columns j, j + 1 for its rows i = i0 + row0 + 8 k, k < 8:
  //   W^(p_k q) = W^(p_0 q) (W^(8 pi q))^k,   p_k = tw_p0 + tw_pi i,
  // so two table look-ups per column (base and step) and a short power tree replace a look-up per element
  // (the gathers of a look-up per element, not the arithmetic, were what this pass spent its time on).
  c32 tw_k[2][kTile / 8];
  if constexpr (TWIDDLE) {
    auto root = [&](unsigned long long e) {
      e &= (unsigned long long)(a.tw_n - 1);
      return tfft::cmul(tw_hi[e >> a.tw_shift], tw_lo[e & ((1ull << a.tw_shift) - 1)]);
    };
    const unsigned long long mask = (unsigned long long)(a.tw_n - 1);
    const unsigned long long p_first = (unsigned long long)(a.tw_p0 + a.tw_pi * (i0 + row0)) & mask;
    const unsigned long long p_step = (unsigned long long)(8 * a.tw_pi) & mask;
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      const unsigned long long q =
          (unsigned long long)(a.tw_q0 + a.tw_qb1 * b1 + a.tw_qb2 * b2 + a.tw_qj * (j0 + lane2 + c)) & mask;
      const c32 base = root(p_first * q);
      c32 pw[kTile / 8];
      tfft::twiddle_powers<kTile / 8>(root(p_step * q), pw);
      tw_k[c][0] = base;
#pragma unroll
      for (int k = 1; k < kTile / 8; ++k) tw_k[c][k] = tfft::cmul(base, pw[k]);
    }
  }
  auto twiddle = [&](c32 v, int k, int c) {
    if constexpr (!TWIDDLE) return v;
    return tfft::cmul(v, tw_k[c][k]);
  };
  auto store2 = [&](float2* p, c32 x, c32 y) {
    const f32x4_t v = FWD ? f32x4_t{x.x, x.y, y.x, y.y} : f32x4_t{x.y, x.x, y.y, y.x};
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4_t*>(p));
  };

  if constexpr (!TRANSPOSE) {
    // plain strided copy: out[i][j] = in[i][j], row pitches in_i and out_j
#pragma unroll
    for (int k = 0; k < kTile / 8; ++k) {
      const long long i = i0 + row0 + 8 * k, j = j0 + lane2;
      if (i < a.ni && j < a.nj) {
        c32 p, q;
        load2(i, j, p, q);
        store2(dst + i * a.out_j + j, twiddle(p, k, 0), twiddle(q, k, 1));
      }
    }
  } else {
#pragma unroll
    for (int k = 0; k < kTile / 8; ++k) {
      const int r = row0 + 8 * k;
      const long long i = i0 + r, j = j0 + lane2;
      if (i < a.ni && j < a.nj) {
        c32 p, q;
        load2(i, j, p, q);
        tile[r][lane2] = twiddle(p, k, 0);
        tile[r][lane2 + 1] = twiddle(q, k, 1);
      }
    }
    __syncthreads();
    // out[j][i]: two adjacent i per thread
#pragma unroll
    for (int k = 0; k < kTile / 8; ++k) {
      const int c = row0 + 8 * k;
      const long long j = j0 + c, i = i0 + lane2;
      if (j < a.nj && i < a.ni) {
        const c32 p = tile[lane2][c], q = tile[lane2 + 1][c];
        // (the swap of the inverse was applied at load and is undone here by store2)
        store2(dst + j * a.out_j + i, p, q);
      }
    }
  }
}

template <int P, bool FWD>
__global__ __launch_bounds__(256) void dist_radix_kernel(const float2* in, float2* out, const tfft_dist_radix_args a,
                                                         const float2* __restrict__ tab) {
  const long long pairs_per_row = a.row_len / 2;
  const long long total = a.rows * pairs_per_row;
  for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
       idx += (long long)gridDim.x * blockDim.x) {
    const long long row = idx / pairs_per_row;
    const long long t = (idx - row * pairs_per_row) * 2;
    c32 x[P], y[P];
#pragma unroll
    for (int s = 0; s < P; ++s) {
      const f32x4_t v =
          __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(in + s * a.in_chunk + row * a.in_pitch + t));
      x[s] = FWD ? make_float2(v.x, v.y) : make_float2(v.y, v.x);
      y[s] = FWD ? make_float2(v.z, v.w) : make_float2(v.w, v.z);
    }
    tfft::dft<P>(x);
    tfft::dft<P>(y);
    // W_M^(t k) and W_M^((t + 1) k), k < P, as powers of the two adjacent table entries W_M^t, W_M^(t + 1) (one
    // coalesced 16-byte load; gathering every power from the table would cost more cache lines than the data)
    const f32x4_t w = *reinterpret_cast<const f32x4_t*>(tab + t);
    c32 px[P], py[P];
    tfft::twiddle_powers<P>(make_float2(w.x, w.y), px);
    tfft::twiddle_powers<P>(make_float2(w.z, w.w), py);
#pragma unroll
    for (int k = 0; k < P; ++k) {
      if (k > 0) {
        x[k] = tfft::cmul(x[k], px[k]);
        y[k] = tfft::cmul(y[k], py[k]);
      }
      const f32x4_t v = FWD ? f32x4_t{x[k].x, x[k].y, y[k].x, y[k].y} : f32x4_t{x[k].y, x[k].x, y[k].y, y[k].x};
      __builtin_nontemporal_store(v, reinterpret_cast<f32x4_t*>(out + k * a.out_chunk + row * a.out_pitch + t));
    }
  }
}

template <bool TRANSPOSE, bool TWIDDLE>
void launch_permute(const void* in, void* out, const tfft_dist_permute_args& a, const void* hi, const void* lo, int forward,
                    unsigned grid, hipStream_t s) {
  auto f = static_cast<const float2*>(in);
  auto o = static_cast<float2*>(out);
  auto h = static_cast<const float2*>(hi), l = static_cast<const float2*>(lo);
  if (forward)
    dist_permute_kernel<TRANSPOSE, TWIDDLE, true><<<dim3(grid), dim3(256), 0, s>>>(f, o, a, h, l);
  else
    dist_permute_kernel<TRANSPOSE, TWIDDLE, false><<<dim3(grid), dim3(256), 0, s>>>(f, o, a, h, l);
}

template <int P>
void launch_radix(const void* in, void* out, const tfft_dist_radix_args& a, const void* tab, int forward, unsigned grid,
                  hipStream_t s) {
  auto f = static_cast<const float2*>(in);
  auto o = static_cast<float2*>(out);
  auto t = static_cast<const float2*>(tab);
  if (forward)
    dist_radix_kernel<P, true><<<dim3(grid), dim3(256), 0, s>>>(f, o, a, t);
  else
    dist_radix_kernel<P, false><<<dim3(grid), dim3(256), 0, s>>>(f, o, a, t);
}

bool pow2(long long v) { return v > 0 && (v & (v - 1)) == 0; }
bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

}  // namespace

// Argument checks live in tfft_c_api.hip's style: return a TFFT_ERR_* code; the message is set by the caller
// wrappers below through tfft_internal_fail.
extern "C" int tfft_internal_fail(int code, const char* message);

extern "C" int tfft_dist_permute(const void* d_in, void* d_out, const tfft_dist_permute_args* args, const void* d_tw_hi,
                                 const void* d_tw_lo, int forward, void* stream) {
  if (!args) return tfft_internal_fail(TFFT_ERR_INVALID, "tfft_dist_permute: null arguments");
  const tfft_dist_permute_args& a = *args;
  if (a.nb1 < 0 || a.nb2 < 0 || a.ni < 0 || a.nj < 0)
    return tfft_internal_fail(TFFT_ERR_INVALID, "tfft_dist_permute: negative extent");
  if (a.nb1 == 0 || a.nb2 == 0 || a.ni == 0 || a.nj == 0) return TFFT_OK;
  if (!d_in || !d_out || d_in == d_out || !aligned16(d_in) || !aligned16(d_out))
    return tfft_internal_fail(TFFT_ERR_INVALID, "tfft_dist_permute: buffers must be distinct, non-null and 16-byte aligned");
  // 16-byte accesses: pairs of adjacent points must stay adjacent and aligned on both sides
  const long long ev = a.transpose ? (a.ni | a.nj | a.in_i | a.out_j | a.in_b1 | a.in_b2 | a.out_b1 | a.out_b2)
                                   : (a.nj | a.in_i | a.out_j | a.in_b1 | a.in_b2 | a.out_b1 | a.out_b2);
  if (ev & 1) return tfft_internal_fail(TFFT_ERR_INVALID, "tfft_dist_permute: extents and strides must be even");
  if (a.twiddle) {
    if (!d_tw_hi || !d_tw_lo || !pow2(a.tw_n) || a.tw_shift < 0 || a.tw_shift > 40)
      return tfft_internal_fail(TFFT_ERR_INVALID, "tfft_dist_permute: twiddle needs both tables and a power-of-two length");
  }
  const long long tiles = a.nb1 * a.nb2 * ((a.ni + kTile - 1) / kTile) * ((a.nj + kTile - 1) / kTile);
  if (tiles > 0x7fffffffLL) return tfft_internal_fail(TFFT_ERR_INVALID, "tfft_dist_permute: too many tiles for one launch");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const unsigned grid = (unsigned)tiles;
  if (a.transpose) {
    if (a.twiddle) launch_permute<true, true>(d_in, d_out, a, d_tw_hi, d_tw_lo, forward, grid, s);
    else launch_permute<true, false>(d_in, d_out, a, d_tw_hi, d_tw_lo, forward, grid, s);
  } else {
    if (a.twiddle) launch_permute<false, true>(d_in, d_out, a, d_tw_hi, d_tw_lo, forward, grid, s);
    else launch_permute<false, false>(d_in, d_out, a, d_tw_hi, d_tw_lo, forward, grid, s);
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return tfft_internal_fail(TFFT_ERR_HIP, hipGetErrorString(e));
  return TFFT_OK;
}

extern "C" int tfft_dist_radix(const void* d_in, void* d_out, const tfft_dist_radix_args* args, const void* d_table,
                               int forward, void* stream) {
  if (!args) return tfft_internal_fail(TFFT_ERR_INVALID, "tfft_dist_radix: null arguments");
  const tfft_dist_radix_args& a = *args;
  if (a.rows < 0 || a.row_len < 0) return tfft_internal_fail(TFFT_ERR_INVALID, "tfft_dist_radix: negative extent");
  if (a.rows == 0 || a.row_len == 0) return TFFT_OK;
  if (a.parts != 2 && a.parts != 4 && a.parts != 8)
    return tfft_internal_fail(TFFT_ERR_UNSUPPORTED, "tfft_dist_radix: 2, 4 or 8 chunks");
  if (!d_in || !d_out || !d_table || !aligned16(d_in) || !aligned16(d_out))
    return tfft_internal_fail(TFFT_ERR_INVALID, "tfft_dist_radix: buffers must be non-null and 16-byte aligned");
  if (!pow2(a.row_len) || ((a.row_len | a.in_chunk | a.in_pitch | a.out_chunk | a.out_pitch) & 1))
    return tfft_internal_fail(TFFT_ERR_INVALID, "tfft_dist_radix: row length a power of two >= 2, strides even");
  if (d_in == d_out && (a.in_chunk != a.out_chunk || a.in_pitch != a.out_pitch))
    return tfft_internal_fail(TFFT_ERR_INVALID, "tfft_dist_radix: in place needs equal strides on both sides");
  const long long total = a.rows * (a.row_len / 2);
  long long grid = (total + 255) / 256;
  if (grid > 256 * 32) grid = 256 * 32;
  hipStream_t s = static_cast<hipStream_t>(stream);
  switch (a.parts) {
    case 2: launch_radix<2>(d_in, d_out, a, d_table, forward, (unsigned)grid, s); break;
    case 4: launch_radix<4>(d_in, d_out, a, d_table, forward, (unsigned)grid, s); break;
    default: launch_radix<8>(d_in, d_out, a, d_table, forward, (unsigned)grid, s); break;
  }
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return tfft_internal_fail(TFFT_ERR_HIP, hipGetErrorString(e));
  return TFFT_OK;
}
