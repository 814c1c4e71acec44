// Local passes of the distributed four-step transform (fft_amd/distributed.py): between the all-to-all
// exchanges every rank reorders its shard and combines the chunks it received. Both kernels are HBM-bound,
// one read and one write of every point, 16-byte accesses. The reference has no distributed or multi-GPU
// code (single cudaSetDevice per sample, thunderfft/sample/fft1d/main.cu:80); there is nothing to cite.
//
//   tfft_dist_permute : batched 2-D copy, transposing (through an LDS tile of 4096 points: 64 x 64, 128 x 32 or
//                       32 x 128) or not (row segments straight through registers), optionally times the four-step
//                       twiddle W_n^(p q) with p, q affine in the element's indices (two-level table)
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
#include "fft_lds_kernel.hpp"

namespace {

using tfft::c32;
typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr int kTile = 64;
// The dispatcher deals workgroups to the eight XCDs round-robin; both kernels re-map their block index so that each
// XCD works on runs of 16 consecutive tiles (tfft::xcd_chunk), which measured +4 ... 8 % over block order on 256 MiB
// shards (runs of 4 and 64: the same within a point).
constexpr int kXcdRunLog2 = 4;
// The copy that does not transpose moves whole row segments: 8 rows x 512 points per workgroup, 1 KiB per wave load.
constexpr int kCopyRows = 8, kCopyCols = 512;

// TI x TJ = 4096 points per workgroup: a transposing pass reads TI rows of TJ points and writes TJ rows of TI points.
// 64 x 64 by default; the side whose rows lie further apart gets the short segments and the other side 1 KiB ones
// (128 x 32 when the input pitch is the larger one, 32 x 128 when the output pitch is; measured +2 ... 3 % on 256 MiB
// and 1 GiB shards, while walking the tiles along i instead of j first lost 5 ... 10 %). The twiddling pass keeps
// 64 x 64 (its twiddle powers step along the rows a thread reads). A copying pass moves 8 rows of 512 points.
template <bool TRANSPOSE, bool TWIDDLE, bool FWD, int TI, int TJ>
__global__ __launch_bounds__(256) void dist_permute_kernel(const float2* __restrict__ in, float2* __restrict__ out,
                                                           const tfft_dist_permute_args a, const float2* __restrict__ tw_hi,
                                                           const float2* __restrict__ tw_lo) {
  static_assert(TI * TJ == 4096 && TJ >= 2 && TI >= 2, "4096 points per workgroup, two adjacent points per thread");
  __shared__ float2 tile[TRANSPOSE ? TI : 1][TJ + 1];
  const int tiles_j = (int)((a.nj + TJ - 1) / TJ), tiles_i = (int)((a.ni + TI - 1) / TI);
  long long t = tfft::xcd_chunk(blockIdx.x, gridDim.x, kXcdRunLog2);
  const int tj = (int)(t % tiles_j);  // consecutive workgroups: adjacent segments of the same input rows
  t /= tiles_j;
  const int ti = (int)(t % tiles_i);
  t /= tiles_i;
  const long long b2 = t % a.nb2, b1 = t / a.nb2;
  const long long i0 = (long long)ti * TI, j0 = (long long)tj * TJ;
  const float2* src = in + b1 * a.in_b1 + b2 * a.in_b2;
  float2* dst = out + b1 * a.out_b1 + b2 * a.out_b2;

  // two adjacent points per thread: TJ / 2 threads per input row, RS rows per step, K = 8 steps
  constexpr int TPR = TJ / 2, RS = 256 / TPR, K = TI / RS;
  static_assert(K == 8, "eight 16-byte loads per thread");
  const int lane2 = (threadIdx.x % TPR) * 2, row0 = threadIdx.x / TPR;
  auto load2 = [&](long long i, long long j, c32& p, c32& q) {
    const f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(src + i * a.in_i + j));
    p = FWD ? make_float2(v.x, v.y) : make_float2(v.y, v.x);
    q = FWD ? make_float2(v.z, v.w) : make_float2(v.w, v.z);
  };
  // All of a thread's loads are issued before anything waits for one of them: rows and columns beyond the extents
  // re-read the last valid pair (loads are unconditional, so nothing sits between them), and only the stores are
  // guarded. The loads go out before the twiddle look-ups below, which then overlap them.
  const long long jc = min(j0 + lane2, a.nj - 2);
  c32 p[K], q[K];
#pragma unroll
  for (int k = 0; k < K; ++k) load2(min(i0 + row0 + RS * k, a.ni - 1), jc, p[k], q[k]);

  // Four-step twiddles of this thread's two columns j, j + 1 for its rows i = i0 + row0 + RS k, k < K:
  //   W^(p_k q) = W^(p_0 q) (W^(RS pi q))^k,   p_k = tw_p0 + tw_pi i,
  // so two table look-ups per column (base and step) and a short power tree replace a look-up per element
  // (the gathers of a look-up per element, not the arithmetic, were what this pass spent its time on).
  c32 tw_k[2][K];
  if constexpr (TWIDDLE) {
    auto root = [&](unsigned long long e) {
      e &= (unsigned long long)(a.tw_n - 1);
      return tfft::cmul(tw_hi[e >> a.tw_shift], tw_lo[e & ((1ull << a.tw_shift) - 1)]);
    };
    const unsigned long long mask = (unsigned long long)(a.tw_n - 1);
    const unsigned long long p_first = (unsigned long long)(a.tw_p0 + a.tw_pi * (i0 + row0)) & mask;
    const unsigned long long p_step = (unsigned long long)(RS * a.tw_pi) & mask;
    // Column j by per-lane look-ups (base and step: the table indices p q differ from lane to lane, so each is a
    // gather); column j + 1 from it: q grows by tw_qj, and W^(p_first tw_qj), W^(p_step tw_qj) depend on the row alone
    // -- the same table entry for every lane of a row, one cache line per wave-load instead of one per lane.
    const unsigned long long qcol =
        (unsigned long long)(a.tw_q0 + a.tw_qb1 * b1 + a.tw_qb2 * b2 + a.tw_qj * (j0 + lane2)) & mask;
    const unsigned long long dq = (unsigned long long)a.tw_qj & mask;
    c32 base[2], step[2];
    base[0] = root(p_first * qcol);
    step[0] = root(p_step * qcol);
    base[1] = tfft::cmul(base[0], root(p_first * dq));
    step[1] = tfft::cmul(step[0], root(p_step * dq));
#pragma unroll
    for (int c = 0; c < 2; ++c) {
      c32 pw[K];
      tfft::twiddle_powers<K>(step[c], pw);
      tw_k[c][0] = base[c];
#pragma unroll
      for (int k = 1; k < K; ++k) tw_k[c][k] = tfft::cmul(base[c], pw[k]);
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
    for (int k = 0; k < K; ++k) {
      const long long i = i0 + row0 + RS * k, j = j0 + lane2;
      if (i < a.ni && j < a.nj) store2(dst + i * a.out_j + j, twiddle(p[k], k, 0), twiddle(q[k], k, 1));
    }
  } else {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const int r = row0 + RS * k;
      tile[r][lane2] = twiddle(p[k], k, 0);
      tile[r][lane2 + 1] = twiddle(q[k], k, 1);
    }
    __syncthreads();
    // out[j][i]: two adjacent i per thread, TI / 2 threads per output row
    constexpr int TPRO = TI / 2, RSO = 256 / TPRO;
    static_assert(TJ / RSO == K, "eight 16-byte stores per thread");
    const int i2 = (threadIdx.x % TPRO) * 2, c0 = threadIdx.x / TPRO;
    c32 u[K], v[K];
#pragma unroll
    for (int k = 0; k < K; ++k) {
      u[k] = tile[i2][c0 + RSO * k];
      v[k] = tile[i2 + 1][c0 + RSO * k];
    }
#pragma unroll
    for (int k = 0; k < K; ++k) {
      const long long j = j0 + c0 + RSO * k, i = i0 + i2;
      // (the swap of the inverse was applied at load and is undone here by store2)
      if (j < a.nj && i < a.ni) store2(dst + j * a.out_j + i, u[k], v[k]);
    }
  }
}

// Copy of long rows without twiddle: one workgroup moves a run of kRunCols = 4096 points of ONE row (32 KiB
// contiguous on both sides), eight 16-byte loads per thread in flight, then eight stores.
constexpr int kRunCols = 4096;
__global__ __launch_bounds__(256) void dist_copy_runs_kernel(const float2* __restrict__ in, float2* __restrict__ out,
                                                             const tfft_dist_permute_args a) {
  const int runs = (int)((a.nj + kRunCols - 1) / kRunCols);
  long long t = tfft::xcd_chunk(blockIdx.x, gridDim.x, kXcdRunLog2);
  const long long j0 = (t % runs) * kRunCols;
  t /= runs;
  const long long i = t % a.ni;
  t /= a.ni;
  const long long b2 = t % a.nb2, b1 = t / a.nb2;
  const float2* src = in + b1 * a.in_b1 + b2 * a.in_b2 + i * a.in_i;
  float2* dst = out + b1 * a.out_b1 + b2 * a.out_b2 + i * a.out_j;
  f32x4_t v[8];
#pragma unroll
  for (int k = 0; k < 8; ++k)
    v[k] = __builtin_nontemporal_load(
        reinterpret_cast<const f32x4_t*>(src + min(j0 + 512 * k + 2 * (long long)threadIdx.x, a.nj - 2)));
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const long long j = j0 + 512 * k + 2 * (long long)threadIdx.x;
    // (a pure reordering has no direction: the inverse's swap at load and at store cancel)
    if (j < a.nj) __builtin_nontemporal_store(v[k], reinterpret_cast<f32x4_t*>(dst + j));
  }
}

template <int P, bool FWD>
__global__ __launch_bounds__(256) void dist_radix_kernel(const float2* in, float2* out, const tfft_dist_radix_args a,
                                                         const float2* __restrict__ tab) {
  // one workgroup per 256 pairs of points where the launch allows (a grid that walked the shard in strides measured
  // 6-8 % slower); the stride loop only serves shards beyond 2^31 workgroups
  const long long pairs_per_row = a.row_len / 2;
  const long long total = a.rows * pairs_per_row;
  for (long long idx = tfft::xcd_chunk(blockIdx.x, gridDim.x, kXcdRunLog2) * blockDim.x + threadIdx.x; idx < total;
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

template <bool TRANSPOSE, bool TWIDDLE, int TI, int TJ>
void launch_permute(const void* in, void* out, const tfft_dist_permute_args& a, const void* hi, const void* lo, int forward,
                    unsigned grid, hipStream_t s) {
  auto f = static_cast<const float2*>(in);
  auto o = static_cast<float2*>(out);
  auto h = static_cast<const float2*>(hi), l = static_cast<const float2*>(lo);
  if (forward)
    dist_permute_kernel<TRANSPOSE, TWIDDLE, true, TI, TJ><<<dim3(grid), dim3(256), 0, s>>>(f, o, a, h, l);
  else
    dist_permute_kernel<TRANSPOSE, TWIDDLE, false, TI, TJ><<<dim3(grid), dim3(256), 0, s>>>(f, o, a, h, l);
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
  // rows of at least half a run, no twiddle, no transposition: whole runs of one row per workgroup
  const bool runs = !a.transpose && !a.twiddle && a.nj >= kRunCols / 2;
  // tile shape of a transposing pass: 1 KiB segments on the side whose rows lie closer together
  const int shape = !a.transpose || a.twiddle ? 0 : a.in_i > a.out_j && a.ni >= 128 ? 1 : a.out_j > a.in_i && a.nj >= 128 ? 2 : 0;
  const long long ti = !a.transpose ? (runs ? 1 : kCopyRows) : shape == 1 ? 128 : shape == 2 ? 32 : 64;
  const long long tj = !a.transpose ? (runs ? kRunCols : kCopyCols) : 4096 / ti;
  const long long tiles = a.nb1 * a.nb2 * ((a.ni + ti - 1) / ti) * ((a.nj + tj - 1) / tj);
  if (tiles > 0x7fffffffLL) return tfft_internal_fail(TFFT_ERR_INVALID, "tfft_dist_permute: too many tiles for one launch");
  hipStream_t s = static_cast<hipStream_t>(stream);
  const unsigned grid = (unsigned)tiles;
  if (runs) {
    dist_copy_runs_kernel<<<dim3(grid), dim3(256), 0, s>>>(static_cast<const float2*>(d_in), static_cast<float2*>(d_out), a);
  } else if (!a.transpose) {
    if (a.twiddle) launch_permute<false, true, kCopyRows, kCopyCols>(d_in, d_out, a, d_tw_hi, d_tw_lo, forward, grid, s);
    else launch_permute<false, false, kCopyRows, kCopyCols>(d_in, d_out, a, d_tw_hi, d_tw_lo, forward, grid, s);
  } else if (shape == 1) {
    if (a.twiddle) launch_permute<true, true, 128, 32>(d_in, d_out, a, d_tw_hi, d_tw_lo, forward, grid, s);
    else launch_permute<true, false, 128, 32>(d_in, d_out, a, d_tw_hi, d_tw_lo, forward, grid, s);
  } else if (shape == 2) {
    if (a.twiddle) launch_permute<true, true, 32, 128>(d_in, d_out, a, d_tw_hi, d_tw_lo, forward, grid, s);
    else launch_permute<true, false, 32, 128>(d_in, d_out, a, d_tw_hi, d_tw_lo, forward, grid, s);
  } else {
    if (a.twiddle) launch_permute<true, true, 64, 64>(d_in, d_out, a, d_tw_hi, d_tw_lo, forward, grid, s);
    else launch_permute<true, false, 64, 64>(d_in, d_out, a, d_tw_hi, d_tw_lo, forward, grid, s);
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
  if (grid > 0x7fffffffLL) grid = 0x7fffffffLL;
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
