// Fused circular convolution through the frequency domain: one workgroup per signal runs
//   load -> forward FFT -> multiply by the filter spectrum -> inverse FFT -> store
// with the points in registers from the first load to the last store. Built from the
// device-level blocks of fft_block.hpp.
//
// Replaces the reference's thunder_conv4096_block_kernel, ..._realin and ..._realin_realout and
// its permute_filter_layout4096 (thunderfft/sample/conv1d/conv1d_fullfft_l4096_thunder.cu:
// 120-146, 148-292). Differences by design: fp32 arithmetic in registers for every storage type
// (the reference multiplies and accumulates in fp16), and consecutive workgroups share a filter
// (block -> (d, b) with b fastest), so that a filter spectrum is read from HBM once and then
// served by L2 to the B signals that use it.
#include <cmath>

#include "conv_plans.hpp"
#include "fft_block.hpp"
#include "fft_conv_mfma.hpp"
#include "../../include/thunderfft_c.h"

namespace tfft {

// 16 x 16 x 16, whole complex words through LDS (fewest barriers: the fused kernel is bound by
// butterflies and LDS exchanges, not by HBM).
using ConvPlan4096 = Plan<4096, 16, 16, 16, 16, 256, 4, 1, 5, 1, 4>;

// Filter spectra in register order: pair j of thread t (registers 2j, 2j+1) of filter d sits at
// raw element  d * (TPB*N/2) + j * WG + t, so one wave instruction reads consecutive raw elements.
template <class P, class HIO>
__global__ __launch_bounds__(P::WG) void conv_filter_layout_kernel(const typename HIO::raw* __restrict__ h,
                                                                  typename HIO::raw* __restrict__ hp, int D) {
  using raw = typename HIO::raw;
  const BlockFFT<P> f(nullptr, nullptr);
  const long long d = blockIdx.x;
  // natural order is addressed per point: view the raw buffers as arrays of single points
  using point = std::conditional_t<sizeof(raw) == 16, float2, unsigned>;
  const point* src = reinterpret_cast<const point*>(h) + d * P::N;
  point* dst = reinterpret_cast<point*>(hp) + d * (long long)(P::TPB * P::N);
#pragma unroll
  for (int e = 0; e < P::E; ++e) dst[((e >> 1) * P::WG + f.tid) * 2 + (e & 1)] = src[f.index_out(e)];
}

template <class P, class IOIN, class IOOUT, class HIO>
__global__ __launch_bounds__(P::WG, P::MINW) void fft_conv_kernel(const typename IOIN::raw* x,  // x and y may alias
                                                                  const typename HIO::raw* __restrict__ hp,
                                                                  typename IOOUT::raw* y,
                                                                  const float2* __restrict__ tw, int B, int D) {
  static_assert(P::TPB == 1, "one signal per workgroup");
  static_assert(BlockFFT<P>::kChainable, "first and last radix must agree");
  constexpr int N = P::N, E = P::E;
  __shared__ __attribute__((aligned(16))) char lds[P::LDS_BYTES];

  // blocks that share a filter are adjacent: block = d * B + b, signal = b * D + d
  const int d = blockIdx.x / B;
  const int b = blockIdx.x - d * B;
  const long long sig = (long long)b * D + d;

  const BlockFFT<P> f(tw, lds);
  typename IOIN::raw pre[P::NRAW];
  f.template load<IOIN>(x + sig * (N / 2), 1, pre);
  typename HIO::raw hraw[E / 2];
  const typename HIO::raw* hsrc = hp + (long long)d * (N / 2) + f.tid;
#pragma unroll
  for (int j = 0; j < E / 2; ++j) hraw[j] = hsrc[j * P::WG];

  c32 v[E];
  f.template unpack<IOIN>(pre, v);
  f.template transform<true>(v, false);
#pragma unroll
  for (int j = 0; j < E / 2; ++j) {
    c32 h0, h1;
    HIO::unpack(hraw[j], h0, h1);
    v[2 * j] = cmul(v[2 * j], h0);
    v[2 * j + 1] = cmul(v[2 * j + 1], h1);
  }
  f.template transform<false>(v, true);
  f.template store<IOOUT>(y + sig * (N / 2), 1, v);
}

// Real signals with a real filter, two signals (b, b+1) of one filter per workgroup: the pair
// travels as one complex signal x_b + i x_{b+1}. Convolution with real taps is linear over the
// reals, so the result is y_b + i y_{b+1}: half the butterflies, LDS traffic and filter reads per
// signal. (A complex filter would mix the two; this kernel is only reached when the caller states
// that the taps are real.)
template <class P, class HIO>
__global__ __launch_bounds__(P::WG, P::MINW) void fft_conv_real_pair_kernel(const unsigned* x,  // x and y may alias
                                                                            const typename HIO::raw* __restrict__ hp,
                                                                            unsigned* y,
                                                                            const float2* __restrict__ tw, int B, int D) {
  static_assert(P::TPB == 1 && P::NB1 == 1 && P::NBL == 1, "one signal pair per workgroup, one butterfly per thread");
  static_assert(BlockFFT<P>::kChainable, "first and last radix must agree");
  using RIO = StoreRealF16;
  constexpr int N = P::N, E = P::E, NRAW = P::NRAW, R1 = P::R1, RL = P::RL;
  __shared__ __attribute__((aligned(16))) char lds[P::LDS_BYTES];

  const int npair = (B + 1) >> 1;
  const int d = blockIdx.x / npair;
  const int b0 = 2 * (blockIdx.x - d * npair);
  const bool two = b0 + 1 < B;
  const long long s0 = (long long)b0 * D + d;
  const long long s1 = s0 + D;

  const BlockFFT<P> f(tw, lds);
  unsigned pre0[NRAW], pre1[NRAW];
  f.template load<RIO>(x + s0 * (N / 2), 1, pre0);
  if (two) {
    f.template load<RIO>(x + s1 * (N / 2), 1, pre1);
  } else {
#pragma unroll
    for (int i = 0; i < NRAW; ++i) pre1[i] = 0u;
  }
  typename HIO::raw hraw[E / 2];
  const typename HIO::raw* hsrc = hp + (long long)d * (N / 2) + f.tid;
#pragma unroll
  for (int j = 0; j < E / 2; ++j) hraw[j] = hsrc[j * P::WG];

  c32 v[E];
#pragma unroll
  for (int i = 0; i < NRAW; ++i) {
    c32 a0, a1, c0, c1;
    RIO::unpack(pre0[i], a0, a1);
    RIO::unpack(pre1[i], c0, c1);
    v[i] = make_float2(a0.x, c0.x);
    v[i + R1 / 2] = make_float2(a1.x, c1.x);
    swap_points(v[i], v[i + R1 / 2]);
  }
  f.template transform<true>(v, false);
#pragma unroll
  for (int j = 0; j < E / 2; ++j) {
    c32 h0, h1;
    HIO::unpack(hraw[j], h0, h1);
    v[2 * j] = cmul(v[2 * j], h0);
    v[2 * j + 1] = cmul(v[2 * j + 1], h1);
  }
  f.template transform<false>(v, true);
  unsigned* p0 = y + s0 * (N / 2) + f.st_base(0);
  unsigned* p1 = y + s1 * (N / 2) + f.st_base(0);
#pragma unroll
  for (int i = 0; i < NRAW; ++i) {
    c32 a = v[i], b = v[i + RL / 2];
    swap_points(a, b);
    RIO::store(p0 + i * (P::CL / 2), RIO::pack(make_float2(a.x, 0.f), make_float2(b.x, 0.f)));
    if (two) RIO::store(p1 + i * (P::CL / 2), RIO::pack(make_float2(a.y, 0.f), make_float2(b.y, 0.f)));
  }
}

template <class P, class HIO>
void launch_conv_real_pair(const void* x, const void* hp, void* y, const float2* tw, int B, int D, int, hipStream_t stream) {
  const unsigned npair = ((unsigned)B + 1u) / 2u;
  fft_conv_real_pair_kernel<P, HIO><<<dim3(npair * (unsigned)D), dim3(P::WG), 0, stream>>>(
      static_cast<const unsigned*>(x), static_cast<const typename HIO::raw*>(hp), static_cast<unsigned*>(y), tw, B, D);
}

template <class P, class IOIN, class IOOUT, class HIO>
void launch_conv(const void* x, const void* hp, void* y, const float2* tw, int B, int D, int, hipStream_t stream) {
  fft_conv_kernel<P, IOIN, IOOUT, HIO><<<dim3((unsigned)B * (unsigned)D), dim3(P::WG), 0, stream>>>(
      static_cast<const typename IOIN::raw*>(x), static_cast<const typename HIO::raw*>(hp),
      static_cast<typename IOOUT::raw*>(y), tw, B, D);
}

template <class P, class HIO>
void launch_layout(const void* h, void* hp, int D, hipStream_t stream) {
  conv_filter_layout_kernel<P, HIO><<<dim3((unsigned)D), dim3(P::WG), 0, stream>>>(
      static_cast<const typename HIO::raw*>(h), static_cast<typename HIO::raw*>(hp), D);
}

// Overlap-save tiles: tile t transforms the n points in[t*valid .. t*valid + n) (zero past the end
// of the input), multiplies by the filter spectrum hf (natural order, n points) and keeps the first
// `valid` points of the inverse transform as out[t*valid .. t*valid + valid). TPB tiles per
// workgroup. Replaces conv1d_direct_fft_kernel (thunderfft/sample/conv1d/main.cu:38-129), whose
// tiles likewise start `valid_tile` apart and overlap by filter_len - 1 points. Tiles start at
// arbitrary points, so global accesses are per point (4 or 8 bytes) rather than 16-byte pairs.
template <class T>
struct PointIO;
template <>
struct PointIO<float2> {
  __device__ static __forceinline__ c32 get(const float2* p) { return *p; }
  __device__ static __forceinline__ void put(float2* p, c32 v) { *p = v; }
};
template <>
struct PointIO<__half2> {
  __device__ static __forceinline__ c32 get(const __half2* p) { return __half22float2(*p); }
  __device__ static __forceinline__ void put(__half2* p, c32 v) { *p = __float22half2_rn(v); }
};

template <class P, class T>
__global__ __launch_bounds__(P::WG, P::MINW) void fft_conv_tiles_kernel(const T* __restrict__ in, const T* __restrict__ hf,
                                                                       T* __restrict__ out, const float2* __restrict__ tw,
                                                                       long long input_len, long long output_len, int valid,
                                                                       long long ntiles) {
  static_assert(BlockFFT<P>::kChainable, "first and last radix must agree");
  constexpr int E = P::E;
  __shared__ __attribute__((aligned(16))) char lds[P::LDS_BYTES];
  const BlockFFT<P> f(tw, lds);
  const long long tile0 = (long long)blockIdx.x * P::TPB;

  // Points beyond the input read point 0 and are zeroed afterwards: the loads are unconditional, so all of a thread's
  // E loads are in flight together (a load under a condition is a basic block of its own, closed by a wait).
  c32 v[E];
  auto in_range = [&](int e, long long& idx) {
    const long long t = tile0 + f.transform_in(e);
    idx = t * valid + f.index_in(e);
    return t < ntiles && idx < input_len;
  };
#pragma unroll
  for (int e = 0; e < E; ++e) {
    long long idx;
    const bool ok = in_range(e, idx);
    v[e] = PointIO<T>::get(in + (ok ? idx : 0));
  }
#pragma unroll
  for (int e = 0; e < E; ++e) {
    long long idx;
    if (!in_range(e, idx)) v[e] = make_float2(0.f, 0.f);
  }
  f.template transform<true>(v, false);
#pragma unroll
  for (int e = 0; e < E; ++e) v[e] = cmul(v[e], PointIO<T>::get(hf + f.index_out(e)));
  f.template transform<false>(v, true);
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const long long t = tile0 + f.transform_out(e);
    const int j = f.index_out(e);
    const long long idx = t * valid + j;
    if (t < ntiles && j < valid && idx < output_len) PointIO<T>::put(out + idx, v[e]);
  }
}

template <class P, class T>
void launch_conv_tiles(const void* in, const void* hf, void* out, const float2* tw, long long input_len,
                       long long output_len, int valid, hipStream_t stream) {
  const long long ntiles = (output_len + valid - 1) / valid;
  const long long blocks = (ntiles + P::TPB - 1) / P::TPB;
  fft_conv_tiles_kernel<P, T><<<dim3((unsigned)blocks), dim3(P::WG), 0, stream>>>(
      static_cast<const T*>(in), static_cast<const T*>(hf), static_cast<T*>(out), tw, input_len, output_len, valid, ntiles);
}

// 4 x 4 x 4 with 16 points per thread: 4 threads per tile, 16 tiles per 64-thread workgroup
using ConvTilePlan64 = Plan<64, 16, 4, 4, 4, 64, 5, 4, 5, 1>;

bool lookup_conv_tiles(int dtype, int n, ConvTilesEntry* out) {
  if (n != 64) return false;
  using P = ConvTilePlan64;
  out->n = P::N;
  out->threads = P::WG;
  out->tiles_per_block = P::TPB;
  if (dtype == TFFT_F32) { out->launch = &launch_conv_tiles<P, float2>; return true; }
  if (dtype == TFFT_F16) { out->launch = &launch_conv_tiles<P, __half2>; return true; }
  return false;
}

template <class P, class IOIN, class IOOUT, class HIO>
ConvEntry make_conv_entry() {
  ConvEntry e;
  e.launch = &launch_conv<P, IOIN, IOOUT, HIO>;
  e.layout = &launch_layout<P, HIO>;
  e.layout_id = kConvLayoutValu;
  e.n = P::N;
  e.threads = P::WG;
  e.points_per_thread = P::E;
  e.lds_bytes = P::LDS_BYTES;
  return e;
}

template <class P>
bool lookup_conv_plan(int dtype, int kind, ConvEntry* out) {
  if (dtype == TFFT_F32 && kind == kConvComplex) {
    *out = make_conv_entry<P, StoreF32, StoreF32, StoreF32>();
    return true;
  }
  if (dtype == TFFT_F16) {
    switch (kind) {
      case kConvComplex: *out = make_conv_entry<P, StoreF16, StoreF16, StoreF16>(); return true;
      case kConvRealIn: *out = make_conv_entry<P, StoreRealF16, StoreF16, StoreF16>(); return true;
      case kConvRealInOut:
        *out = make_conv_entry<P, StoreRealF16, StoreRealF16, StoreF16>();
        out->launch = &launch_conv_real_pair<P, StoreF16>;
        return true;
      default: return false;
    }
  }
  return false;
}

// per_cu: workgroups per CU of the resident form (0: one workgroup per work item, the default: at ~190 VGPRs the
// resident kernel fits two workgroups per CU and measured 10-15 % slower than one-shot workgroups).
template <int KIND>
void launch_conv_mfma(const void* x, const void* hp, void* y, const float2* tw, int B, int D, int per_cu,
                      hipStream_t stream) {
  const unsigned nb = (KIND == kConvMfmaRealPair) ? ((unsigned)B + 1u) / 2u : (unsigned)B;
  const unsigned items = nb * (unsigned)D;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  }
  if (per_cu > 0 && items > (unsigned)(per_cu * cus)) {
    fft_conv_mfma4096_kernel<KIND, true><<<dim3((unsigned)(per_cu * cus)), dim3(256), 0, stream>>>(
        x, static_cast<const u32x4_t*>(hp), y, tw, B, D);
  } else {
    fft_conv_mfma4096_kernel<KIND, false><<<dim3(items), dim3(256), 0, stream>>>(x, static_cast<const u32x4_t*>(hp), y, tw,
                                                                                 B, D);
  }
}

// How many work items of one filter a workgroup of the staged kernel takes (8, 4 or 2), or 0 for the
// one-workgroup-per-item kernel. A run costs its items plus about a quarter of an item for the 48 KiB of tables and
// filter it reads before its first signal, so long runs win while they still fill the `slots` resident workgroups
// (three per CU) about evenly: the estimate is (workgroups / slots, rounded up to whole rounds below four of them)
// x (items per run + 1/4), against 1.6 per item for the kernel that overlaps nothing. It names the fastest run length,
// or one within 1 % of it, on all thirteen shapes it was checked on (B = 2 ... 128 real pairs, D = 24 ... 768).
inline int conv_run_length(unsigned nb, unsigned D, unsigned slots) {
  auto rounds = [&](unsigned long long groups) {
    const double r = (double)groups / (double)slots;
    return r >= 4.0 ? r : std::ceil(r);
  };
  int run = 0;
  double best = rounds((unsigned long long)nb * D) * 1.6;
  if (nb < 2) return 0;
  for (const unsigned r : {2u, 4u, 8u}) {
    const unsigned long long groups = (unsigned long long)((nb + r - 1) / r) * D;
    const double cost = rounds(groups) * ((double)(r < nb ? r : nb) + 0.25);
    if (cost <= best) {
      best = cost;
      run = (int)r;
    }
  }
  return run;
}

// The default form for fp16 storage: one workgroup per run of RUN work items of one filter, the next item's signal
// travelling into LDS under the current item's passes (fft_conv_mfma4096_dma_kernel), RUN from conv_run_length;
// calls that ask for the resident grid (per_cu > 0) take the one-workgroup-per-item kernel.
template <int KIND>
void launch_conv_mfma_default(const void* x, const void* hp, void* y, const float2* tw, int B, int D, int per_cu,
                              hipStream_t stream) {
  const unsigned nb = (KIND == kConvMfmaRealPair) ? ((unsigned)B + 1u) / 2u : (unsigned)B;
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
  }
  const int run = per_cu > 0 ? 0 : conv_run_length(nb, (unsigned)D, 3u * (unsigned)cus);
  auto groups = [&](unsigned r) { return (unsigned)(((nb + r - 1) / r) * (unsigned)D); };
  const u32x4_t* h = static_cast<const u32x4_t*>(hp);
  switch (run) {
    case 8: fft_conv_mfma4096_dma_kernel<KIND, 8><<<dim3(groups(8)), dim3(256), 0, stream>>>(x, h, y, tw, B, D); break;
    case 4: fft_conv_mfma4096_dma_kernel<KIND, 4><<<dim3(groups(4)), dim3(256), 0, stream>>>(x, h, y, tw, B, D); break;
    case 2: fft_conv_mfma4096_dma_kernel<KIND, 2><<<dim3(groups(2)), dim3(256), 0, stream>>>(x, h, y, tw, B, D); break;
    default: launch_conv_mfma<KIND>(x, hp, y, tw, B, D, per_cu, stream); break;
  }
}

// The staged form whatever the problem size (tests: every RUN on small shapes).
template <int KIND, int RUN, bool PRIV>
void launch_conv_mfma_dma(const void* x, const void* hp, void* y, const float2* tw, int B, int D, int per_cu,
                          hipStream_t stream) {
  (void)per_cu;
  const unsigned nb = (KIND == kConvMfmaRealPair) ? ((unsigned)B + 1u) / 2u : (unsigned)B;
  fft_conv_mfma4096_dma_kernel<KIND, RUN, PRIV><<<dim3(((nb + RUN - 1) / RUN) * (unsigned)D), dim3(256), 0, stream>>>(
      x, static_cast<const u32x4_t*>(hp), y, tw, B, D);
}

template <int RUN, bool PRIV = false>
bool conv_mfma_dma_entry(int kind, ConvEntry* out) {
  switch (kind) {
    case kConvComplex: out->launch = &launch_conv_mfma_dma<kConvMfmaComplex, RUN, PRIV>; return true;
    case kConvRealIn: out->launch = &launch_conv_mfma_dma<kConvMfmaRealIn, RUN, PRIV>; return true;
    case kConvRealInOut: out->launch = &launch_conv_mfma_dma<kConvMfmaRealPair, RUN, PRIV>; return true;
    default: return false;
  }
}

void launch_layout_mfma(const void* h, void* hp, int D, hipStream_t stream) {
  conv_mfma4096_filter_layout_kernel<0><<<dim3((unsigned)D), dim3(256), 0, stream>>>(static_cast<const unsigned*>(h),
                                                                                   static_cast<u32x4_t*>(hp), D);
}

bool lookup_conv(int dtype, int n, int kind, int plan, ConvEntry* out) {
  if (n != 4096 || (plan != 0 && plan != 3 && !(dtype == TFFT_F16 && plan >= 4 && plan <= 7))) return false;
  out->layout_id = kConvLayoutValu;
  if (dtype == TFFT_F16 && plan != 3) {
    // fp16 storage: both transforms on the matrix cores (fft_conv_mfma.hpp)
    out->n = 4096;
    out->threads = 256;
    out->points_per_thread = 16;
    out->lds_bytes = 4 * (Mfma4096::OUT_IMG + Mfma4096::IMG);
    out->layout = &launch_layout_mfma;
    out->layout_id = kConvLayoutMfma;
    out->spectra = [](const void* taps, void* hp, const float2* tw, int D, hipStream_t stream) {
      launch_conv_spectra_f16(taps, hp, tw, D, stream);
    };
    // plan 0: the staged (LDS-DMA) form with RUN chosen from the problem size; 4: one workgroup per item (the
    // resident grid of grid_blocks_per_cu belongs to it); 5, 6: the staged form with RUN = 2 / 8 whatever the size;
    if (plan == 5) return conv_mfma_dma_entry<2>(kind, out);
    if (plan == 6) return conv_mfma_dma_entry<8>(kind, out);
    // 7: runs of 4 with wave-private staging and alternating images (two barriers fewer per item; measured level
    // with the shared staging area, kept for A/B)
    if (plan == 7) return conv_mfma_dma_entry<4, true>(kind, out);
    if (plan == 0) {
      switch (kind) {
        case kConvComplex: out->launch = &launch_conv_mfma_default<kConvMfmaComplex>; return true;
        case kConvRealIn: out->launch = &launch_conv_mfma_default<kConvMfmaRealIn>; return true;
        case kConvRealInOut: out->launch = &launch_conv_mfma_default<kConvMfmaRealPair>; return true;
        default: return false;
      }
    }
    switch (kind) {
      case kConvComplex: out->launch = &launch_conv_mfma<kConvMfmaComplex>; return true;
      case kConvRealIn: out->launch = &launch_conv_mfma<kConvMfmaRealIn>; return true;
      case kConvRealInOut: out->launch = &launch_conv_mfma<kConvMfmaRealPair>; return true;
      default: return false;
    }
  }
  return lookup_conv_plan<ConvPlan4096>(dtype, kind, out);
}

}  // namespace tfft
