// Device-level building blocks: one workgroup, TPB whole transforms, the pieces of the batched
// kernel as separate calls so that a user kernel can put work between them.
//
// Counterpart of the reference's composable device pipeline
//   ThunderFFT_gmem2smem -> ThunderFFT_smem2reg -> ThunderFFT_kernel_reg -> ThunderFFT_reg2smem ->
//   ThunderFFT_smem2gmem            (thunderfft/include/thunderfft/thunderfft.cuh:55-130)
// on wave64 hardware. The reference stages rows through shared memory into MMA fragments; here
// a thread's E points go from global memory straight into VGPRs in the order the first butterfly
// pass wants them, and leave from the order of the last pass:
//
//   BlockFFT<P> f(tw, lds);             // per-thread index math + twiddle bases (tw: W_N^k, k < N)
//   f.load<IO>(src, valid, raw);        // global -> registers        (gmem2smem + smem2reg)
//   f.unpack<IO>(raw, v);               // storage type -> fp32 complex in first-pass order
//   f.transform<forward>(v, lead);      // the N-point transform      (ThunderFFT_kernel_reg)
//   f.store<IO>(dst, valid, v);         // registers -> global        (reg2smem + smem2gmem)
//
// Register e of a thread holds time index  index_in(e)  of transform  transform_in(e)  before
// transform(), and frequency index  index_out(e)  of transform  transform_out(e)  after it. For
// plans whose first and last radix agree (R1 == RL) the two maps are the same function, so the
// output registers of one transform are valid input registers of the next: forward transform,
// pointwise work in the frequency domain, inverse transform, with no data movement in between
// (what the reference's conv samples do with reverse_data()/swap_thread_data() between their two
// ThunderFFT_kernel_reg calls, sample/conv1d/conv1d_fullfft_l4096_thunder.cu:178-189).
#pragma once
#include "fft_lds_kernel.hpp"

namespace tfft {

// Real-valued 16-bit storage: two adjacent real points per 32-bit word; imaginary parts are zero
// on load and dropped on store (the reference's *_realio loaders,
// sample/conv1d/conv1d_fullfft_l4096_thunder.cu:75-118).
struct StoreRealF16 {
  using raw = unsigned;
  static constexpr int kBytesPerPoint = 2;
  __device__ static __forceinline__ raw load(const raw* p) { return __builtin_nontemporal_load(p); }
  __device__ static __forceinline__ void store(raw* p, raw r) { __builtin_nontemporal_store(r, p); }
  __device__ static __forceinline__ raw zero() { return 0u; }
  __device__ static __forceinline__ void unpack(raw r, c32& a, c32& b) {
    const __half2 h = *reinterpret_cast<const __half2*>(&r);
    const float2 f = __half22float2(h);
    a = make_float2(f.x, 0.f);
    b = make_float2(f.y, 0.f);
  }
  __device__ static __forceinline__ raw pack(c32 a, c32 b) {
    const __half2 h = __float22half2_rn(make_float2(a.x, b.x));
    return *reinterpret_cast<const unsigned*>(&h);
  }
};

// Pair exchange between lanes l and l+32 on unpacked points: afterwards the lower half-wave holds
// (a_low, a_high) and the upper (b_low, b_high) of the original (a, b) pairs.
__device__ __forceinline__ void swap_points(c32& a, c32& b) {
  auto sx = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
  auto sy = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.y), __float_as_uint(b.y), false, false);
  a.x = __uint_as_float(sx[0]);
  b.x = __uint_as_float(sx[1]);
  a.y = __uint_as_float(sy[0]);
  b.y = __uint_as_float(sy[1]);
}
__device__ __forceinline__ void swap_reals(c32& a, c32& b) {
  auto sx = __builtin_amdgcn_permlane32_swap(__float_as_uint(a.x), __float_as_uint(b.x), false, false);
  a.x = __uint_as_float(sx[0]);
  b.x = __uint_as_float(sx[1]);
}

template <class IO>
struct io_is_real : std::false_type {};
template <>
struct io_is_real<StoreRealF16> : std::true_type {};

// (cos, -sin)(2 pi k / n) evaluated on the spot: k / n is exact for the power-of-two lengths here, so sincospif
// keeps its full accuracy. Stands in for a table entry when a kernel has no table at hand (the reference
// evaluates every twiddle this way, with fast-math intrinsics: rotate(), detail/util_arith.h:85-91).
__device__ __forceinline__ c32 unit_root(int k, int n) {
  float s, c;
  sincospif(-2.0f * (float)k / (float)n, &s, &c);
  return make_float2(c, s);
}

// TW_OPTIONAL: a null table pointer is allowed and means "evaluate the twiddle bases in the kernel".
template <class P, bool TW_OPTIONAL = false>
struct BlockFFT {
  static constexpr int N = P::N, E = P::E, R1 = P::R1, R2 = P::R2, R3 = P::R3, RL = P::RL, WG = P::WG;
  static constexpr int C1 = P::C1, C2 = P::C2, CL = P::CL, NB1 = P::NB1, NB2 = P::NB2, NBL = P::NBL;
  static constexpr int NRAW = P::NRAW, TPB = P::TPB;
  static constexpr int NG1 = (NB1 == 1) ? 1 : NB1 / 2, LPG1 = NRAW / NG1;
  static constexpr int NGL = (NBL == 1) ? 1 : NBL / 2, SPGL = NRAW / NGL;
  static constexpr bool kChainable = (R1 == RL);  // index_out == index_in: transforms can be chained in registers

  // State is the thread id, the table and the workspace only: every index map below is a pure
  // function of the thread id (shifts and masks, all sizes being powers of two), recomputed where
  // it is used and folded by common-subexpression elimination. (Member arrays of butterfly indices
  // defeat scalar replacement once several members are used in one kernel and end up in scratch.)
  int tid, hi;
  const float2* tw;
  float2* lds2;
  float* lds1;

  // tw: the table W_N^k = (cos, -sin)(2 pi k / N), k < N, that tfft_initialize builds; lds: a
  // workgroup-shared buffer of P::LDS_BYTES bytes, 16-byte aligned.
  __device__ __forceinline__ BlockFFT(const float2* __restrict__ tw_, char* lds) {
    tid = threadIdx.x;
    hi = (tid >> 5) & 1;
    tw = tw_;
    lds2 = reinterpret_cast<float2*>(lds);
    lds1 = reinterpret_cast<float*>(lds);
  }

  // table entry W_N^k
  __device__ __forceinline__ c32 twiddle(int k) const {
    if constexpr (TW_OPTIONAL) {
      if (tw == nullptr) return unit_root(k, N);
    }
    return tw[k];
  }

  // ---- butterflies of a thread. First and last pass: butterfly b handles column g % C of
  // transform g / C, lanes l and l+32 (one butterfly per thread) or registers 2i, 2i+1 (several)
  // on adjacent columns; middle pass: butterfly (k1, n3).
  template <int NB>
  __device__ __forceinline__ int edge_g(int b) const {
    if constexpr (NB == 1) return pair_swizzle(tid);
    return 2 * (tid + WG * (b >> 1)) + (b & 1);
  }
  __device__ __forceinline__ int b1_tr(int b) const { return edge_g<NB1>(b) / C1; }
  __device__ __forceinline__ int b1_c(int b) const { return edge_g<NB1>(b) % C1; }
  __device__ __forceinline__ int bl_tr(int b) const { return edge_g<NBL>(b) / CL; }
  __device__ __forceinline__ int bl_q(int b) const { return edge_g<NBL>(b) % CL; }
  __device__ __forceinline__ int b2_tr(int b) const { return (tid + WG * b) / C2; }
  __device__ __forceinline__ int b2_k1(int b) const { return ((tid + WG * b) % C2) / R3; }
  __device__ __forceinline__ int b2_n3(int b) const { return ((tid + WG * b) % C2) % R3; }
  // raw-element offsets (two points each) of a group's first load / store, and its transform
  __device__ __forceinline__ int ld_tr(int g) const { return b1_tr(NB1 == 1 ? 0 : 2 * g); }
  __device__ __forceinline__ int ld_base(int g) const {
    if constexpr (NB1 == 1) return (b1_tr(0) * N + hi * (R1 / 2) * C1 + (b1_c(0) & ~1)) >> 1;
    return (b1_tr(2 * g) * N + b1_c(2 * g)) >> 1;
  }
  __device__ __forceinline__ int st_tr(int g) const { return bl_tr(NBL == 1 ? 0 : 2 * g); }
  __device__ __forceinline__ int st_base(int g) const {
    if constexpr (NBL == 1) return (bl_tr(0) * N + hi * (RL / 2) * CL + (bl_q(0) & ~1)) >> 1;
    return (bl_tr(2 * g) * N + bl_q(2 * g)) >> 1;
  }

  // ---- which point a register holds
  __device__ __forceinline__ int index_in(int e) const { return b1_c(e / R1) + (e % R1) * C1; }
  __device__ __forceinline__ int transform_in(int e) const { return b1_tr(e / R1); }
  __device__ __forceinline__ int index_out(int e) const { return bl_q(e / RL) + (e % RL) * CL; }
  __device__ __forceinline__ int transform_out(int e) const { return bl_tr(e / RL); }

  // ---- global -> registers. `src` addresses this workgroup's TPB transforms ([TPB][N] points,
  // two points per raw element); transforms numbered `valid` and above read as zero.
  template <class IO>
  __device__ __forceinline__ void load(const typename IO::raw* src, int valid, typename IO::raw (&pre)[NRAW]) const {
#pragma unroll
    for (int g = 0; g < NG1; ++g) {
      const bool ok = (TPB == 1) || (ld_tr(g) < valid);
      const typename IO::raw* p = src + ld_base(g);
#pragma unroll
      for (int i = 0; i < LPG1; ++i) pre[g * LPG1 + i] = ok ? IO::load(p + i * (C1 / 2)) : IO::zero();
    }
  }

  template <class IO>
  __device__ __forceinline__ void unpack(const typename IO::raw (&pre)[NRAW], c32 (&v)[E]) const {
    if constexpr (NB1 == 1) {
#pragma unroll
      for (int i = 0; i < NRAW; ++i) {
        IO::unpack(pre[i], v[i], v[i + R1 / 2]);
        if constexpr (io_is_real<IO>::value)
          swap_reals(v[i], v[i + R1 / 2]);
        else
          swap_points(v[i], v[i + R1 / 2]);
      }
    } else {
#pragma unroll
      for (int pi = 0; pi < NB1 / 2; ++pi)
#pragma unroll
        for (int n1 = 0; n1 < R1; ++n1)
          IO::unpack(pre[pi * R1 + n1], v[(2 * pi) * R1 + n1], v[(2 * pi + 1) * R1 + n1]);
    }
  }

  // ---- registers -> global, the mirror image of load + unpack
  template <class IO>
  __device__ __forceinline__ void store(typename IO::raw* dst, int valid, const c32 (&vin)[E]) const {
    // a private copy: pairing registers b*RL + i of two butterflies straight out of the caller's
    // array leaves part of that array in scratch memory (scalar replacement gives up on it)
    c32 v[E];
#pragma unroll
    for (int e = 0; e < E; ++e) v[e] = make_float2(vin[e].x, vin[e].y);
#pragma unroll
    for (int g = 0; g < NGL; ++g) {
      const bool ok = (TPB == 1) || (st_tr(g) < valid);
      typename IO::raw* p = dst + st_base(g);
#pragma unroll
      for (int i = 0; i < SPGL; ++i) {
        typename IO::raw r;
        if constexpr (NBL == 1) {
          c32 a = v[i], b = v[i + RL / 2];
          if constexpr (io_is_real<IO>::value)
            swap_reals(a, b);
          else
            swap_points(a, b);
          r = IO::pack(a, b);
        } else {
          r = IO::pack(v[(2 * g) * RL + i], v[(2 * g + 1) * RL + i]);
        }
        if (ok) IO::store(p + i * (CL / 2), r);
      }
    }
  }

  // ---- LDS slots of the two exchanges (same maps as fft_lds_kernel)
  __device__ __forceinline__ int ex1_wslot(int b, int k1) const {
    const int base = b1_tr(b) * N + b1_c(b);
    return walk_slot<P::PS1, P::PW1, C1, R1, false>(base, pad_idx<P::PS1, P::PW1>(base), k1);
  }
  __device__ __forceinline__ int ex1_rslot(int b, int n2) const {
    if constexpr (P::PASSES == 3) {
      const int base = b2_tr(b) * N + b2_k1(b) * C1 + b2_n3(b);
      return walk_slot<P::PS1, P::PW1, R3, R2, false>(base, pad_idx<P::PS1, P::PW1>(base), n2);
    } else {
      const int base = bl_tr(b) * N + bl_q(b) * C1;
      return walk_slot<P::PS1, P::PW1, 1, R2, true>(base, pad_idx<P::PS1, P::PW1>(base), n2);
    }
  }
  __device__ __forceinline__ int ex2_wslot(int b, int k2) const {
    const int base = b2_tr(b) * N + b2_k1(b) * R3 + b2_n3(b);
    return walk_slot<P::PS2, P::PW2, R1 * R3, R2, false>(base, pad_idx<P::PS2, P::PW2>(base), k2);
  }
  __device__ __forceinline__ int ex2_rslot(int b, int n3) const {
    const int k1 = bl_q(b) % R1, k2 = bl_q(b) / R1;
    const int base = bl_tr(b) * N + k2 * (R1 * R3) + k1 * R3;
    return walk_slot<P::PS2, P::PW2, 1, R3, true>(base, pad_idx<P::PS2, P::PW2>(base), n3);
  }

  template <int WHICH>
  __device__ __forceinline__ int wslot(int e) const {
    if constexpr (WHICH == 1) return ex1_wslot(e / R1, e % R1);
    return ex2_wslot(e / R2, e % R2);
  }
  template <int WHICH>
  __device__ __forceinline__ int rslot(int e) const {
    if constexpr (WHICH == 1) return ex1_rslot(e / R2, e % R2);
    return ex2_rslot(e / R3, e % R3);
  }

  // One exchange through LDS: writes in the finished pass's order, reads in the next pass's.
  template <int WHICH>
  __device__ __forceinline__ void exchange(c32 (&v)[E], bool lead) const {
    if (lead) __syncthreads();
    if constexpr (!P::SPLIT) {
#pragma unroll
      for (int e = 0; e < E; ++e) lds2[wslot<WHICH>(e)] = make_float2(v[e].x, v[e].y);
      __syncthreads();
#pragma unroll
      for (int e = 0; e < E; ++e) {
        const float2 t = lds2[rslot<WHICH>(e)];
        v[e].x = t.x;
        v[e].y = t.y;
      }
    } else {
      float re[E];
#pragma unroll
      for (int e = 0; e < E; ++e) lds1[wslot<WHICH>(e)] = v[e].x;
      __syncthreads();
#pragma unroll
      for (int e = 0; e < E; ++e) re[e] = lds1[rslot<WHICH>(e)];
      __syncthreads();
#pragma unroll
      for (int e = 0; e < E; ++e) lds1[wslot<WHICH>(e)] = v[e].y;
      __syncthreads();
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] = make_float2(re[e], lds1[rslot<WHICH>(e)]);
    }
  }

  // radix-R butterflies on v[b*R .. b*R+R-1], then the inter-pass twiddles of set TW (0: none,
  // 1: after the first pass, 2: after the middle pass)
  template <int R, int TW>
  __device__ __forceinline__ void butterflies(c32 (&v)[E], const c32* w) const {
#pragma unroll
    for (int b = 0; b < E / R; ++b) {
      c32 x[R];
#pragma unroll
      for (int j = 0; j < R; ++j) x[j] = v[b * R + j];
      dft<R>(x);
      if constexpr (TW != 0) apply_twiddles<R>(x, fresh<false>(w[b]));
#pragma unroll
      for (int j = 0; j < R; ++j) v[b * R + j] = x[j];
    }
  }

  // The N-point transform on registers, unnormalised in both directions. In: first-pass order
  // (index_in); out: last-pass order (index_out). `lead`: the LDS buffer may still be read by a
  // previous transform() or by user code, so start with a barrier. The inverse runs the forward
  // code on data with real and imaginary parts swapped. EXCHANGE = false (a measurement build)
  // leaves the LDS exchanges out: the butterflies alone, on the wrong data.
  template <bool FWD, bool EXCHANGE = true>
  __device__ __forceinline__ void transform(c32 (&v)[E], bool lead) const {
    if constexpr (!FWD) {
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] = make_float2(v[e].y, v[e].x);
    }
    // twiddle bases, one per butterfly: W_N^c after the first pass, W_{R2 R3}^{n3} after the middle one
    c32 w1[NB1], w2[NB2];
#pragma unroll
    for (int b = 0; b < NB1; ++b) w1[b] = twiddle(b1_c(b));
    if constexpr (P::PASSES == 3) {
#pragma unroll
      for (int b = 0; b < NB2; ++b) w2[b] = twiddle(b2_n3(b) * R1);
    }
    butterflies<R1, 1>(v, w1);
    if constexpr (EXCHANGE) exchange<1>(v, lead);
    if constexpr (P::PASSES == 3) {
      butterflies<R2, 2>(v, w2);
      if constexpr (EXCHANGE) exchange<2>(v, true);
    }
    butterflies<RL, 0>(v, nullptr);
    if constexpr (!FWD) {
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] = make_float2(v[e].y, v[e].x);
    }
  }
};

}  // namespace tfft
