// Batched 1D complex FFT, one LDS-resident transform (or several small ones) per workgroup.
//
// Replaces the reference's device pipeline gmem2smem -> smem2reg -> ThunderFFT_kernel_reg ->
// reg2smem -> smem2gmem (thunderfft/include/thunderfft/detail/kernel/global_kernel.cuh:16-51
// and the per-N files detail/kernel/shared_kernel_n*.cuh) with a wave64 design:
//
//   N = R1*R2 or R1*R2*R3, every thread keeps E complex points in VGPRs and runs E/R
//   radix-R butterflies per pass; passes exchange data through one padded LDS buffer.
//
//   first pass : rows n1 of column c sit N/R1 apart in memory, so one wave-instruction reads
//                whole 512-byte row segments. Each lane loads 16 bytes (two adjacent columns);
//                lanes l and l+32 own adjacent columns and trade halves with
//                v_permlane32_swap, which leaves every lane with all R1 rows of one column.
//   last pass  : the mirror image: outputs k of butterfly q go to X[q + (N/RL) k], lanes l and
//                l+32 swap halves and store 16 bytes each.
//   twiddles   : one table entry w = W^c per butterfly (loop invariant, kept in VGPRs), its
//                powers by squaring/multiplying; the table is built in fp64 on the host.
//   batch loop : a workgroup walks chunks blockIdx.x, blockIdx.x + gridDim.x, ... and issues
//                the global loads of its next chunk before it computes the current one, so
//                HBM requests stay in flight under the butterflies and LDS exchanges.
//
// Inverse transforms run the same forward code on data with re/im swapped at load and store
// (IDFT(x) = swap(DFT(swap(x)))), unnormalised like the reference
// (sample/conv1d/conv1d_fullfft_l4096_thunder.cu:354-356 pre-scales by 1/N itself).
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include "butterfly.hpp"

namespace tfft {

// ---------------------------------------------------------------------------------------
// Storage traits: how two adjacent complex points travel between HBM and registers.
// ---------------------------------------------------------------------------------------
// Every point is read once and written once, so global accesses are nontemporal (streaming):
// they do not displace the twiddle tables from L2 and measure ~5 % faster on a plain copy.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

struct StoreF32 {
  using raw = float4;  // (re0, im0, re1, im1)
  static constexpr int kBytesPerPoint = 8;
  __device__ static __forceinline__ raw load(const raw* p) {
    const f32x4_t v = __builtin_nontemporal_load(reinterpret_cast<const f32x4_t*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
  }
  __device__ static __forceinline__ void store(raw* p, raw r) {
    const f32x4_t v = {r.x, r.y, r.z, r.w};
    __builtin_nontemporal_store(v, reinterpret_cast<f32x4_t*>(p));
  }
  __device__ static __forceinline__ raw zero() { return make_float4(0.f, 0.f, 0.f, 0.f); }
  __device__ static __forceinline__ void unpack(raw r, c32& a, c32& b) {
    a = make_float2(r.x, r.y);
    b = make_float2(r.z, r.w);
  }
  __device__ static __forceinline__ raw pack(c32 a, c32 b) { return make_float4(a.x, a.y, b.x, b.y); }
  // Lanes 32-63 of the first point trade places with lanes 0-31 of the second.
  __device__ static __forceinline__ void swap_halves(raw& r) {
    auto sx = __builtin_amdgcn_permlane32_swap(__float_as_uint(r.x), __float_as_uint(r.z), false, false);
    auto sy = __builtin_amdgcn_permlane32_swap(__float_as_uint(r.y), __float_as_uint(r.w), false, false);
    r.x = __uint_as_float(sx[0]);
    r.z = __uint_as_float(sx[1]);
    r.y = __uint_as_float(sy[0]);
    r.w = __uint_as_float(sy[1]);
  }
};

struct StoreF16 {
  using raw = uint2;  // two __half2 (re, im)
  static constexpr int kBytesPerPoint = 4;
  __device__ static __forceinline__ raw load(const raw* p) {
    const u32x2_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(p));
    return make_uint2(v.x, v.y);
  }
  __device__ static __forceinline__ void store(raw* p, raw r) {
    const u32x2_t v = {r.x, r.y};
    __builtin_nontemporal_store(v, reinterpret_cast<u32x2_t*>(p));
  }
  __device__ static __forceinline__ raw zero() { return make_uint2(0u, 0u); }
  __device__ static __forceinline__ c32 cvt(unsigned u) {
    const __half2 h = *reinterpret_cast<const __half2*>(&u);
    return __half22float2(h);
  }
  __device__ static __forceinline__ unsigned cvt(c32 v) {
    const __half2 h = __float22half2_rn(v);
    return *reinterpret_cast<const unsigned*>(&h);
  }
  __device__ static __forceinline__ void unpack(raw r, c32& a, c32& b) {
    a = cvt(r.x);
    b = cvt(r.y);
  }
  __device__ static __forceinline__ raw pack(c32 a, c32 b) { return make_uint2(cvt(a), cvt(b)); }
  __device__ static __forceinline__ void swap_halves(raw& r) {
    auto s = __builtin_amdgcn_permlane32_swap(r.x, r.y, false, false);
    r.x = s[0];
    r.y = s[1];
  }
};

struct StoreBF16 {
  using raw = uint2;  // two (bf16 re, bf16 im) pairs
  static constexpr int kBytesPerPoint = 4;
  __device__ static __forceinline__ raw load(const raw* p) {
    const u32x2_t v = __builtin_nontemporal_load(reinterpret_cast<const u32x2_t*>(p));
    return make_uint2(v.x, v.y);
  }
  __device__ static __forceinline__ void store(raw* p, raw r) {
    const u32x2_t v = {r.x, r.y};
    __builtin_nontemporal_store(v, reinterpret_cast<u32x2_t*>(p));
  }
  __device__ static __forceinline__ raw zero() { return make_uint2(0u, 0u); }
  __device__ static __forceinline__ c32 cvt(unsigned u) {
    return make_float2(__uint_as_float(u << 16), __uint_as_float(u & 0xffff0000u));
  }
  __device__ static __forceinline__ unsigned rne(float f) {  // round to nearest even, NaN kept
    const unsigned u = __float_as_uint(f);
    if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
    return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
  }
  __device__ static __forceinline__ unsigned cvt(c32 v) { return rne(v.x) | (rne(v.y) << 16); }
  __device__ static __forceinline__ void unpack(raw r, c32& a, c32& b) {
    a = cvt(r.x);
    b = cvt(r.y);
  }
  __device__ static __forceinline__ raw pack(c32 a, c32 b) { return make_uint2(cvt(a), cvt(b)); }
  __device__ static __forceinline__ void swap_halves(raw& r) {
    auto s = __builtin_amdgcn_permlane32_swap(r.x, r.y, false, false);
    r.x = s[0];
    r.y = s[1];
  }
};

// ---------------------------------------------------------------------------------------
// Plan constants shared by host and device.
// ---------------------------------------------------------------------------------------
template <int N_, int E_, int R1_, int R2_, int R3_, int WG_, int PS1_, int PW1_, int PS2_, int PW2_, int MINW_ = 1, bool HOIST_ = false>
struct Plan {
  static constexpr int N = N_, E = E_, R1 = R1_, R2 = R2_, R3 = R3_, WG = WG_;
  static constexpr int PS1 = PS1_, PW1 = PW1_, PS2 = PS2_, PW2 = PW2_;
  static constexpr int MINW = MINW_;  // waves per SIMD the register allocator must leave room for
  // true: the twiddle powers are loop invariants the compiler may keep in VGPRs across chunks;
  // false: they are recomputed from the base twiddle for every chunk (fewer live registers)
  static constexpr bool HOIST = HOIST_;
  static constexpr int PASSES = (R3 > 1) ? 3 : 2;
  static constexpr int RL = (R3 > 1) ? R3 : R2;  // radix of the last pass
  static constexpr int T = N / E;                // threads per transform
  static constexpr int TPB = WG / T;             // transforms per workgroup chunk
  static constexpr int C1 = N / R1, C2 = N / R2, CL = N / RL;
  static constexpr int NB1 = E / R1, NB2 = E / R2, NBL = E / RL;
  static constexpr int NRAW = E / 2;
  static constexpr int TOTAL = N * TPB;
  static constexpr int pad1(int i) { return PW1 ? i + (i >> PS1) * PW1 : i; }
  static constexpr int pad2(int i) { return PW2 ? i + (i >> PS2) * PW2 : i; }
  static constexpr int LDS1 = pad1(TOTAL - 1) + 1;
  static constexpr int LDS2 = (PASSES == 3) ? pad2(TOTAL - 1) + 1 : 0;
  static constexpr int LDS_ELEMS = LDS1 > LDS2 ? LDS1 : LDS2;
  static_assert(R1 * R2 * R3 == N, "radices must multiply to N");
  static_assert(T * TPB == WG && TPB >= 1, "workgroup must hold whole transforms");
  static_assert(E % R1 == 0 && E % R2 == 0 && E % RL == 0, "radix must divide points per thread");
  static_assert(NB1 == 1 || NB1 % 2 == 0, "first pass needs 1 or an even number of butterflies");
  static_assert(NBL == 1 || NBL % 2 == 0, "last pass needs 1 or an even number of butterflies");
};

// Thread id that puts lanes l and l+32 of a wave on adjacent butterflies.
__device__ __forceinline__ int pair_swizzle(int tid) {
  return (tid & ~63) | ((tid & 31) << 1) | ((tid >> 5) & 1);
}

// i + (i >> PS) * PW, the padded LDS slot of logical exchange index i.
template <int PS, int PW>
__device__ __forceinline__ int pad_idx(int i) {
  if constexpr (PW == 0) return i;
  return i + (i >> PS) * PW;
}

// Slots of base + e*S for e = 0..R-1. When the stride cannot carry into the padded bits in
// a data dependent way the slots are pad(base) + e*const and become ds_* immediates.
template <int PS, int PW, int S, int R>
struct PadWalk {
  static constexpr bool kWhole = (PW == 0) || (S % (1 << PS) == 0);
  // base is always a multiple of R*S in the callers, so small strides stay inside one pad period
  static constexpr bool kInside = (PW != 0) && (R * S <= (1 << PS));
  static constexpr bool kAffine = kWhole || kInside;
  static constexpr int kStep = (PW == 0) ? S : (kWhole ? S + (S >> PS) * PW : S);
  __device__ static __forceinline__ int slot(int base_logical, int base_slot, int e) {
    if constexpr (kAffine) return base_slot + e * kStep;
    return pad_idx<PS, PW>(base_logical + e * S);
  }
};

enum : int { kModeFFT = 0, kModeCopy = 1, kModeNoLds = 2 };

// Identity that the optimiser cannot see through unless HOIST: stops loop-invariant code
// motion from parking every power of a twiddle in registers for the whole batch loop.
template <bool HOIST>
__device__ __forceinline__ c32 fresh(c32 w) {
  if constexpr (!HOIST) asm volatile("" : "+v"(w.x), "+v"(w.y));
  return w;
}

template <class P, class IO, bool FWD, int MODE>
__global__ __launch_bounds__(P::WG, P::MINW) void fft_lds_kernel(const typename IO::raw* in, typename IO::raw* out,
                                                         const float2* __restrict__ tw, long long batch,
                                                         long long nchunks) {
  using raw = typename IO::raw;
  constexpr int N = P::N, E = P::E, R1 = P::R1, R2 = P::R2, R3 = P::R3, RL = P::RL, WG = P::WG;
  constexpr int C1 = P::C1, C2 = P::C2, CL = P::CL, NB1 = P::NB1, NB2 = P::NB2, NBL = P::NBL;
  constexpr int NRAW = P::NRAW, TPB = P::TPB;

  __shared__ float2 lds[P::LDS_ELEMS];

  const int tid = threadIdx.x;
  const int hi = (tid >> 5) & 1;  // upper half of the wave

  // ---- first pass: which butterflies, which raw loads (offsets in units of `raw` = 2 points).
  // Load i of a group sits i * (C1/2) raw elements after the group's base.
  constexpr int NG1 = (NB1 == 1) ? 1 : NB1 / 2;  // load groups: one column pair each
  constexpr int LPG1 = NRAW / NG1;               // loads per group
  int ld_base[NG1], ld_tr[NG1];
  int b1_tr[NB1], b1_c[NB1];
  if constexpr (NB1 == 1) {
    const int g = pair_swizzle(tid);
    b1_tr[0] = g / C1;
    b1_c[0] = g % C1;
    ld_tr[0] = b1_tr[0];
    ld_base[0] = (b1_tr[0] * N + hi * (R1 / 2) * C1 + (b1_c[0] & ~1)) >> 1;
  } else {
#pragma unroll
    for (int pi = 0; pi < NB1 / 2; ++pi) {
      const int g = 2 * (tid + WG * pi);
      b1_tr[2 * pi] = b1_tr[2 * pi + 1] = g / C1;
      b1_c[2 * pi] = g % C1;
      b1_c[2 * pi + 1] = b1_c[2 * pi] + 1;
      ld_tr[pi] = g / C1;
      ld_base[pi] = (ld_tr[pi] * N + b1_c[2 * pi]) >> 1;
    }
  }

  // ---- last pass: butterflies and raw stores, same shape with CL in place of C1
  constexpr int NGL = (NBL == 1) ? 1 : NBL / 2;
  constexpr int SPGL = NRAW / NGL;
  int st_base[NGL], st_tr[NGL];
  int bl_tr[NBL], bl_q[NBL];
  if constexpr (NBL == 1) {
    const int g = pair_swizzle(tid);
    bl_tr[0] = g / CL;
    bl_q[0] = g % CL;
    st_tr[0] = bl_tr[0];
    st_base[0] = (bl_tr[0] * N + hi * (RL / 2) * CL + (bl_q[0] & ~1)) >> 1;
  } else {
#pragma unroll
    for (int pi = 0; pi < NBL / 2; ++pi) {
      const int g = 2 * (tid + WG * pi);
      bl_tr[2 * pi] = bl_tr[2 * pi + 1] = g / CL;
      bl_q[2 * pi] = g % CL;
      bl_q[2 * pi + 1] = bl_q[2 * pi] + 1;
      st_tr[pi] = g / CL;
      st_base[pi] = (st_tr[pi] * N + bl_q[2 * pi]) >> 1;
    }
  }

  // ---- loop-invariant twiddle bases
  c32 w1[NB1];
#pragma unroll
  for (int b = 0; b < NB1; ++b) w1[b] = tw[b1_c[b]];  // W_N^c
  c32 w2[NB2];
  int b2_tr[NB2], b2_k1[NB2], b2_n3[NB2];
  if constexpr (P::PASSES == 3) {
#pragma unroll
    for (int b = 0; b < NB2; ++b) {
      const int g = tid + WG * b;
      b2_tr[b] = g / C2;
      const int q = g % C2;
      b2_k1[b] = q / R3;
      b2_n3[b] = q % R3;
      w2[b] = tw[b2_n3[b] * R1];  // W_{R2 R3}^{n3}
    }
  }

  const long long chunk_raw = (long long)P::TOTAL / 2;  // raw elements per chunk

  auto issue_loads = [&](long long chunk, raw (&dst)[NRAW]) {
    const raw* src = in + chunk * chunk_raw;
    const long long tr0 = chunk * TPB;
#pragma unroll
    for (int g = 0; g < NG1; ++g) {
      const bool ok = (TPB == 1) || (tr0 + ld_tr[g] < batch);
      const raw* p = src + ld_base[g];
#pragma unroll
      for (int i = 0; i < LPG1; ++i) dst[g * LPG1 + i] = ok ? IO::load(p + i * (C1 / 2)) : IO::zero();
    }
  };

  // Software pipeline: the rows of chunk i+1 are requested right after chunk i has been
  // unpacked, so they land while chunk i is computed and stored. The first trip is written
  // out ahead of the loop: both ways into the loop then carry "8 loads, then 8 stores" in
  // flight and the wait for the prefetched rows is a counted vmcnt that leaves the stores
  // of the previous chunk outstanding.
  raw pre[NRAW];
  auto process = [&](long long chunk) __attribute__((always_inline)) {
    c32 v[E];
    // ---- unpack the prefetched rows into butterfly order
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
        for (int n1 = 0; n1 < R1; ++n1)
          IO::unpack(pre[pi * R1 + n1], v[(2 * pi) * R1 + n1], v[(2 * pi + 1) * R1 + n1]);
    }
    if constexpr (!FWD) {
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] = make_float2(v[e].y, v[e].x);
    }

    // ---- the next chunk's loads go out now
    const long long next = chunk + gridDim.x;
    if (next < nchunks) issue_loads(next, pre);

    if constexpr (MODE != kModeCopy) {
      // ================= first pass =================
#pragma unroll
      for (int b = 0; b < NB1; ++b) {
        c32 x[R1];
#pragma unroll
        for (int j = 0; j < R1; ++j) x[j] = v[b * R1 + j];
        dft<R1>(x);
        c32 pw[R1];
        twiddle_powers<R1>(fresh<P::HOIST>(w1[b]), pw);
#pragma unroll
        for (int j = 1; j < R1; ++j) x[j] = cmul(x[j], pw[j]);
#pragma unroll
        for (int j = 0; j < R1; ++j) v[b * R1 + j] = x[j];
      }

      if constexpr (MODE == kModeFFT) {
        // ---- exchange 1: write Y1[tr][k1][c]
#pragma unroll
        for (int b = 0; b < NB1; ++b) {
          using W = PadWalk<P::PS1, P::PW1, C1, R1>;
          const int base = b1_tr[b] * N + b1_c[b];
          const int slot0 = pad_idx<P::PS1, P::PW1>(base);
#pragma unroll
          for (int k1 = 0; k1 < R1; ++k1) {
            // the column index c is not a multiple of R1*C1, so only whole-period strides are affine here
            const int s = W::kWhole ? slot0 + k1 * W::kStep : pad_idx<P::PS1, P::PW1>(base + k1 * C1);
            lds[s] = v[b * R1 + k1];
          }
        }
        __syncthreads();
      }

      if constexpr (P::PASSES == 3) {
        // ================= middle pass =================
#pragma unroll
        for (int b = 0; b < NB2; ++b) {
          c32 x[R2];
          if constexpr (MODE == kModeFFT) {
            using W = PadWalk<P::PS1, P::PW1, R3, R2>;
            const int base = b2_tr[b] * N + b2_k1[b] * C1 + b2_n3[b];
            const int slot0 = pad_idx<P::PS1, P::PW1>(base);
#pragma unroll
            for (int n2 = 0; n2 < R2; ++n2) {
              const int s = W::kWhole ? slot0 + n2 * W::kStep : pad_idx<P::PS1, P::PW1>(base + n2 * R3);
              x[n2] = lds[s];
            }
          } else {
#pragma unroll
            for (int n2 = 0; n2 < R2; ++n2) x[n2] = v[b * R2 + n2];
          }
          dft<R2>(x);
          c32 pw[R2];
          twiddle_powers<R2>(fresh<P::HOIST>(w2[b]), pw);
#pragma unroll
          for (int j = 1; j < R2; ++j) x[j] = cmul(x[j], pw[j]);
#pragma unroll
          for (int j = 0; j < R2; ++j) v[b * R2 + j] = x[j];
        }
        if constexpr (MODE == kModeFFT) {
          __syncthreads();  // every wave has read exchange 1 before it is overwritten
          // ---- exchange 2: write Y2[tr][k2][k1][n3]
#pragma unroll
          for (int b = 0; b < NB2; ++b) {
            using W = PadWalk<P::PS2, P::PW2, R1 * R3, R2>;
            const int base = b2_tr[b] * N + b2_k1[b] * R3 + b2_n3[b];
            const int slot0 = pad_idx<P::PS2, P::PW2>(base);
#pragma unroll
            for (int k2 = 0; k2 < R2; ++k2) {
              const int s = W::kWhole ? slot0 + k2 * W::kStep : pad_idx<P::PS2, P::PW2>(base + k2 * (R1 * R3));
              lds[s] = v[b * R2 + k2];
            }
          }
          __syncthreads();
        }
      }

      // ================= last pass =================
#pragma unroll
      for (int b = 0; b < NBL; ++b) {
        c32 x[RL];
        if constexpr (MODE == kModeFFT) {
          if constexpr (P::PASSES == 3) {
            // butterfly q = k1 + R1*k2 reads Y2[tr][k2][k1][n3], n3 = 0..R3-1 (stride 1)
            using W = PadWalk<P::PS2, P::PW2, 1, R3>;
            const int k1 = bl_q[b] % R1, k2 = bl_q[b] / R1;
            const int base = bl_tr[b] * N + k2 * (R1 * R3) + k1 * R3;
            const int slot0 = pad_idx<P::PS2, P::PW2>(base);
#pragma unroll
            for (int n3 = 0; n3 < R3; ++n3) x[n3] = lds[W::slot(base, slot0, n3)];
          } else {
            // butterfly q = k1 reads Y1[tr][k1][n2], n2 = 0..R2-1 (stride 1)
            using W = PadWalk<P::PS1, P::PW1, 1, R2>;
            const int base = bl_tr[b] * N + bl_q[b] * C1;
            const int slot0 = pad_idx<P::PS1, P::PW1>(base);
#pragma unroll
            for (int n2 = 0; n2 < R2; ++n2) x[n2] = lds[W::slot(base, slot0, n2)];
          }
        } else {
#pragma unroll
          for (int j = 0; j < RL; ++j) x[j] = v[b * RL + j];
        }
        dft<RL>(x);
#pragma unroll
        for (int j = 0; j < RL; ++j) v[b * RL + j] = x[j];
      }
    }

    if constexpr (!FWD) {
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] = make_float2(v[e].y, v[e].x);
    }

    // ---- store X[q + CL*k]
    {
      raw* dst = out + chunk * chunk_raw;
      const long long tr0 = chunk * TPB;
#pragma unroll
      for (int g = 0; g < NGL; ++g) {
        const bool ok = (TPB == 1) || (tr0 + st_tr[g] < batch);
        raw* p = dst + st_base[g];
#pragma unroll
        for (int i = 0; i < SPGL; ++i) {
          raw r;
          if constexpr (NBL == 1) {
            r = IO::pack(v[i], v[i + RL / 2]);
            IO::swap_halves(r);
          } else {
            r = IO::pack(v[(2 * g) * RL + i], v[(2 * g + 1) * RL + i]);
          }
          if (ok) IO::store(p + i * (CL / 2), r);
        }
      }
    }
    if constexpr (MODE == kModeFFT) __syncthreads();  // exchange buffer is free for the next chunk
  };

  long long chunk = blockIdx.x;
  if (chunk >= nchunks) return;
  issue_loads(chunk, pre);
  process(chunk);
  for (chunk += gridDim.x; chunk < nchunks; chunk += gridDim.x) process(chunk);
}

}  // namespace tfft
