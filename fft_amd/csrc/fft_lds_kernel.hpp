// Batched 1D complex FFT, one LDS-resident transform (or several small ones) per workgroup: storage traits,
// plan constants and index helpers. The transform itself is BlockFFT (fft_block.hpp); the batched kernel built on
// it is fft_lds_kernel (fft_batch_kernel.hpp).
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

#include <type_traits>

#include "butterfly.hpp"

namespace tfft {

// ---------------------------------------------------------------------------------------
// Storage traits: how two adjacent complex points travel between HBM and registers.
// ---------------------------------------------------------------------------------------
// Every point is read once and written once, so global accesses are nontemporal (streaming):
// they do not displace the twiddle tables from L2 and measure ~5 % faster on a plain copy.
typedef float f32x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));

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
  // round to nearest even in hardware: one v_cvt_pk_bf16_f32 per complex point
  __device__ static __forceinline__ unsigned cvt(c32 v) {
    const f32x2_t f = {v.x, v.y};
    const bf16x2_t r = __builtin_convertvector(f, bf16x2_t);
    return *reinterpret_cast<const unsigned*>(&r);
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

// ---------------------------------------------------------------------------------------
// Plan constants shared by host and device.
// ---------------------------------------------------------------------------------------
enum : int { kFlagPrefetch = 1, kFlagSplit = 2 };

template <int N_, int E_, int R1_, int R2_, int R3_, int WG_, int PS1_, int PW1_, int PS2_, int PW2_, int MINW_ = 1,
          int FLAGS_ = 0>
struct Plan {
  static constexpr int N = N_, E = E_, R1 = R1_, R2 = R2_, R3 = R3_, WG = WG_;
  static constexpr int PS1 = PS1_, PW1 = PW1_, PS2 = PS2_, PW2 = PW2_;
  static constexpr int MINW = MINW_;  // waves per SIMD the register allocator must leave room for
  // PREFETCH: a resident workgroup requests the rows of its next chunk before it computes the
  //           current one (costs E/2 more live 16-byte registers); without it the kernel is a
  //           plain load -> compute -> store body, meant for one workgroup per chunk.
  // SPLIT   : the LDS exchanges move real and imaginary parts in two rounds through a buffer of
  //           N floats instead of N float2: half the LDS, twice the barriers.
  static constexpr bool PREFETCH = (FLAGS_ & kFlagPrefetch) != 0;
  static constexpr bool SPLIT = (FLAGS_ & kFlagSplit) != 0;
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
  static constexpr int LDS_BYTES = LDS_ELEMS * (SPLIT ? 4 : 8);
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

// Slot of logical index base + e*S (e = 0..R-1). Two cases are affine in e, so that the slots
// become base register + immediate offset: strides that are whole pad periods, and runs that stay
// inside one pad period (INSIDE: the caller guarantees base is a multiple of R*S).
template <int PS, int PW, int S, int R, bool INSIDE>
__device__ __forceinline__ int walk_slot(int base, int base_slot, int e) {
  constexpr bool whole = (PW == 0) || (S % (1 << PS) == 0);
  constexpr bool inside = INSIDE && (R * S <= (1 << PS));
  if constexpr (whole) return base_slot + e * ((PW == 0) ? S : S + (S >> PS) * PW);
  if constexpr (inside) return base_slot + e * S;
  return pad_idx<PS, PW>(base + e * S);
}

enum : int { kModeFFT = 0, kModeCopy = 1, kModeNoLds = 2 };

typedef _Float16 mfma_h8 __attribute__((ext_vector_type(8)));
typedef float mfma_f4 __attribute__((ext_vector_type(4)));

// Identity that the optimiser cannot see through (HOIST = false): stops loop-invariant code
// motion from parking every power of a twiddle in registers for the whole batch loop.
template <bool HOIST>
__device__ __forceinline__ c32 fresh(c32 w) {
  if constexpr (!HOIST) asm volatile("" : "+v"(w.x), "+v"(w.y));
  return w;
}

// Which chunk the v-th workgroup-sized unit of work is. The dispatcher deals workgroups to the eight
// XCDs round-robin (workgroup b runs on XCD b mod 8), so with the identity map every XCD -- its L2 and
// its address-translation cache -- touches every eighth chunk of the whole range. Handing each XCD
// runs of 2^lb consecutive chunks instead keeps its stream contiguous (256 KiB ... 2 MiB per run
// measured best: +2 ... +6 points of HBM bandwidth depending on the length). A bijection on
// [0, total): whole groups of 8 * 2^lb chunks are permuted, the remainder keeps its order.
__device__ __forceinline__ long long xcd_chunk(long long v, long long total, int lb) {
  const long long span = 8ll << lb;
  if (lb <= 0 || v >= (total & ~(span - 1))) return v;
  const long long slot = v >> 3, xcd = v & 7;
  return ((slot >> lb) << (lb + 3)) + (xcd << lb) + (slot & ((1ll << lb) - 1));
}

}  // namespace tfft
