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
enum : int { kFlagPrefetch = 1, kFlagSplit = 2, kFlagHoist = 4, kFlagMfmaMid = 8 };

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
  // HOIST   : twiddle powers are loop invariants the compiler may keep in VGPRs across chunks.
  static constexpr bool PREFETCH = (FLAGS_ & kFlagPrefetch) != 0;
  static constexpr bool SPLIT = (FLAGS_ & kFlagSplit) != 0;
  static constexpr bool HOIST = (FLAGS_ & kFlagHoist) != 0;
  // MFMA_MID: (N = 16*16*16 only) the middle radix-16 pass runs on the matrix cores: per k1 the
  //           16x16 complex DFT matrix times the 16x16 block Y1[k1][m1][m2] is two
  //           v_mfma_f32_16x16x32_f16 (real and imaginary rows of the real-ified matrix, K = 32 =
  //           16 points x (re, im)); exchange 1 then carries fp16 pairs. For 16-bit storage only:
  //           the fp16 operands cost ~5e-4 relative accuracy.
  static constexpr bool MFMA_MID = (FLAGS_ & kFlagMfmaMid) != 0;
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
  // fp16 exchange of the MFMA middle pass: rows (k1, m2) of 16 half2 (m1 fastest), 80-byte pitch
  static constexpr int MFMA_PITCH = 20;  // dwords; 16-byte aligned rows, 2-way write conflicts at most
  static constexpr int LDS_MFMA = MFMA_MID ? (N / 16) * MFMA_PITCH * 4 : 0;
  static constexpr int LDS_BYTES = (LDS_ELEMS * (SPLIT ? 4 : 8)) > LDS_MFMA ? (LDS_ELEMS * (SPLIT ? 4 : 8)) : LDS_MFMA;
  static_assert(!MFMA_MID || (R1 == 16 && R2 == 16 && R3 == 16 && E == 16 && WG == 256), "MFMA middle pass: N = 4096 plan only");
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

// Identity that the optimiser cannot see through unless HOIST: stops loop-invariant code
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

template <class P, class IO, bool FWD, int MODE>
__global__ __launch_bounds__(P::WG, P::MINW) void fft_lds_kernel(const typename IO::raw* in, typename IO::raw* out,
                                                                 const float2* __restrict__ tw, long long batch,
                                                                 long long nchunks, int xcd_log2) {
  using raw = typename IO::raw;
  constexpr int N = P::N, E = P::E, R1 = P::R1, R2 = P::R2, R3 = P::R3, RL = P::RL, WG = P::WG;
  constexpr int C1 = P::C1, C2 = P::C2, CL = P::CL, NB1 = P::NB1, NB2 = P::NB2, NBL = P::NBL;
  constexpr int NRAW = P::NRAW, TPB = P::TPB;
  constexpr bool SPLIT = P::SPLIT;

  __shared__ __attribute__((aligned(16))) char lds_raw[P::LDS_BYTES];
  float2* const lds2 = reinterpret_cast<float2*>(lds_raw);
  float* const lds1 = reinterpret_cast<float*>(lds_raw);

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

  // ---- the first chunk's rows are requested before anything else is set up: a workgroup's
  // lifetime starts with this HBM round trip, and everything below (store offsets, twiddle
  // bases, LDS slots) is computed while it is in flight.
  const long long chunk_raw = (long long)P::TOTAL / 2;  // raw elements per chunk
  // (the lambdas take the position in the launch order; xcd_chunk turns it into the chunk)
  auto issue_loads = [&](long long unit, raw (&dst)[NRAW]) {
    const long long chunk = xcd_chunk(unit, nchunks, xcd_log2);
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
  raw pre[NRAW];
  long long chunk = blockIdx.x;
  if (chunk >= nchunks) return;
  issue_loads(chunk, pre);

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

  // ---- middle pass butterflies (three-pass plans): butterfly (k1, n3) of transform tr
  int b2_tr[NB2], b2_k1[NB2], b2_n3[NB2];
  if constexpr (P::PASSES == 3) {
#pragma unroll
    for (int b = 0; b < NB2; ++b) {
      const int g = tid + WG * b;
      b2_tr[b] = g / C2;
      const int q = g % C2;
      b2_k1[b] = q / R3;
      b2_n3[b] = q % R3;
    }
  }

  // ---- twiddle bases, one per butterfly: W_N^c after the first pass, W_{R2 R3}^{n3} after the middle one
  c32 w1[NB1], w2[NB2];
#pragma unroll
  for (int b = 0; b < NB1; ++b) w1[b] = tw[b1_c[b]];
  if constexpr (P::PASSES == 3) {
#pragma unroll
    for (int b = 0; b < NB2; ++b) w2[b] = tw[b2_n3[b] * R1];
  }

  // ---- matrix-core middle pass: this lane's fragments of the real-ified DFT-16 matrix and its
  // four output twiddles. Lane l = (g = l >> 4, i = l & 15) supplies A[row i][k = 8g .. 8g+7] with
  // k = 2*m1 + part, holds B rows m1 = 4g .. 4g+3 of column m2 = i, and receives rows k2 = 4g .. 4g+3.
  const int mf_g = (tid >> 4) & 3, mf_i = tid & 15;
  mfma_h8 mf_are, mf_aim;
  c32 mf_tw[4];
  if constexpr (P::MFMA_MID) {
#pragma unroll
    for (int jj = 0; jj < 4; ++jj) {
      const int m1 = 4 * mf_g + jj;
      const c32 f = tw[((mf_i * m1) & 15) * (N / 16)];  // W_16^(k2*m1) = (cos, -sin)
      mf_are[2 * jj] = (_Float16)f.x;       // Yr += cos * Xr
      mf_are[2 * jj + 1] = (_Float16)(-f.y);  //     + sin * Xi
      mf_aim[2 * jj] = (_Float16)f.y;       // Yi += -sin * Xr
      mf_aim[2 * jj + 1] = (_Float16)f.x;   //     + cos * Xi
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) mf_tw[j] = tw[(mf_i * (4 * mf_g + j)) * R1];  // W_256^(m2*k2)
  }

  // ---- LDS slots. Writer element (b, j) and reader element (b, j) of each exchange.
  // exchange 1 holds Y1[tr][k1][c] at logical index tr*N + k1*C1 + c
  auto ex1_wslot = [&](int b, int k1) {
    const int base = b1_tr[b] * N + b1_c[b];
    return walk_slot<P::PS1, P::PW1, C1, R1, false>(base, pad_idx<P::PS1, P::PW1>(base), k1);
  };
  auto ex1_rslot_mid = [&](int b, int n2) {  // middle pass: butterfly (k1, n3) reads n2 = 0..R2-1, stride R3
    const int base = b2_tr[b] * N + b2_k1[b] * C1 + b2_n3[b];
    return walk_slot<P::PS1, P::PW1, R3, R2, false>(base, pad_idx<P::PS1, P::PW1>(base), n2);
  };
  auto ex1_rslot_last = [&](int b, int n2) {  // two-pass plans: butterfly k1 reads n2 = 0..R2-1, stride 1
    const int base = bl_tr[b] * N + bl_q[b] * C1;
    return walk_slot<P::PS1, P::PW1, 1, R2, true>(base, pad_idx<P::PS1, P::PW1>(base), n2);
  };
  // exchange 2 holds Y2[tr][k2][k1][n3] at logical index tr*N + k2*(R1*R3) + k1*R3 + n3
  auto ex2_wslot = [&](int b, int k2) {
    const int base = b2_tr[b] * N + b2_k1[b] * R3 + b2_n3[b];
    return walk_slot<P::PS2, P::PW2, R1 * R3, R2, false>(base, pad_idx<P::PS2, P::PW2>(base), k2);
  };
  auto ex2_rslot = [&](int b, int n3) {  // last pass: butterfly q = k1 + R1*k2 reads n3 = 0..R3-1
    const int k1 = bl_q[b] % R1, k2 = bl_q[b] / R1;
    const int base = bl_tr[b] * N + k2 * (R1 * R3) + k1 * R3;
    return walk_slot<P::PS2, P::PW2, 1, R3, true>(base, pad_idx<P::PS2, P::PW2>(base), n3);
  };

  // One exchange: every thread writes its E values (writer radix RW: element e = b*RW + j) and
  // reads back E values in the next pass's order (reader radix RR). `lead` inserts the barrier
  // that protects a previous exchange's readers from this one's writes.
  auto exchange = [&](c32 (&v)[E], auto wslot, auto rslot, auto rw_tag, auto rr_tag, bool lead) {
    constexpr int RW = decltype(rw_tag)::value, RR = decltype(rr_tag)::value;
    if (lead) __syncthreads();
    if constexpr (!SPLIT) {
#pragma unroll
      for (int e = 0; e < E; ++e) lds2[wslot(e / RW, e % RW)] = v[e];
      __syncthreads();
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] = lds2[rslot(e / RR, e % RR)];
    } else {
      float re[E];
#pragma unroll
      for (int e = 0; e < E; ++e) lds1[wslot(e / RW, e % RW)] = v[e].x;
      __syncthreads();
#pragma unroll
      for (int e = 0; e < E; ++e) re[e] = lds1[rslot(e / RR, e % RR)];
      __syncthreads();
#pragma unroll
      for (int e = 0; e < E; ++e) lds1[wslot(e / RW, e % RW)] = v[e].y;
      __syncthreads();
#pragma unroll
      for (int e = 0; e < E; ++e) v[e] = make_float2(re[e], lds1[rslot(e / RR, e % RR)]);
    }
  };

  // radix-R butterflies on v[b*R .. b*R+R-1], then the inter-pass twiddles w[b]^j
  auto butterflies = [&](c32 (&v)[E], auto r_tag, const c32* w) {
    constexpr int R = decltype(r_tag)::value;
#pragma unroll
    for (int b = 0; b < E / R; ++b) {
      c32 x[R];
#pragma unroll
      for (int j = 0; j < R; ++j) x[j] = v[b * R + j];
      dft<R>(x);
      if (w != nullptr) apply_twiddles<R>(x, fresh<P::HOIST>(w[b]));
#pragma unroll
      for (int j = 0; j < R; ++j) v[b * R + j] = x[j];
    }
  };

  auto process = [&](long long unit, bool more_after) __attribute__((always_inline)) {
    const long long chunk = xcd_chunk(unit, nchunks, xcd_log2);
    c32 v[E];
    // ---- unpack the rows into butterfly order
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

    if constexpr (P::PREFETCH) {
      // the next chunk's loads go out now and land while this chunk is computed
      const long long next = unit + gridDim.x;
      if (next < nchunks) issue_loads(next, pre);
    }

    if constexpr (MODE != kModeCopy) {
      using I1 = std::integral_constant<int, R1>;
      using I2 = std::integral_constant<int, R2>;
      using IL = std::integral_constant<int, RL>;
      butterflies(v, I1{}, w1);
      if constexpr (P::MFMA_MID && MODE == kModeFFT) {
        // ---- exchange 1 as fp16 pairs: thread (column c = 16*m1 + m2) writes Y1[k1][c], k1 = 0..15
        unsigned* const ldsh = reinterpret_cast<unsigned*>(lds_raw);
        {
          const int m1 = b1_c[0] >> 4, m2 = b1_c[0] & 15;
#pragma unroll
          for (int k1 = 0; k1 < 16; ++k1) {
            const __half2 h = __float22half2_rn(v[k1]);
            ldsh[(k1 * 16 + m2) * P::MFMA_PITCH + m1] = *reinterpret_cast<const unsigned*>(&h);
          }
        }
        __syncthreads();
        // ---- middle pass on the matrix cores: wave w owns k1 = 4w .. 4w+3
        const int wave = tid >> 6;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
          const int k1 = 4 * wave + t;
          const mfma_h8 bfrag = *reinterpret_cast<const mfma_h8*>(&ldsh[(k1 * 16 + mf_i) * P::MFMA_PITCH + 4 * mf_g]);
          mfma_f4 cr = {0.f, 0.f, 0.f, 0.f}, ci = {0.f, 0.f, 0.f, 0.f};
          cr = __builtin_amdgcn_mfma_f32_16x16x32_f16(mf_are, bfrag, cr, 0, 0, 0);
          ci = __builtin_amdgcn_mfma_f32_16x16x32_f16(mf_aim, bfrag, ci, 0, 0, 0);
#pragma unroll
          for (int j = 0; j < 4; ++j) v[t * 4 + j] = cmul(make_float2(cr[j], ci[j]), mf_tw[j]);
        }
        // ---- exchange 2: lane (g, i) of wave w holds Y2[k1 = 4w+t][k2 = 4g+j][n3 = i] in v[4t + j]
        auto ex2_wslot_mfma = [&](int t, int j) {
          return pad_idx<P::PS2, P::PW2>((4 * mf_g + j) * (R1 * R3) + (4 * wave + t) * R3 + mf_i);
        };
        exchange(v, ex2_wslot_mfma, ex2_rslot, std::integral_constant<int, 4>{}, IL{}, true);
      } else if constexpr (P::PASSES == 3) {
        if constexpr (MODE == kModeFFT) exchange(v, ex1_wslot, ex1_rslot_mid, I1{}, I2{}, false);
        butterflies(v, I2{}, w2);
        if constexpr (MODE == kModeFFT) exchange(v, ex2_wslot, ex2_rslot, I2{}, IL{}, true);
      } else {
        if constexpr (MODE == kModeFFT) exchange(v, ex1_wslot, ex1_rslot_last, I1{}, IL{}, false);
      }
      butterflies(v, IL{}, static_cast<const c32*>(nullptr));
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
    // the exchange buffer is reused by the next chunk of this workgroup
    if (MODE == kModeFFT && more_after) __syncthreads();
  };

  if constexpr (P::PREFETCH) {
    // Software pipeline. The first trip is written out ahead of the loop: both ways into the loop
    // then carry "loads, then stores" in flight, and the wait for the prefetched rows is a counted
    // vmcnt that leaves the previous chunk's stores outstanding.
    process(chunk, true);
    for (chunk += gridDim.x; chunk < nchunks; chunk += gridDim.x) process(chunk, true);
  } else {
    for (;;) {
      const long long next = chunk + gridDim.x;
      process(chunk, next < nchunks);
      if (next >= nchunks) break;
      chunk = next;
      issue_loads(chunk, pre);
    }
  }
}

}  // namespace tfft
