// The shipped plans: one per transform length. Radices, points per thread, workgroup size
// and LDS padding (PS, PW per exchange; slot = i + (i >> PS) * PW) were chosen with
// tools/lds_conflicts.py so that every ds_read_b64 of the exchanges is conflict free.
//
// This table takes the place of the reference's batch_per_block<N> / warp_per_block<N>
// (thunderfft/include/thunderfft/detail/thunderfft_impl.cuh:14-32) and of its tuned
// shared-memory layouts (thunderfft/include/thunderfft/thunderfft_layout_tuned.h:6-97).
#pragma once
#include <type_traits>
#include "fft_batch_kernel.hpp"
#include "fft_dit_kernel.hpp"
#include "fft_mfma_kernel.hpp"

namespace tfft {

// Two exchange forms exist per length: whole complex words through an 8-byte-per-point buffer
// ("Full": fewer barriers, ~34 KiB of LDS for 4096 points, 4 workgroups per CU) and real parts
// then imaginary parts through a 4-byte-per-point buffer ("Split": twice the barriers, half the
// LDS, 6-8 workgroups per CU). The default of each length is the one that measured faster on
// MI355X at 2^28 points; the other stays reachable as the A/B plan (tfft_exec_options.plan = 3).
//
//                           N      E   R1  R2  R3  WG   PS1 PW1 PS2 PW2 MINW FLAGS
using Plan64Full = Plan<64, 16, 16, 4, 1, 256, 5, 1, 0, 0>;
using Plan128Full = Plan<128, 16, 16, 8, 1, 256, 5, 1, 0, 0>;
using Plan256Full = Plan<256, 16, 16, 16, 1, 256, 5, 1, 0, 0>;
using Plan512Full = Plan<512, 16, 16, 8, 4, 256, 5, 4, 5, 1>;
using Plan1024Full = Plan<1024, 16, 16, 16, 4, 256, 4, 1, 5, 1>;
using Plan2048Full = Plan<2048, 16, 16, 16, 8, 256, 4, 1, 5, 1>;
using Plan4096Full = Plan<4096, 16, 16, 16, 16, 256, 4, 1, 5, 1>;
using Plan8192Full = Plan<8192, 32, 32, 16, 16, 256, 4, 1, 5, 1>;
using Plan16384Full = Plan<16384, 32, 32, 32, 16, 512, 9, 16, 5, 1>;

using Plan64Split = Plan<64, 16, 16, 4, 1, 256, 5, 1, 0, 0, 6, kFlagSplit>;
using Plan128Split = Plan<128, 16, 16, 8, 1, 256, 5, 1, 0, 0, 6, kFlagSplit>;
using Plan256Split = Plan<256, 16, 16, 16, 1, 256, 5, 1, 0, 0, 5, kFlagSplit>;
using Plan512Split = Plan<512, 16, 16, 8, 4, 256, 5, 4, 5, 1, 6, kFlagSplit>;
using Plan1024Split = Plan<1024, 16, 16, 16, 4, 256, 6, 4, 5, 1, 6, kFlagSplit>;
using Plan2048Split = Plan<2048, 16, 16, 16, 8, 256, 7, 8, 5, 1, 5, kFlagSplit>;
using Plan4096Split = Plan<4096, 16, 16, 16, 16, 256, 8, 16, 5, 1, 6, kFlagSplit>;
using Plan8192Split = Plan<8192, 32, 32, 16, 16, 256, 8, 16, 5, 1, 4, kFlagSplit>;
using Plan16384Split = Plan<16384, 32, 32, 32, 16, 512, 5, 1, 5, 1, 4, kFlagSplit>;

// "Wide" three-pass forms of the short lengths: radix 4 or 8 first and last, so that the row
// segments of the first-pass loads and last-pass stores are N/4 or N/8 points long (128 ... 512
// bytes) instead of N/16 (32 bytes at N = 64), at the price of a second LDS exchange.
using Plan64Wide = Plan<64, 16, 4, 4, 4, 256, 5, 4, 5, 1>;
using Plan128Wide = Plan<128, 16, 4, 8, 4, 256, 5, 4, 5, 1>;
using Plan256Wide = Plan<256, 16, 4, 16, 4, 256, 4, 1, 5, 1>;
using Plan512Wide = Plan<512, 16, 8, 16, 4, 256, 6, 4, 5, 1, 5, kFlagSplit>;

// Small-workgroup forms: 64 or 128 threads per workgroup, so one workgroup requests 8 or 16 KiB
// at a time instead of 32 KiB (whole complex words through LDS, 8-17 KiB of it).
using Plan64Small = Plan<64, 16, 4, 4, 4, 64, 5, 4, 5, 1>;
using Plan128Small = Plan<128, 16, 4, 8, 4, 64, 5, 4, 5, 1>;
using Plan256Small = Plan<256, 16, 16, 16, 1, 64, 5, 1, 0, 0>;
using Plan256Mid = Plan<256, 16, 16, 16, 1, 128, 5, 1, 0, 0>;
using Plan512Mid = Plan<512, 16, 16, 8, 4, 128, 5, 4, 5, 1>;
using Plan1024Mid = Plan<1024, 16, 16, 16, 4, 128, 4, 1, 5, 1>;
using Plan512Small = Plan<512, 16, 16, 8, 4, 64, 5, 4, 5, 1>;
using Plan1024Small = Plan<1024, 16, 16, 16, 4, 64, 4, 1, 5, 1>;
using Plan2048Small = Plan<2048, 16, 16, 16, 8, 128, 4, 1, 5, 1>;
// Two passes with 32 points per thread (one LDS exchange instead of two): 64 threads, 16 KiB per
// workgroup; ~225 VGPRs, two waves per SIMD, and still ~5 % faster than the three-pass forms.
using Plan512E32 = Plan<512, 32, 16, 32, 1, 64, 6, 1, 0, 0>;
using Plan1024E32 = Plan<1024, 32, 32, 32, 1, 64, 6, 1, 0, 0>;
// 128 threads x 32 points (plan 4 of fp32 N = 4096). Which of the two is faster depends on the device's state: in a
// process's first launches (3 + 20 from idle) the 256 x 16 default reads 77.8-78.2 % of 8 TB/s and this plan
// 77.1-78.0 %; on a settled device (100 launches behind 60 others) the default drops to 76.2-76.6 % and this plan keeps
// 77.5-78.0 %. Callers that transform in long runs gain 1.3 points with tfft_exec_options.plan = 4.
using Plan4096E32F = Plan<4096, 32, 16, 16, 16, 128, 4, 1, 5, 1, 2>;

// default / A-B alternative per length
using Plan64 = Plan64Small;       using Plan64Alt = Plan64Wide;
using Plan128 = Plan128Small;     using Plan128Alt = Plan128Full;
using Plan256 = Plan256Mid;       using Plan256Alt = Plan256Small;
using Plan512 = Plan512E32;       using Plan512Alt = Plan512Small;
using Plan1024 = Plan1024E32;     using Plan1024Alt = Plan1024Small;
using Plan2048 = Plan2048Small;   using Plan2048Alt = Plan2048Split;
using Plan4096 = Plan4096Full;    using Plan4096Alt = Plan4096Split;
using Plan8192 = Plan8192Full;    using Plan8192Alt = Plan8192Split;
using Plan16384 = Plan16384Split; using Plan16384Alt = Plan16384Full;

// A resident, software-pipelined walk of the batch (prefetches the next chunk's rows); measured
// slower than one workgroup per chunk on MI355X and kept for that comparison only.
using Plan4096ResidentSplit = Plan<4096, 16, 16, 16, 16, 256, 8, 16, 5, 1, 5, kFlagSplit | kFlagPrefetch>;
using Plan4096Resident = Plan<4096, 16, 16, 16, 16, 256, 4, 1, 5, 1, 1, kFlagPrefetch>;

// Per-launch shape the host chooses for a kernel.
struct LaunchShape {
  unsigned grid = 0;       // workgroups (resident kernels: the grid that walks the batch)
  unsigned extra_lds = 0;  // unused dynamic LDS, only lowers co-residency
  int xcd_log2 = 0;        // runs of 2^xcd_log2 consecutive chunks per XCD
};

// What the host needs to launch one instantiation.
struct KernelEntry {
  const void* fn;  // __global__ function address (for the occupancy query)
  void (*launch)(const void* in, void* out, const float2* tw, long long batch, long long nchunks,
                 const LaunchShape& shape, hipStream_t stream);
  int n, threads, tpb, e, radix[3], lds_bytes;
  int resident_per_cu = 0;  // > 0: a resident grid of this many workgroups per CU is the kernel's default launch
  int xcd_log2 = 0;         // log2 of the run of consecutive chunks one XCD takes (xcd_chunk in fft_lds_kernel.hpp)
  int xcd_log2_in_place = -1;  // the same for in-place calls (-1: as xcd_log2)
  bool one_workgroup_per_transform = false;  // grid = batch, whatever LaunchShape::grid says (no resident form)
};

// Measured best run of consecutive chunks per XCD (interleaved A/B at 2^28 points). Out of place the rate also
// depends on where the two buffers lie physically: for 16 KiB chunks, runs of 16 give 81 % or 69-75 % from one
// allocation to the next, runs of 64 give 78 % either way, so in-place calls take the short runs and
// out-of-place calls the long ones.
constexpr int default_xcd_log2(int n, int bytes_per_point, bool in_place) {
  if (bytes_per_point != 8) return 4;
  if (n <= 128) return 7;
  if (n <= 2048) return in_place ? 4 : 6;
  return n == 4096 || n == 16384 ? 5 : 4;
}

// Plans whose one-workgroup-per-chunk launch runs a straight-line instantiation of the kernel instead of the single
// trip of its chunk loop. N = 16384: 72.3 -> 75.3 % of 8 TB/s (all sixteen row loads of a thread in flight at once;
// in the loop form the last two left after the first had arrived). Measured the other way for the shorter lengths
// (interleaved A/B of the two builds, x -> y / in place): N = 2048 76.4 -> 75.1, N = 256 ... 1024 77.5 -> 76.0 and
// 80.3 -> 74.3 in place at N = 512, where the 32-point plans drop from 232 to 135 VGPRs and a third wave per SIMD
// spreads the addresses in flight; N = 64, 128, 4096 equal within the run-to-run spread. So only N = 16384 opts in.
template <class P>
constexpr bool kStraightLine = P::N >= 16384 && !P::PREFETCH;

template <class P, class IO, bool FWD, int MODE>
void launch_plan(const void* in, void* out, const float2* tw, long long batch, long long nchunks,
                 const LaunchShape& shape, hipStream_t stream) {
  using raw = typename IO::raw;
  // extra_lds: unused dynamic LDS that only lowers the number of co-resident workgroups per CU
  // A grid that covers every chunk runs the straight-line instantiation where the plan has one (kStraightLine),
  // a smaller (resident) grid the loop.
  auto go = [&](auto kernel) {
    if (shape.extra_lds)
      (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)shape.extra_lds);
    kernel<<<dim3(shape.grid), dim3(P::WG), shape.extra_lds, stream>>>(static_cast<const raw*>(in), static_cast<raw*>(out),
                                                                      tw, batch, nchunks, shape.xcd_log2);
  };
  if constexpr (kStraightLine<P>) {
    if ((long long)shape.grid >= nchunks) return go(&fft_lds_kernel<P, IO, FWD, MODE, false>);
  }
  go(&fft_lds_kernel<P, IO, FWD, MODE, true>);
}

template <class P, class IO, bool FWD, int MODE>
KernelEntry make_entry() {
  KernelEntry e;
  // (the occupancy query that sizes resident grids asks about the loop instantiation, the one such grids run)
  e.fn = reinterpret_cast<const void*>(&fft_lds_kernel<P, IO, FWD, MODE, true>);
  e.launch = &launch_plan<P, IO, FWD, MODE>;
  e.n = P::N;
  e.threads = P::WG;
  e.tpb = P::TPB;
  e.e = P::E;
  e.radix[0] = P::R1;
  e.radix[1] = P::R2;
  e.radix[2] = P::R3;
  e.lds_bytes = P::LDS_BYTES;
  // software-pipelined plans are resident grids by construction: as many workgroups per CU as MINW waves per SIMD hold
  if (P::PREFETCH) e.resident_per_cu = P::MINW * 256 / P::WG;
  e.xcd_log2 = default_xcd_log2(P::N, IO::kBytesPerPoint, false);
  e.xcd_log2_in_place = default_xcd_log2(P::N, IO::kBytesPerPoint, true);
  return e;
}

// Matrix-core kernel of fft_mfma_kernel.hpp as a table entry: N = 256 R1, 16 / R1 transforms per workgroup.
template <class T16, bool FWD, bool DOUBLE, int MINW, bool PERSIST, int R1, bool INCR, int Q>
void launch_mfma4096(const void* in, void* out, const float2* tw, long long batch, long long nchunks,
                     const LaunchShape& shape, hipStream_t stream) {
  // one-shot: one workgroup per tile of 4096 points; resident: the caller's grid (workgroups per CU x CUs)
  const unsigned blocks = PERSIST ? shape.grid : (unsigned)nchunks;
  fft_mfma4096_kernel<T16, FWD, DOUBLE, MINW, PERSIST, R1, INCR, false, false, Q>
      <<<dim3(blocks), dim3(Mfma4096::WG), 0, stream>>>(static_cast<const u32x4_t*>(in), static_cast<u32x4_t*>(out), tw,
                                                        batch, shape.xcd_log2);
}

// Q < 16 (R1 = 1): N = 16 Q = 64 or 128, 4096 / N transforms per workgroup.
template <class T16, bool FWD, bool DOUBLE, int MINW, bool PERSIST = false, int R1 = 16, bool INCR = false, int Q = 16>
KernelEntry make_mfma_entry() {
  KernelEntry e;
  e.fn = reinterpret_cast<const void*>(&fft_mfma4096_kernel<T16, FWD, DOUBLE, MINW, PERSIST, R1, INCR, false, false, Q>);
  e.launch = &launch_mfma4096<T16, FWD, DOUBLE, MINW, PERSIST, R1, INCR, Q>;
  e.n = 16 * Q * R1;
  e.threads = Mfma4096::WG;
  e.tpb = 256 / (Q * R1);
  e.e = 16;
  e.radix[0] = R1 == 1 ? Q : R1;
  e.radix[1] = 16;
  e.radix[2] = R1 == 1 ? 1 : 16;
  e.lds_bytes = 4 * (mfma_out_words<R1>() + (DOUBLE ? Mfma4096::IMG : 0));
  // three workgroups per CU measured best for the resident form (two: 66 %, three: 72-75 %, four: 67 %)
  e.resident_per_cu = PERSIST ? 3 : 0;
  // Runs of consecutive tiles per XCD. In place: 8 tiles (128 KiB), +0.5 ... 2 points over 16 in eight settled A/Bs
  // and in bench.py alike (fp16 / bf16, N = 256 ... 4096: 79.5-80.2 -> 81.4-82.0 %). x -> y stays at 16: runs of 8 won
  // the interleaved A/Bs there too (+0.2 ... 1.5) but lost 2-3 points in bench.py's own process, whose two buffers
  // lie elsewhere -- the rate of two streams depends on where they lie, and the longer run is the one that holds up
  // either way. Runs of 32 and more are 2-6 points behind in both forms.
  e.xcd_log2 = 4;
  e.xcd_log2_in_place = PERSIST ? 4 : 3;
  return e;
}

// Measurement build of the default one-shot matrix-core kernel of N = 256 R1 (REPEAT: the passes run `repeats`
// times per tile on chip). KernelEntry::launch has no slot for the count, so these are launched directly.
template <class T16, int R1>
void launch_mfma_repeat(const void* in, void* out, const float2* tw, long long batch, unsigned repeats, unsigned io_mask,
                        hipStream_t stream) {
  constexpr bool INCR = R1 > 1;
  constexpr bool DOUBLE = R1 != 1;  // the defaults: N = 256 one image, the others two
  const long long tiles = (batch * R1 + 15) / 16;
  fft_mfma4096_kernel<T16, true, DOUBLE, DOUBLE ? 4 : 8, false, R1, INCR, true>
      <<<dim3((unsigned)tiles), dim3(Mfma4096::WG), 0, stream>>>(static_cast<const u32x4_t*>(in), static_cast<u32x4_t*>(out),
                                                                 tw, batch, 3, repeats, io_mask);
}
template <class T16>
bool launch_mfma_repeat_n(int n, const void* in, void* out, const float2* tw, long long batch, unsigned repeats,
                          unsigned io_mask, hipStream_t stream) {
  switch (n) {
    case 256: launch_mfma_repeat<T16, 1>(in, out, tw, batch, repeats, io_mask, stream); return true;
    case 512: launch_mfma_repeat<T16, 2>(in, out, tw, batch, repeats, io_mask, stream); return true;
    case 1024: launch_mfma_repeat<T16, 4>(in, out, tw, batch, repeats, io_mask, stream); return true;
    case 2048: launch_mfma_repeat<T16, 8>(in, out, tw, batch, repeats, io_mask, stream); return true;
    case 4096: launch_mfma_repeat<T16, 16>(in, out, tw, batch, repeats, io_mask, stream); return true;
    default: return false;
  }
}
// io_mask: bit 0 = with the tile's loads, bit 1 = with its stores (3 = the whole kernel)
bool launch_repeat_f16(int n, const void* in, void* out, const float2* tw, long long batch, unsigned repeats, unsigned io_mask,
                       hipStream_t s);
bool launch_repeat_bf16(int n, const void* in, void* out, const float2* tw, long long batch, unsigned repeats, unsigned io_mask,
                        hipStream_t s);

// The matrix-core kernels of one 16-bit storage type for N = 256 R1 < 4096. Mode 0, the default: N = 256 (no first
// pass, no twiddle table to fetch) one workgroup per tile with one LDS image (12); otherwise one workgroup per tile
// with two images and the pass-1 twiddles built incrementally (10): on a settled device 77-80 % of 8 TB/s against
// 76 % for the resident grid with the lane-ordered twiddle table (11), which had looked level while the device was
// still ramping up. 6: one workgroup per tile, table; 12: plan 10 with one LDS image (eight workgroups per CU).
template <class T16, int R1>
bool mfma_entry_for(bool forward, int mode, KernelEntry* out) {
  constexpr bool INCR = R1 > 1;
  if (mode == 0) mode = R1 == 1 ? 12 : 10;
  if (mode == 11) { *out = forward ? make_mfma_entry<T16, true, true, 3, true, R1>() : make_mfma_entry<T16, false, true, 3, true, R1>(); return true; }
  if (mode == 6) { *out = forward ? make_mfma_entry<T16, true, true, 4, false, R1>() : make_mfma_entry<T16, false, true, 4, false, R1>(); return true; }
  if (mode == 10) { *out = forward ? make_mfma_entry<T16, true, true, 4, false, R1, INCR>() : make_mfma_entry<T16, false, true, 4, false, R1, INCR>(); return true; }
  if (mode == 12) { *out = forward ? make_mfma_entry<T16, true, false, 8, false, R1, INCR>() : make_mfma_entry<T16, false, false, 8, false, R1, INCR>(); return true; }
  return false;
}

// N = 64, 128 (Q = 4, 8): the N = 256 kernel with a block-diagonal second pass, one LDS image, eight workgroups per CU.
// Mode 0 and 12 only; the fp32-VALU kernels of these lengths are modes 3 and 4.
template <class T16, int Q>
bool mfma_small_entry_for(bool forward, int mode, KernelEntry* out) {
  if (mode != 0 && mode != 12) return false;
  *out = forward ? make_mfma_entry<T16, true, false, 8, false, 1, false, Q>() : make_mfma_entry<T16, false, false, 8, false, 1, false, Q>();
  return true;
}

// N = 4096 R0 (8192, 16384), 16-bit storage: fft_mfma_big_kernel, one workgroup per transform.
template <class T16, bool FWD, int R0, int MINW, bool H2>
void launch_mfma_big(const void* in, void* out, const float2* tw, long long batch, long long nchunks,
                     const LaunchShape& shape, hipStream_t stream) {
  (void)nchunks;
  fft_mfma_big_kernel<T16, FWD, R0, MINW, H2><<<dim3((unsigned)batch), dim3(Mfma4096::WG), 0, stream>>>(
      static_cast<const u32x4_t*>(in), static_cast<u32x4_t*>(out), tw, batch, shape.xcd_log2);
}

template <class T16, bool FWD, int R0, int MINW, bool H2 = false>
KernelEntry make_mfma_big_entry() {
  KernelEntry e;
  e.fn = reinterpret_cast<const void*>(&fft_mfma_big_kernel<T16, FWD, R0, MINW, H2>);
  e.launch = &launch_mfma_big<T16, FWD, R0, MINW, H2>;
  e.n = 4096 * R0;
  e.threads = Mfma4096::WG;
  e.tpb = 1;
  e.e = 16 * R0;
  e.radix[0] = R0;
  e.radix[1] = 4096;
  e.radix[2] = 1;
  e.lds_bytes = 4 * 2 * mfma_out_words<16>();
  // runs of consecutive transforms per XCD, from settled interleaved A/Bs (50 + launches per variant): N = 16384 best
  // at 16 (75.2 % x -> y; 4: 71 %, 8: 75.0, 32: 74, 128: 72-73); N = 8192 x -> y at 4 (+0.6 ... 1.6 points over 16
  // in four A/Bs, fp16 and bf16; 2: -3), in place at 8 (+0.7 ... 1.6)
  e.xcd_log2 = R0 == 2 ? 2 : 4;
  e.xcd_log2_in_place = R0 == 2 ? 3 : 4;
  e.one_workgroup_per_transform = true;
  return e;
}

// N = 8192, fp32: fft_dit_kernel, one workgroup per transform on the 4096-point block transform.
template <bool FWD, int MINW>
void launch_dit(const void* in, void* out, const float2* tw, long long batch, long long nchunks, const LaunchShape& shape,
                hipStream_t stream) {
  (void)nchunks;
  fft_dit_kernel<Plan4096Full, FWD, MINW><<<dim3((unsigned)batch), dim3(Plan4096Full::WG), 0, stream>>>(
      static_cast<const float4*>(in), static_cast<float4*>(out), tw, batch, shape.xcd_log2);
}

template <bool FWD, int MINW>
KernelEntry make_dit_entry() {
  constexpr int R0 = 2;
  KernelEntry e;
  e.fn = reinterpret_cast<const void*>(&fft_dit_kernel<Plan4096Full, FWD, MINW>);
  e.launch = &launch_dit<FWD, MINW>;
  e.n = 4096 * R0;
  e.threads = Plan4096Full::WG;
  e.tpb = 1;
  e.e = 16 * R0;
  e.radix[0] = R0;
  e.radix[1] = 4096;
  e.radix[2] = 1;
  e.lds_bytes = Plan4096Full::LDS_BYTES;
  e.xcd_log2 = 3;  // runs of 8 transforms (512 KiB) per XCD: 76.6-76.8 % against 76.2 (16), 74.7 (32), 72.1 (4)
  e.one_workgroup_per_transform = true;
  return e;
}

template <class T16, int R0>
bool mfma_big_entry_for(bool forward, int mode, KernelEntry* out) {
  // R0 = 4 holds 16 + 16 packed results and inputs per subsequence: 168 VGPRs, three workgroups per CU
  constexpr int MINW = R0 == 2 ? 4 : 3;
  // Pass-1 twiddles of the R0 sub-transforms. bf16: built in fp32 from evaluated start values and step values (8
  // mantissa bits carry no twiddle). fp16 N = 16384: read once per transform from the packed-fp16 table (settled x -> y
  // 74.3 -> 75.9 % of 8 TB/s over the fp32 form with its start table, relative L2 error 5.3e-4 -> 6.1e-4); fp16
  // N = 8192: the fp32 form with evaluated start values (78.0 % against 77.3 % for the table, error 5.3e-4). Plan 10 is
  // the other form of the two for fp16.
  constexpr bool kFp16 = std::is_same<T16, MfmaF16>::value;
  constexpr bool kHalfTable = kFp16 && R0 == 4;
  if (mode == 10 && kFp16) {
    *out = forward ? make_mfma_big_entry<T16, true, R0, MINW, !kHalfTable>() : make_mfma_big_entry<T16, false, R0, MINW, !kHalfTable>();
    return true;
  }
  if (mode != 0) return false;
  *out = forward ? make_mfma_big_entry<T16, true, R0, MINW, kHalfTable>() : make_mfma_big_entry<T16, false, R0, MINW, kHalfTable>();
  return true;
}

// Defined one per translation unit (inst_*.hip) so the instantiations compile in parallel.
// Returns false when (n, forward, mode) has no kernel for that storage type.
bool lookup_f32_a(int n, bool forward, int mode, KernelEntry* out);  // 64 ... 1024
bool lookup_f32_b(int n, bool forward, int mode, KernelEntry* out);  // 2048, 4096
bool lookup_f32_c(int n, bool forward, int mode, KernelEntry* out);  // 8192, 16384
bool lookup_f16(int n, bool forward, int mode, KernelEntry* out);
bool lookup_bf16(int n, bool forward, int mode, KernelEntry* out);

}  // namespace tfft
