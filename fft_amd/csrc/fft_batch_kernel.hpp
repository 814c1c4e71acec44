// The batched kernel: `batch` transforms of one length, TPB per workgroup chunk, on the block transform of
// fft_block.hpp. Replaces the reference's ThunderFFT_global_kernel
// (thunderfft/include/thunderfft/detail/kernel/global_kernel.cuh:16-51: gmem -> smem -> reg -> FFT -> smem -> gmem)
// with global memory -> registers -> BlockFFT::transform -> global memory.
//
//   launch  : one workgroup per chunk by default (the dispatcher hands chunks out in address order as workgroups
//             retire); a smaller grid walks chunks blockIdx.x, blockIdx.x + gridDim.x, ... (resident form).
//             The two are separate instantiations (LOOP): as the single trip of the loop, a one-shot workgroup's rows
//             shared registers with the loop-carried ones, and the register allocator made room by waiting for the
//             first rows to arrive and moving them before it requested the last ones (N = 16384: two of a
//             thread's sixteen loads left one memory round trip late). The straight-line form has every load in
//             flight before the first wait.
//   loads   : the first chunk's rows are requested before anything else is set up: a workgroup's life starts
//             with that HBM round trip (worth 6-8 %)
//   PREFETCH plans request the rows of the next chunk as soon as the current rows are unpacked, so they land
//             under the butterflies (E/2 more live 16-byte registers); measured slower than one workgroup per
//             chunk and kept for that comparison
//   MODE    : kModeFFT, or the measurement builds kModeCopy (loads and stores only) and kModeNoLds (no LDS
//             exchange), whose output is meaningless
#pragma once
#include "fft_block.hpp"

namespace tfft {

template <class P, class IO, bool FWD, int MODE, bool LOOP>
__global__ __launch_bounds__(P::WG, P::MINW) void fft_lds_kernel(const typename IO::raw* in, typename IO::raw* out,
                                                                 const float2* __restrict__ tw, long long batch,
                                                                 long long nchunks, int xcd_log2) {
  using raw = typename IO::raw;
  constexpr int E = P::E, NRAW = P::NRAW, TPB = P::TPB;
  __shared__ __attribute__((aligned(16))) char lds_raw[P::LDS_BYTES];

  long long unit = blockIdx.x;  // position in the launch order; xcd_chunk turns it into the chunk
  if (unit >= nchunks) return;
  const BlockFFT<P> f(tw, lds_raw);
  constexpr long long chunk_raw = (long long)P::TOTAL / 2;  // raw elements (two points) per chunk
  auto valid_in = [&](long long chunk) {
    const long long left = batch - chunk * TPB;
    return (int)(left < TPB ? left : TPB);
  };
  raw pre[NRAW];
  if constexpr (!LOOP) {
    // one workgroup per chunk (the launcher picks this instantiation when the grid covers every chunk)
    const long long chunk = xcd_chunk(unit, nchunks, xcd_log2);
    f.template load<IO>(in + chunk * chunk_raw, valid_in(chunk), pre);
    c32 v[E];
    f.template unpack<IO>(pre, v);
    if constexpr (MODE != kModeCopy) f.template transform<FWD, MODE == kModeFFT>(v, false);
    f.template store<IO>(out + chunk * chunk_raw, valid_in(chunk), v);
  } else {
    {
      const long long chunk = xcd_chunk(unit, nchunks, xcd_log2);
      f.template load<IO>(in + chunk * chunk_raw, valid_in(chunk), pre);
    }
    for (;;) {
      const long long chunk = xcd_chunk(unit, nchunks, xcd_log2);
      const long long next = unit + gridDim.x;
      const bool more = next < nchunks;  // uniform: every thread of the workgroup walks the same chunks
      c32 v[E];
      f.template unpack<IO>(pre, v);
      if constexpr (P::PREFETCH) {
        if (more) {
          const long long nc = xcd_chunk(next, nchunks, xcd_log2);
          f.template load<IO>(in + nc * chunk_raw, valid_in(nc), pre);
        }
      }
      if constexpr (MODE != kModeCopy) f.template transform<FWD, MODE == kModeFFT>(v, false);
      f.template store<IO>(out + chunk * chunk_raw, valid_in(chunk), v);
      if (!more) break;
      // the exchange buffer is reused by the next chunk of this workgroup
      if constexpr (MODE == kModeFFT) __syncthreads();
      unit = next;
      if constexpr (!P::PREFETCH) {
        const long long nc = xcd_chunk(unit, nchunks, xcd_log2);
        f.template load<IO>(in + nc * chunk_raw, valid_in(nc), pre);
      }
    }
  }
}

}  // namespace tfft
