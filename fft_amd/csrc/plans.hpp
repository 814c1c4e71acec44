// The shipped plans: one per transform length. Radices, points per thread, workgroup size
// and LDS padding (PS, PW per exchange; slot = i + (i >> PS) * PW) were chosen with
// tools/lds_conflicts.py so that every ds_read_b64 of the exchanges is conflict free.
//
// This table takes the place of the reference's batch_per_block<N> / warp_per_block<N>
// (thunderfft/include/thunderfft/detail/thunderfft_impl.cuh:14-32) and of its tuned
// shared-memory layouts (thunderfft/include/thunderfft/thunderfft_layout_tuned.h:6-97).
#pragma once
#include "fft_lds_kernel.hpp"

namespace tfft {

//                      N      E   R1  R2  R3  WG   PS1 PW1 PS2 PW2
using Plan64 = Plan<64, 16, 16, 4, 1, 256, 5, 1, 0, 0>;
using Plan128 = Plan<128, 16, 16, 8, 1, 256, 5, 1, 0, 0>;
using Plan256 = Plan<256, 16, 16, 16, 1, 256, 5, 1, 0, 0>;
using Plan512 = Plan<512, 16, 16, 8, 4, 256, 5, 4, 5, 1>;
using Plan1024 = Plan<1024, 16, 16, 16, 4, 256, 4, 1, 5, 1>;
using Plan2048 = Plan<2048, 16, 16, 16, 8, 256, 4, 1, 5, 1>;
using Plan4096 = Plan<4096, 16, 16, 16, 16, 256, 4, 1, 5, 1>;
using Plan8192 = Plan<8192, 32, 32, 16, 16, 256, 4, 1, 5, 1>;
using Plan16384 = Plan<16384, 32, 32, 32, 16, 512, 9, 16, 5, 1>;

// What the host needs to launch one instantiation.
struct KernelEntry {
  const void* fn;  // __global__ function address (for the occupancy query)
  void (*launch)(const void* in, void* out, const float2* tw, long long batch, long long nchunks,
                 unsigned grid, unsigned extra_lds, hipStream_t stream);
  int n, threads, tpb, e, radix[3], lds_bytes;
};

template <class P, class IO, bool FWD, int MODE>
void launch_plan(const void* in, void* out, const float2* tw, long long batch, long long nchunks,
                 unsigned grid, unsigned extra_lds, hipStream_t stream) {
  using raw = typename IO::raw;
  // extra_lds: unused dynamic LDS that only lowers the number of co-resident workgroups per CU
  if (extra_lds)
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&fft_lds_kernel<P, IO, FWD, MODE>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)extra_lds);
  fft_lds_kernel<P, IO, FWD, MODE><<<dim3(grid), dim3(P::WG), extra_lds, stream>>>(
      static_cast<const raw*>(in), static_cast<raw*>(out), tw, batch, nchunks);
}

template <class P, class IO, bool FWD, int MODE>
KernelEntry make_entry() {
  KernelEntry e;
  e.fn = reinterpret_cast<const void*>(&fft_lds_kernel<P, IO, FWD, MODE>);
  e.launch = &launch_plan<P, IO, FWD, MODE>;
  e.n = P::N;
  e.threads = P::WG;
  e.tpb = P::TPB;
  e.e = P::E;
  e.radix[0] = P::R1;
  e.radix[1] = P::R2;
  e.radix[2] = P::R3;
  e.lds_bytes = P::LDS_ELEMS * (int)sizeof(float2);
  return e;
}

// Defined one per translation unit (inst_*.hip) so the instantiations compile in parallel.
// Returns false when (n, forward, mode) has no kernel for that storage type.
bool lookup_f32_a(int n, bool forward, int mode, KernelEntry* out);  // 64 ... 1024
bool lookup_f32_b(int n, bool forward, int mode, KernelEntry* out);  // 2048, 4096
bool lookup_f32_c(int n, bool forward, int mode, KernelEntry* out);  // 8192, 16384
bool lookup_f16(int n, bool forward, int mode, KernelEntry* out);
bool lookup_bf16(int n, bool forward, int mode, KernelEntry* out);

}  // namespace tfft
