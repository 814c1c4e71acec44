// Host-visible table of the fused FFT-convolution kernels (fft_conv.hip).
#pragma once
#include <hip/hip_runtime.h>

namespace tfft {

// What the signal buffers hold: complex in and out, real in / complex out, real in and out
// (the three kernels of the reference's sample,
// thunderfft/sample/conv1d/conv1d_fullfft_l4096_thunder.cu:148-292).
enum : int { kConvComplex = 0, kConvRealIn = 1, kConvRealInOut = 2 };

struct ConvEntry {
  // y[b][d] = IDFT_n( DFT_n(x[b][d]) * H[d] ), unnormalised; hp = H in the kernel's register order
  // resident_per_cu > 0: a resident grid of that many workgroups per CU (matrix-core kernels only; others ignore it)
  void (*launch)(const void* x, const void* hp, void* y, const float2* tw, int B, int D, int resident_per_cu,
                 hipStream_t stream);
  // hp <- h: natural-order spectra [D][n] to register order
  void (*layout)(const void* h, void* hp, int D, hipStream_t stream);
  // hp <- DFT_n(taps): transform and re-ordering in one kernel, where the plan has it (else null)
  void (*spectra)(const void* taps, void* hp, const float2* tw, int D, hipStream_t stream) = nullptr;
  int n, threads, points_per_thread, lds_bytes;
  int layout_id;  // which register order `layout` writes and `launch` reads (kConvLayout*)
};
enum : int { kConvLayoutValu = 1, kConvLayoutMfma = 2 };

// The default N = 4096 matrix-core kernel of fp16 storage with its result in the matrix-core convolution's filter
// order (inst_f16.hip): `count` forward transforms [count][4096] -> [count][4][256] 16-byte entries.
void launch_conv_spectra_f16(const void* in, void* out, const float2* tw, long long count, hipStream_t s);

// plan 0: the default kernel of (dtype, n, kind); plan 3: the fp32-VALU kernel where the default runs on the
// matrix cores (fp16 storage). The two read the permuted filters in different orders (layout_id).
bool lookup_conv(int dtype, int n, int kind, int plan, ConvEntry* out);

// Overlap-save tiles of length n, `valid` outputs kept per tile.
struct ConvTilesEntry {
  void (*launch)(const void* in, const void* hf, void* out, const float2* tw, long long input_len, long long output_len,
                 int valid, hipStream_t stream);
  int n, threads, tiles_per_block;
};
bool lookup_conv_tiles(int dtype, int n, ConvTilesEntry* out);

}  // namespace tfft
