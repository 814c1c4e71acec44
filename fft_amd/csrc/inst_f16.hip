// fp16 storage (fp32 arithmetic), every length.
#include "conv_plans.hpp"
#include "inst_common.hpp"
namespace tfft {
// 16-bit storage halves the bytes per workgroup; where the A/B of the fp32 plans came out the other way
// (N = 1024: three passes of 16 points per thread; N = 8192: split exchange) the default is swapped.
bool lookup_f16(int n, bool forward, int mode, KernelEntry* out) {
  switch (n) {
    case 64:
      // matrix cores (settled 79-80 % of 8 TB/s against 64-66 % for the VALU kernels, plans 3 and 4)
      if (mfma_small_entry_for<MfmaF16, 4>(forward, mode, out)) return true;
      return entry_for<Plan64, StoreF16, false, Plan64Alt>(forward, mode == 3 ? 0 : mode == 4 ? 3 : -1, out);
    case 128:
      // matrix cores: level with the VALU kernels (plans 3 and 4) on a settled device x -> y (79-80 %), 2 points behind
      // in place, 13 points ahead over a process's first launches (77.5-78.4 % against 63.7-65.9 %)
      if (mfma_small_entry_for<MfmaF16, 8>(forward, mode, out)) return true;
      return entry_for<Plan128, StoreF16, false, Plan128Alt>(forward, mode == 3 ? 0 : mode == 4 ? 3 : -1, out);
    case 256:
      if (mfma_entry_for<MfmaF16, 1>(forward, mode, out)) return true;
      return entry_for<Plan256, StoreF16, false, Plan256Alt>(forward, mode == 3 ? 0 : mode == 4 ? 3 : -1, out);
    case 512:
      if (mfma_entry_for<MfmaF16, 2>(forward, mode, out)) return true;
      return entry_for<Plan512, StoreF16, false, Plan512Alt>(forward, mode == 3 ? 0 : mode == 4 ? 3 : -1, out);
    case 1024:
      if (mfma_entry_for<MfmaF16, 4>(forward, mode, out)) return true;
      return entry_for<Plan1024Alt, StoreF16, false, Plan1024>(forward, mode == 3 ? 0 : mode == 4 ? 3 : -1, out);
    case 2048:
      if (mfma_entry_for<MfmaF16, 8>(forward, mode, out)) return true;
      return entry_for<Plan2048, StoreF16, false, Plan2048Alt>(forward, mode == 3 ? 0 : mode == 4 ? 3 : -1, out);
    case 4096:
      // All three passes on the matrix cores (fft_mfma_kernel.hpp). Default: one workgroup per transform, pass-1
      // twiddles built incrementally, two LDS images. 12: the same with one LDS image; 5 / 6: one workgroup per
      // transform, twiddles from the lane-ordered table, one / two LDS images; 11 / 7: that kernel as a resident
      // grid with row prefetch, two / one LDS images. Mode 3: the VALU kernel; 9: VALU kernel as a resident
      // prefetching grid.
      if (mode == 0 || mode == 10) { *out = forward ? make_mfma_entry<MfmaF16, true, true, 4, false, 16, true>() : make_mfma_entry<MfmaF16, false, true, 4, false, 16, true>(); return true; }
      if (mode == 12) { *out = forward ? make_mfma_entry<MfmaF16, true, false, 8, false, 16, true>() : make_mfma_entry<MfmaF16, false, false, 8, false, 16, true>(); return true; }
      if (mode == 11) { *out = forward ? make_mfma_entry<MfmaF16, true, true, 4, true>() : make_mfma_entry<MfmaF16, false, true, 4, true>(); return true; }
      if (mode == 5) { *out = forward ? make_mfma_entry<MfmaF16, true, false, 4>() : make_mfma_entry<MfmaF16, false, false, 4>(); return true; }
      if (mode == 6) { *out = forward ? make_mfma_entry<MfmaF16, true, true, 4>() : make_mfma_entry<MfmaF16, false, true, 4>(); return true; }
      if (mode == 7) { *out = forward ? make_mfma_entry<MfmaF16, true, false, 4, true>() : make_mfma_entry<MfmaF16, false, false, 4, true>(); return true; }
      if (mode == 3) mode = 0;
      else if (mode == 9) mode = 3;
      else if (mode != 1 && mode != 2) return false;
      return entry_for<Plan4096, StoreF16, true, Plan4096ResidentSplit>(forward, mode, out);
    case 8192:
      if (mfma_big_entry_for<MfmaF16, 2>(forward, mode, out)) return true;
      return entry_for<Plan8192Alt, StoreF16, false, Plan8192>(forward, mode == 3 ? 0 : mode == 4 ? 3 : -1, out);
    case 16384:
      if (mfma_big_entry_for<MfmaF16, 4>(forward, mode, out)) return true;
      return entry_for<Plan16384, StoreF16, false, Plan16384Alt>(forward, mode == 3 ? 0 : mode == 4 ? 3 : -1, out);
    default: return false;
  }
}
bool launch_repeat_f16(int n, const void* in, void* out, const float2* tw, long long batch, unsigned repeats, unsigned io_mask,
                        hipStream_t s) {
  return launch_mfma_repeat_n<MfmaF16>(n, in, out, tw, batch, repeats, io_mask, s);
}
void launch_conv_spectra_f16(const void* in, void* out, const float2* tw, long long count, hipStream_t s) {
  // the default kernel of fp16 N = 4096 (two LDS images, incremental pass-1 twiddles), forward, filter order out
  fft_mfma4096_kernel<MfmaF16, true, true, 4, false, 16, true, false, true>
      <<<dim3((unsigned)count), dim3(Mfma4096::WG), 0, s>>>(static_cast<const u32x4_t*>(in), static_cast<u32x4_t*>(out), tw,
                                                            count, 4);
}
}  // namespace tfft
