// bf16 storage (fp32 arithmetic), every length.
#include "inst_common.hpp"
namespace tfft {
// 16-bit storage halves the bytes per workgroup; where the A/B of the fp32 plans came out the other way
// (N = 1024: three passes of 16 points per thread; N = 8192: split exchange) the default is swapped.
bool lookup_bf16(int n, bool forward, int mode, KernelEntry* out) {
  switch (n) {
    case 64:
      // matrix cores (settled 79-80 % of 8 TB/s against 64-66 % for the VALU kernels, plans 3 and 4)
      if (mfma_small_entry_for<MfmaBF16Single, 4>(forward, mode, out)) return true;
      return entry_for<Plan64, StoreBF16, false, Plan64Alt>(forward, mode == 3 ? 0 : mode == 4 ? 3 : -1, out);
    case 128:
      // matrix cores: level with the VALU kernels (plans 3 and 4) on a settled device x -> y (79-80 %), 2 points behind
      // in place, 13 points ahead over a process's first launches (77.5-78.4 % against 63.7-65.9 %)
      if (mfma_small_entry_for<MfmaBF16Single, 8>(forward, mode, out)) return true;
      return entry_for<Plan128, StoreBF16, false, Plan128Alt>(forward, mode == 3 ? 0 : mode == 4 ? 3 : -1, out);
    case 256:
      if (mfma_entry_for<MfmaBF16Single, 1>(forward, mode, out)) return true;
      return entry_for<Plan256, StoreBF16, false, Plan256Alt>(forward, mode == 3 ? 0 : mode == 4 ? 3 : -1, out);
    case 512:
      if (mfma_entry_for<MfmaBF16Single, 2>(forward, mode, out)) return true;
      return entry_for<Plan512, StoreBF16, false, Plan512Alt>(forward, mode == 3 ? 0 : mode == 4 ? 3 : -1, out);
    case 1024:
      if (mfma_entry_for<MfmaBF16Single, 4>(forward, mode, out)) return true;
      return entry_for<Plan1024Alt, StoreBF16, false, Plan1024>(forward, mode == 3 ? 0 : mode == 4 ? 3 : -1, out);
    case 2048:
      if (mfma_entry_for<MfmaBF16Single, 8>(forward, mode, out)) return true;
      return entry_for<Plan2048, StoreBF16, false, Plan2048Alt>(forward, mode == 3 ? 0 : mode == 4 ? 3 : -1, out);
    case 4096:
      // Default: three passes on the matrix cores with bf16 operands, one workgroup per transform, pass-1
      // twiddles built incrementally (rel-L2 3.5e-3 against the 1e-2 gate). Mode 11: the kernel with the
      // lane-ordered twiddle table as a resident prefetching grid; 8: that with the DFT matrix split into bf16
      // high and low parts (2.9e-3, 6 points of bandwidth slower); 6: split matrix, one workgroup per
      // transform; 3: the fp32 VALU kernel (1.7e-3).
      if (mode == 0 || mode == 10) { *out = forward ? make_mfma_entry<MfmaBF16Single, true, true, 4, false, 16, true>() : make_mfma_entry<MfmaBF16Single, false, true, 4, false, 16, true>(); return true; }
      if (mode == 11) { *out = forward ? make_mfma_entry<MfmaBF16Single, true, true, 4, true>() : make_mfma_entry<MfmaBF16Single, false, true, 4, true>(); return true; }
      if (mode == 8) { *out = forward ? make_mfma_entry<MfmaBF16, true, true, 4, true>() : make_mfma_entry<MfmaBF16, false, true, 4, true>(); return true; }
      if (mode == 6) { *out = forward ? make_mfma_entry<MfmaBF16, true, true, 4>() : make_mfma_entry<MfmaBF16, false, true, 4>(); return true; }
      if (mode == 3) mode = 0;
      else if (mode != 1 && mode != 2) return false;
      return entry_for<Plan4096, StoreBF16, true>(forward, mode, out);
    case 8192:
      if (mfma_big_entry_for<MfmaBF16Single, 2>(forward, mode, out)) return true;
      return entry_for<Plan8192Alt, StoreBF16, false, Plan8192>(forward, mode == 3 ? 0 : mode == 4 ? 3 : -1, out);
    case 16384:
      if (mfma_big_entry_for<MfmaBF16Single, 4>(forward, mode, out)) return true;
      return entry_for<Plan16384, StoreBF16, false, Plan16384Alt>(forward, mode == 3 ? 0 : mode == 4 ? 3 : -1, out);
    default: return false;
  }
}
bool launch_repeat_bf16(int n, const void* in, void* out, const float2* tw, long long batch, unsigned repeats, unsigned io_mask,
                        hipStream_t s) {
  return launch_mfma_repeat_n<MfmaBF16Single>(n, in, out, tw, batch, repeats, io_mask, s);
}
}  // namespace tfft
