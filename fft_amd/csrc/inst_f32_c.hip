// fp32 storage, N = 8192 and 16384.
#include "inst_common.hpp"
namespace tfft {
bool lookup_f32_c(int n, bool forward, int mode, KernelEntry* out) {
  switch (n) {
    case 8192:
      // default: two 4096-point transforms per workgroup, four workgroups per CU (fft_dit_kernel.hpp; 78-79 % of
      // 8 TB/s against 72 % for the one-pass plan with 32 points per thread, which LDS limits to two workgroups per
      // CU). Plans 3 and 4: the one-pass kernels (full and split exchange); the measurement builds price the
      // one-pass kernel's access pattern.
      if (mode == kModeFFT) {
        *out = forward ? make_dit_entry<true, 4>() : make_dit_entry<false, 4>();
        return true;
      }
      if (mode == kModeAlt) mode = kModeFFT;
      else if (mode == kModeAlt2) mode = kModeAlt;
      return entry_for<Plan8192, StoreF32, true, Plan8192Alt>(forward, mode, out);
    case 16384: return entry_for<Plan16384, StoreF32, true, Plan16384Alt>(forward, mode, out);
    default: return false;
  }
}
}  // namespace tfft
