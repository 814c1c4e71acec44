// fp32 storage, N = 8192 and 16384.
#include "inst_common.hpp"
namespace tfft {
namespace {
// R0 interleaved 4096-point transforms per workgroup (fft_dit_kernel.hpp)
template <int R0, int MINW>
bool dit_entry(bool forward, KernelEntry* out) {
  *out = forward ? make_dit_entry<R0, true, MINW>() : make_dit_entry<R0, false, MINW>();
  return true;
}
}  // namespace
bool lookup_f32_c(int n, bool forward, int mode, KernelEntry* out) {
  switch (n) {
    case 8192:
      // default: two 4096-point transforms per workgroup, four workgroups per CU (76-77 % of 8 TB/s against 72 % for
      // the one-pass plan with 32 points per thread, which LDS limits to two workgroups per CU). Plans 3 and 4: the
      // one-pass kernels (full and split exchange); the measurement builds price the one-pass kernel's access pattern.
      if (mode == kModeFFT) return dit_entry<2, 4>(forward, out);
      if (mode == kModeAlt) mode = kModeFFT;
      else if (mode == kModeAlt2) mode = kModeAlt;
      return entry_for<Plan8192, StoreF32, true, Plan8192Alt>(forward, mode, out);
    case 16384:
      // plan 5: four 4096-point transforms per workgroup; 64 points per thread leave two 256-thread workgroups per
      // CU, and it runs at 50 %: kept for that comparison
      if (mode == 5) return dit_entry<4, 2>(forward, out);
      return entry_for<Plan16384, StoreF32, true, Plan16384Alt>(forward, mode, out);
    default: return false;
  }
}
}  // namespace tfft
