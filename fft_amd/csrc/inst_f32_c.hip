// fp32 storage, N = 8192 and 16384.
#include "inst_common.hpp"
namespace tfft {
bool lookup_f32_c(int n, bool forward, int mode, KernelEntry* out) {
  switch (n) {
    case 8192: return entry_for<Plan8192, StoreF32, false, Plan8192Alt>(forward, mode, out);
    case 16384: return entry_for<Plan16384, StoreF32, true, Plan16384Alt>(forward, mode, out);
    default: return false;
  }
}
}  // namespace tfft
