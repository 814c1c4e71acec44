// fp32 storage, N = 2048 and 4096.
#include "inst_common.hpp"
namespace tfft {
bool lookup_f32_b(int n, bool forward, int mode, KernelEntry* out) {
  switch (n) {
    case 2048: return entry_for<Plan2048, StoreF32, false, Plan2048Alt>(forward, mode, out);
    case 4096: return entry_for<Plan4096, StoreF32, true, Plan4096Alt, Plan4096E32F>(forward, mode, out);
    default: return false;
  }
}
}  // namespace tfft
