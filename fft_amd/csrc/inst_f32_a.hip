// fp32 storage, N = 64 ... 1024.
#include "inst_common.hpp"
namespace tfft {
bool lookup_f32_a(int n, bool forward, int mode, KernelEntry* out) {
  switch (n) {
    case 64: return entry_for<Plan64, StoreF32, false, Plan64Alt>(forward, mode, out);
    case 128: return entry_for<Plan128, StoreF32, false, Plan128Alt>(forward, mode, out);
    case 256: return entry_for<Plan256, StoreF32, true, Plan256Alt>(forward, mode, out);
    case 512: return entry_for<Plan512, StoreF32, false, Plan512Alt>(forward, mode, out);
    case 1024: return entry_for<Plan1024, StoreF32, true, Plan1024Alt>(forward, mode, out);
    default: return false;
  }
}
}  // namespace tfft
