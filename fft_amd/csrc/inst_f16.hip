// fp16 storage (fp32 arithmetic), every length.
#include "inst_common.hpp"
namespace tfft {
bool lookup_f16(int n, bool forward, int mode, KernelEntry* out) {
  switch (n) {
    case 64: return entry_for<Plan64, StoreF16, false>(forward, mode, out);
    case 128: return entry_for<Plan128, StoreF16, false>(forward, mode, out);
    case 256: return entry_for<Plan256, StoreF16, false>(forward, mode, out);
    case 512: return entry_for<Plan512, StoreF16, false>(forward, mode, out);
    case 1024: return entry_for<Plan1024, StoreF16, false>(forward, mode, out);
    case 2048: return entry_for<Plan2048, StoreF16, false>(forward, mode, out);
    case 4096: return entry_for<Plan4096, StoreF16, true, Plan4096Mfma>(forward, mode, out);
    case 8192: return entry_for<Plan8192, StoreF16, false>(forward, mode, out);
    case 16384: return entry_for<Plan16384, StoreF16, false>(forward, mode, out);
    default: return false;
  }
}
}  // namespace tfft
