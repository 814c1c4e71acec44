// bf16 storage (fp32 arithmetic), every length.
#include "inst_common.hpp"
namespace tfft {
bool lookup_bf16(int n, bool forward, int mode, KernelEntry* out) {
  switch (n) {
    case 64: return entry_for<Plan64, StoreBF16, false>(forward, mode, out);
    case 128: return entry_for<Plan128, StoreBF16, false>(forward, mode, out);
    case 256: return entry_for<Plan256, StoreBF16, false>(forward, mode, out);
    case 512: return entry_for<Plan512, StoreBF16, false>(forward, mode, out);
    case 1024: return entry_for<Plan1024, StoreBF16, false>(forward, mode, out);
    case 2048: return entry_for<Plan2048, StoreBF16, false>(forward, mode, out);
    case 4096: return entry_for<Plan4096, StoreBF16, true, Plan4096Mfma, Plan4096ResidentSplit>(forward, mode, out);
    case 8192: return entry_for<Plan8192, StoreBF16, false>(forward, mode, out);
    case 16384: return entry_for<Plan16384, StoreBF16, false>(forward, mode, out);
    default: return false;
  }
}
}  // namespace tfft
