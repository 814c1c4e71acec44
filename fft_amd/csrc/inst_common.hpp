// Helper for the inst_*.hip translation units: maps (n, forward, mode) to a KernelEntry for
// one storage type. Ablation modes (load/store only, no LDS exchange) exist for forward only.
#pragma once
#include "plans.hpp"

namespace tfft {

template <class P, class IO, bool WITH_MODES>
bool entry_for(bool forward, int mode, KernelEntry* out) {
  if (mode == kModeFFT) {
    *out = forward ? make_entry<P, IO, true, kModeFFT>() : make_entry<P, IO, false, kModeFFT>();
    return true;
  }
  if constexpr (WITH_MODES) {
    if (!forward) return false;
    if (mode == kModeCopy) { *out = make_entry<P, IO, true, kModeCopy>(); return true; }
    if (mode == kModeNoLds) { *out = make_entry<P, IO, true, kModeNoLds>(); return true; }
  }
  return false;
}

}  // namespace tfft
