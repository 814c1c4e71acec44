// Helper for the inst_*.hip translation units: maps (n, forward, mode) to a KernelEntry for
// one storage type. Ablation modes (load/store only, no LDS exchange) exist for forward only.
#pragma once
#include "plans.hpp"

namespace tfft {

// modes 3 and 4 (kModeAlt, kModeAlt2) run the alternative plans PALT / PALT2 of the same length,
// for A/B timing.
enum : int { kModeAlt = 3, kModeAlt2 = 4 };

template <class P, class IO, bool WITH_MODES, class PALT = void, class PALT2 = void>
bool entry_for(bool forward, int mode, KernelEntry* out) {
  if (mode == kModeFFT) {
    *out = forward ? make_entry<P, IO, true, kModeFFT>() : make_entry<P, IO, false, kModeFFT>();
    return true;
  }
  if constexpr (!std::is_void_v<PALT>) {
    if (mode == kModeAlt) {
      *out = forward ? make_entry<PALT, IO, true, kModeFFT>() : make_entry<PALT, IO, false, kModeFFT>();
      return true;
    }
  }
  if constexpr (!std::is_void_v<PALT2>) {
    if (mode == kModeAlt2) {
      *out = forward ? make_entry<PALT2, IO, true, kModeFFT>() : make_entry<PALT2, IO, false, kModeFFT>();
      return true;
    }
  }
  if constexpr (WITH_MODES) {
    if (!forward) return false;
    if (mode == kModeCopy) { *out = make_entry<P, IO, true, kModeCopy>(); return true; }
    if (mode == kModeNoLds) { *out = make_entry<P, IO, true, kModeNoLds>(); return true; }
  }
  return false;
}

}  // namespace tfft
