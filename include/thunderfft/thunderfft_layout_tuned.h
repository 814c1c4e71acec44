// thunderfft::bench_layout<T, N, BPB>::{L_in, L_out}: the shared-memory layouts the bench and the samples stage
// their transforms through. Same include path and names as the reference
// (thunderfft/include/thunderfft/thunderfft_layout_tuned.h:6-97, where every (T, N) carries hand-tuned padding for
// its tensor-core fragment maps). The register maps of this build read and write an unpadded natural-order array
// without bank conflicts, so every entry is the natural layout; the template itself lives with the other
// device-level declarations.
#pragma once
#include "thunderfft_device.hpp"
