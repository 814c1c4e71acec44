// Tile plans of the multi-pass path (fft_tile_kernel.hpp): one per short length. Padding (PS, PW)
// from `python tools/lds_conflicts.py --tile`.
#pragma once
#include "fft_tile_kernel.hpp"

namespace tfft {

//                         N     E   R1  R2  TPB  PS PW MINW FLAGS
using Tile64 = TilePlan<64, 16, 16, 4, 128, 5, 1, 4, kFlagSplit>;
using Tile128 = TilePlan<128, 16, 16, 8, 64, 5, 1, 4, kFlagSplit>;
using Tile256 = TilePlan<256, 16, 16, 16, 32, 5, 1, 4, kFlagSplit>;
using Tile512 = TilePlan<512, 32, 16, 32, 16, 5, 1, 3, kFlagSplit>;
using Tile1024 = TilePlan<1024, 32, 32, 32, 16, 5, 1, 3, kFlagSplit>;
// Twice the transforms per tile (512-byte and 256-byte row segments instead of 256 and 128) at the price of
// 128 KiB tiles in 512-thread workgroups, two per CU: the A/B partners of Tile256 and Tile512 (plan 3).
using Tile256Wide = TilePlan<256, 32, 16, 16, 64, 5, 1, 4, kFlagSplit>;
using Tile512Wide = TilePlan<512, 32, 16, 32, 32, 5, 1, 4, kFlagSplit>;

struct TileEntry {
  void (*launch)(const TileArgs& args, hipStream_t stream);
  int n, tpb, threads, lds_bytes;
};

// iomode: kTileInCol | kTileOutCol | kTileTwiddle (a strided sweep with four-step twiddles) or
// kTileOutCol (the last sweep: contiguous rows in, transposed out)
// wide: the 128 KiB tiles where the length has them
bool lookup_tile(int n, bool forward, int iomode, bool wide, TileEntry* out);

}  // namespace tfft
