// Instantiations of the strided tile kernels (fp32).
#include "tile_plans.hpp"

namespace tfft {
namespace {

template <class P, bool FWD, int IOMODE>
void launch_tile(const TileArgs& args, hipStream_t stream) {
  fft_tile_kernel<P, FWD, IOMODE><<<dim3((unsigned)args.ntiles), dim3(P::WG), 0, stream>>>(args);
}

template <class P>
bool entry(bool forward, int iomode, TileEntry* out) {
  constexpr int kSweep = kTileInCol | kTileOutCol | kTileTwiddle;
  constexpr int kLast = kTileOutCol;
  out->n = P::N;
  out->tpb = P::TPB;
  out->threads = P::WG;
  out->lds_bytes = P::LDS_BYTES;
  if (iomode == kSweep) {
    out->launch = forward ? &launch_tile<P, true, kSweep> : &launch_tile<P, false, kSweep>;
    return true;
  }
  if (iomode == kLast) {
    out->launch = forward ? &launch_tile<P, true, kLast> : &launch_tile<P, false, kLast>;
    return true;
  }
  return false;
}

}  // namespace

bool lookup_tile(int n, bool forward, int iomode, bool wide, TileEntry* out) {
  if (wide && n == 256) return entry<Tile256Wide>(forward, iomode, out);
  if (wide && n == 512) return entry<Tile512Wide>(forward, iomode, out);
  switch (n) {
    case 64: return entry<Tile64>(forward, iomode, out);
    case 128: return entry<Tile128>(forward, iomode, out);
    case 256: return entry<Tile256>(forward, iomode, out);
    case 512: return entry<Tile512>(forward, iomode, out);
    case 1024: return entry<Tile1024>(forward, iomode, out);
    default: return false;
  }
}

}  // namespace tfft
