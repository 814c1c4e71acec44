// Fifth streaming micro-benchmark: what a strided column sweep costs when its row segments are shorter than a
// 128-byte cache line. A two-sweep four-step plan of a long transform (N = rows x cols) walks columns of `rows`
// points; a workgroup's tile is `rows` row segments of SEG bytes, one every cols * 8 bytes. With SEG < 128 the
// 128 / SEG tiles that share every line must meet in one L2 (one XCD) while the line is resident, or the line is
// fetched once per tile.
//
// Every variant copies a [rows][cols] matrix of 8-byte points tile by tile (all loads of a workgroup, then all its
// stores: the shape of a transform kernel), nontemporal 16-byte accesses, and reports (read + write) / time.
//   order 0: tile = launch order (neighbouring tiles land on different XCDs: workgroup b runs on XCD b mod 8)
//   order 1: the 128 / SEG tiles of a line group are launched 8 apart (same XCD, consecutive slots)
// Run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE for the bytes that cross the L2 boundary.
//
// Usage: membench5 [--lg 26] [--runs 10] [--only NAME]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>

#define CHECK_HIP(call)                                                                   \
  do {                                                                                    \
    hipError_t err_ = (call);                                                             \
    if (err_ != hipSuccess) {                                                             \
      std::fprintf(stderr, "HIP error %s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(err_)); \
      std::exit(EXIT_FAILURE);                                                            \
    }                                                                                     \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));

// ROWS x SEG bytes per workgroup of WG threads; K = loads per thread
template <int ROWS, int SEG, int WG, int ORDER>
__global__ __launch_bounds__(WG) void tile_copy(const f32x4* __restrict__ in, f32x4* __restrict__ out, long long cols) {
  constexpr int LPR = SEG / 16;             // lanes per row segment
  constexpr int RPS = WG / LPR;             // rows per step of the workgroup
  constexpr int K = ROWS / RPS;             // 16-byte accesses per thread
  constexpr int G = SEG >= 128 ? 1 : 128 / SEG;  // tiles per line group
  static_assert(ROWS % RPS == 0 && K >= 1, "tile must be whole steps");
  long long tile = blockIdx.x;
  if constexpr (ORDER == 1 && G > 1) {
    // launch index b = 8 * slot + xcd; the G members of a line group take consecutive slots of one XCD
    const long long ntiles = gridDim.x;
    const long long span = 8ll * G;
    if (tile < (ntiles / span) * span) {
      const long long slot = tile >> 3, xcd = tile & 7;
      const long long group = (slot / G) * 8 + xcd, member = slot % G;
      tile = group * G + member;
    }
  }
  const int lane_in_row = threadIdx.x % LPR, row0 = threadIdx.x / LPR;
  const long long row_pitch = cols / 2;  // in 16-byte units
  const long long base = tile * LPR + lane_in_row;
  f32x4 v[K];
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = __builtin_nontemporal_load(in + base + (long long)(row0 + k * RPS) * row_pitch);
#pragma unroll
  for (int k = 0; k < K; ++k) __builtin_nontemporal_store(v[k], out + base + (long long)(row0 + k * RPS) * row_pitch);
}

template <int ROWS, int SEG, int WG, int ORDER>
static void run(const char* name, const char* only, int lg, int runs, const f32x4* in, f32x4* out) {
  if (only && std::strcmp(only, name) != 0) return;
  const long long n = 1ll << lg, cols = n / ROWS;
  const long long ntiles = cols * 8 / SEG;
  hipEvent_t a, b;
  CHECK_HIP(hipEventCreate(&a));
  CHECK_HIP(hipEventCreate(&b));
  float best = 1e9f, sum = 0.f;
  for (int r = 0; r < runs + 1; ++r) {
    CHECK_HIP(hipEventRecord(a));
    tile_copy<ROWS, SEG, WG, ORDER><<<dim3((unsigned)ntiles), dim3(WG)>>>(in, out, cols);
    CHECK_HIP(hipEventRecord(b));
    CHECK_HIP(hipEventSynchronize(b));
    float ms;
    CHECK_HIP(hipEventElapsedTime(&ms, a, b));
    if (r > 0) {
      sum += ms;
      best = ms < best ? ms : best;
    }
  }
  CHECK_HIP(hipGetLastError());
  const double bytes = 16.0 * (double)n;
  std::printf("%-22s rows=%5d seg=%3d B wg=%4d k=%2d order=%d tiles=%8lld  mean %.4f ms = %5.1f %% of 8 TB/s   best %5.1f %%\n",
              name, ROWS, SEG, WG, ROWS * SEG / 16 / WG, ORDER, ntiles, sum / runs, bytes / (sum / runs * 1e-3) / 8e12 * 100,
              bytes / (best * 1e-3) / 8e12 * 100);
}

int main(int argc, char** argv) {
  int lg = 26, runs = 10;
  const char* only = nullptr;
  for (int i = 1; i < argc; ++i) {
    if (!std::strcmp(argv[i], "--lg") && i + 1 < argc) lg = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--runs") && i + 1 < argc) runs = std::atoi(argv[++i]);
    else if (!std::strcmp(argv[i], "--only") && i + 1 < argc) only = argv[++i];
  }
  const size_t bytes = 8ull << lg;
  f32x4 *in, *out;
  CHECK_HIP(hipMalloc(&in, bytes));
  CHECK_HIP(hipMalloc(&out, bytes));
  CHECK_HIP(hipMemset(in, 1, bytes));
  CHECK_HIP(hipMemset(out, 0, bytes));
  // 2048-row tiles (a 2^22-point transform as 2048 x 2048, or longer rows): 128, 64, 32-byte segments
  run<2048, 128, 512, 0>("r2048_s128", only, lg, runs, in, out);
  run<2048, 64, 512, 0>("r2048_s64_o0", only, lg, runs, in, out);
  run<2048, 64, 512, 1>("r2048_s64_o1", only, lg, runs, in, out);
  run<2048, 64, 256, 0>("r2048_s64_w256_o0", only, lg, runs, in, out);
  run<2048, 64, 256, 1>("r2048_s64_w256_o1", only, lg, runs, in, out);
  run<2048, 32, 256, 0>("r2048_s32_o0", only, lg, runs, in, out);
  run<2048, 32, 256, 1>("r2048_s32_o1", only, lg, runs, in, out);
  // 4096- and 8192-row tiles (2^26 as 4096 x 16384 or 8192 x 8192)
  run<4096, 64, 512, 0>("r4096_s64_o0", only, lg, runs, in, out);
  run<4096, 64, 512, 1>("r4096_s64_o1", only, lg, runs, in, out);
  run<4096, 32, 512, 0>("r4096_s32_o0", only, lg, runs, in, out);
  run<4096, 32, 512, 1>("r4096_s32_o1", only, lg, runs, in, out);
  run<8192, 32, 512, 0>("r8192_s32_o0", only, lg, runs, in, out);
  run<8192, 32, 512, 1>("r8192_s32_o1", only, lg, runs, in, out);
  run<8192, 16, 512, 0>("r8192_s16_o0", only, lg, runs, in, out);
  run<8192, 16, 512, 1>("r8192_s16_o1", only, lg, runs, in, out);
  // today's sweeps for comparison: 256-row tiles of 256-byte segments, 1024-row tiles of 128-byte segments
  run<256, 256, 256, 0>("r256_s256", only, lg, runs, in, out);
  run<1024, 128, 512, 0>("r1024_s128", only, lg, runs, in, out);
  return 0;
}
