// Second streaming micro-benchmark: what limits a one-shot "load a chunk, store a chunk" kernel
// below the plain float4 copy? Sweeps bytes per workgroup, workgroups per CU (via an LDS
// reservation) and workgroup width, all with nontemporal accesses and grid = number of chunks.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <functional>
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
__device__ __forceinline__ f32x4 ldnt(const f32x4* p) { return __builtin_nontemporal_load(p); }
__device__ __forceinline__ void stnt(f32x4* p, f32x4 v) { __builtin_nontemporal_store(v, p); }

extern __shared__ float dyn_lds[];

// one workgroup = one contiguous chunk of WG*U float4
template <int WG, int U, int SPIN>
__global__ __launch_bounds__(WG) void copy_oneshot(const f32x4* in, f32x4* out, int touch_lds) {
  const size_t c = blockIdx.x;
  const f32x4* src = in + c * (WG * U) + threadIdx.x;
  f32x4* dst = out + c * (WG * U) + threadIdx.x;
  f32x4 v[U];
#pragma unroll
  for (int u = 0; u < U; ++u) v[u] = ldnt(src + u * WG);
  if (touch_lds) dyn_lds[threadIdx.x] = v[0].x;  // keeps the LDS reservation alive
  if (SPIN > 0) {  // fake compute between load and store: SPIN dependent FMAs per value
#pragma unroll 1
    for (int s = 0; s < SPIN; ++s) {
#pragma unroll
      for (int u = 0; u < U; ++u) v[u] = v[u] * 1.0000001f + 1e-9f;
    }
  }
#pragma unroll
  for (int u = 0; u < U; ++u) stnt(dst + u * WG, v[u]);
}

template <int U>
__global__ __launch_bounds__(256) void read_oneshot(const f32x4* in, f32x4* out) {
  const f32x4* src = in + (size_t)blockIdx.x * (256 * U) + threadIdx.x;
  f32x4 acc = {0, 0, 0, 0};
#pragma unroll
  for (int u = 0; u < U; ++u) acc += ldnt(src + u * 256);
  if (acc.x == 1.2345f) out[0] = acc;
}
template <int U>
__global__ __launch_bounds__(256) void write_oneshot(f32x4* out) {
  f32x4* dst = out + (size_t)blockIdx.x * (256 * U) + threadIdx.x;
  const f32x4 v = {1, 2, 3, 4};
#pragma unroll
  for (int u = 0; u < U; ++u) stnt(dst + u * 256, v);
}

static float time_ms(int runs, const std::function<void()>& fn) {
  hipEvent_t a, b;
  CHECK_HIP(hipEventCreate(&a));
  CHECK_HIP(hipEventCreate(&b));
  fn();
  CHECK_HIP(hipDeviceSynchronize());
  CHECK_HIP(hipEventRecord(a, 0));
  for (int i = 0; i < runs; ++i) fn();
  CHECK_HIP(hipEventRecord(b, 0));
  CHECK_HIP(hipEventSynchronize(b));
  float ms;
  CHECK_HIP(hipEventElapsedTime(&ms, a, b));
  return ms / runs;
}

template <int WG, int U, int SPIN>
void bench(const char* tag, f32x4* a, f32x4* dst, size_t bytes, int lds, int runs) {
  const size_t nchunks = bytes / ((size_t)WG * U * 16);
  if (lds > 65536) CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&copy_oneshot<WG, U, SPIN>),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  const float ms = time_ms(runs, [&] { copy_oneshot<WG, U, SPIN><<<(unsigned)nchunks, WG, lds>>>(a, dst, lds > 0); });
  const double gbs = 2.0 * bytes / ms * 1e-6;
  std::printf("%-10s WG=%4d U=%2d (%3d KiB/wg) lds=%6d spin=%3d : %8.4f ms %8.1f GB/s %5.1f%%\n", tag, WG, U,
              WG * U * 16 / 1024, lds, SPIN, ms, gbs, gbs / 80.0);
}

int main(int argc, char** argv) {
  size_t mib = 2048;
  int runs = 10;
  for (int i = 1; i + 1 < argc; i += 2) {
    if (std::string(argv[i]) == "--mib") mib = std::stoull(argv[i + 1]);
    if (std::string(argv[i]) == "--runs") runs = std::stoi(argv[i + 1]);
  }
  const size_t bytes = mib << 20;
  f32x4 *a, *b;
  CHECK_HIP(hipMalloc(&a, bytes));
  CHECK_HIP(hipMalloc(&b, bytes));
  CHECK_HIP(hipMemset(a, 1, bytes));
  CHECK_HIP(hipMemset(b, 2, bytes));
  for (int oop = 0; oop < 2; ++oop) {
    f32x4* dst = oop ? b : a;
    const char* tag = oop ? "outofplace" : "inplace";
    bench<256, 1, 0>(tag, a, dst, bytes, 0, runs);
    bench<256, 2, 0>(tag, a, dst, bytes, 0, runs);
    bench<256, 4, 0>(tag, a, dst, bytes, 0, runs);
    bench<256, 8, 0>(tag, a, dst, bytes, 0, runs);
    bench<256, 16, 0>(tag, a, dst, bytes, 0, runs);
    bench<512, 4, 0>(tag, a, dst, bytes, 0, runs);
    bench<1024, 2, 0>(tag, a, dst, bytes, 0, runs);
    bench<1024, 1, 0>(tag, a, dst, bytes, 0, runs);
    bench<64, 8, 0>(tag, a, dst, bytes, 0, runs);
    bench<64, 32, 0>(tag, a, dst, bytes, 0, runs);
    for (int lds : {20000, 32000, 40000, 53000, 80000, 160000}) bench<256, 8, 0>(tag, a, dst, bytes, lds, runs);
    for (int lds : {20000, 40000, 80000}) bench<256, 1, 0>(tag, a, dst, bytes, lds, runs);
    bench<256, 8, 8>(tag, a, dst, bytes, 35000, runs);
    bench<256, 8, 32>(tag, a, dst, bytes, 35000, runs);
    bench<256, 8, 64>(tag, a, dst, bytes, 35000, runs);
    bench<256, 8, 128>(tag, a, dst, bytes, 35000, runs);
    bench<256, 8, 32>(tag, a, dst, bytes, 0, runs);
    bench<256, 8, 128>(tag, a, dst, bytes, 0, runs);
  }
  {
    const size_t n1 = bytes / 4096, n8 = bytes / 32768;
    float ms = time_ms(runs, [&] { read_oneshot<1><<<(unsigned)n1, 256>>>(a, b); });
    std::printf("read oneshot U=1  : %8.4f ms %8.1f GB/s\n", ms, bytes / ms * 1e-6);
    ms = time_ms(runs, [&] { read_oneshot<8><<<(unsigned)n8, 256>>>(a, b); });
    std::printf("read oneshot U=8  : %8.4f ms %8.1f GB/s\n", ms, bytes / ms * 1e-6);
    ms = time_ms(runs, [&] { write_oneshot<1><<<(unsigned)n1, 256>>>(b); });
    std::printf("write oneshot U=1 : %8.4f ms %8.1f GB/s\n", ms, bytes / ms * 1e-6);
    ms = time_ms(runs, [&] { write_oneshot<8><<<(unsigned)n8, 256>>>(b); });
    std::printf("write oneshot U=8 : %8.4f ms %8.1f GB/s\n", ms, bytes / ms * 1e-6);
  }
  return 0;
}
