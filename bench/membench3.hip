// Third streaming micro-benchmark: address-window and issue-shape sensitivity of one-shot copies.
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

// U=1 copy of 4 KiB blocks; block b handles chunk perm(b): consecutive workgroups are spread
// over `spread` regions of the buffer (spread = 1: linear sweep).
__global__ __launch_bounds__(256) void copy_spread(const f32x4* in, f32x4* out, unsigned nblocks, unsigned spread) {
  const unsigned b = blockIdx.x;
  const unsigned per = nblocks / spread;
  const size_t c = (size_t)(b % spread) * per + b / spread;
  stnt(out + c * 256 + threadIdx.x, ldnt(in + c * 256 + threadIdx.x));
}

// 32 KiB per workgroup, 8 loads per thread; SHAPE 0: thread's loads 4 KiB apart (wg sweeps the
// chunk 8 times); SHAPE 1: each wave owns a contiguous 8 KiB; SHAPE 2: two dependent halves
template <int SHAPE, bool NTLD, bool NTST>
__global__ __launch_bounds__(256) void copy32k(const f32x4* in, f32x4* out, int touch) {
  const size_t c = blockIdx.x;
  const int tid = threadIdx.x, w = tid >> 6, l = tid & 63;
  const int base = (SHAPE == 1) ? w * 512 + l : tid;
  const int step = (SHAPE == 1) ? 64 : 256;
  const f32x4* src = in + c * 2048 + base;
  f32x4* dst = out + c * 2048 + base;
  f32x4 v[8];
  if (SHAPE == 2) {
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = NTLD ? ldnt(src + u * step) : src[u * step];
    float s = v[0].x + v[1].x + v[2].x + v[3].x;
    const int extra = (s == 123.456f) ? 1 : 0;  // forces the first half to land before the second issues
#pragma unroll
    for (int u = 4; u < 8; ++u) v[u] = NTLD ? ldnt(src + u * step + extra) : src[u * step + extra];
  } else {
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = NTLD ? ldnt(src + u * step) : src[u * step];
  }
  if (touch) dyn_lds[tid] = v[0].x;
#pragma unroll
  for (int u = 0; u < 8; ++u) {
    if (NTST) stnt(dst + u * step, v[u]);
    else dst[u * step] = v[u];
  }
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

int main() {
  const size_t bytes = 2048ull << 20;
  const int runs = 10;
  f32x4 *a, *b;
  CHECK_HIP(hipMalloc(&a, bytes));
  CHECK_HIP(hipMalloc(&b, bytes));
  CHECK_HIP(hipMemset(a, 1, bytes));
  CHECK_HIP(hipMemset(b, 2, bytes));
  auto rep = [&](const char* name, float ms) {
    std::printf("%-52s %8.4f ms %8.1f GB/s %5.1f%%\n", name, ms, 2.0 * bytes / ms * 1e-6, 2.0 * bytes / ms * 1e-6 / 80.0);
  };
  const unsigned nb = (unsigned)(bytes / 4096);
  for (int oop = 0; oop < 2; ++oop) {
    f32x4* dst = oop ? b : a;
    std::printf("---- %s\n", oop ? "out of place" : "in place");
    for (unsigned spread : {1u, 2u, 4u, 8u, 16u, 64u, 256u, 1024u, 4096u}) {
      char nm[80];
      std::snprintf(nm, sizeof nm, "U=1 4KiB blocks, %u interleaved regions", spread);
      rep(nm, time_ms(runs, [&] { copy_spread<<<nb, 256>>>(a, dst, nb, spread); }));
    }
    const unsigned nc = (unsigned)(bytes / 32768);
    rep("32K shape0 (4KiB stride) nt/nt", time_ms(runs, [&] { copy32k<0, true, true><<<nc, 256>>>(a, dst, 0); }));
    rep("32K shape1 (wave-contiguous 8KiB) nt/nt", time_ms(runs, [&] { copy32k<1, true, true><<<nc, 256>>>(a, dst, 0); }));
    rep("32K shape2 (two dependent halves) nt/nt", time_ms(runs, [&] { copy32k<2, true, true><<<nc, 256>>>(a, dst, 0); }));
    rep("32K shape0 ntld/plain st", time_ms(runs, [&] { copy32k<0, true, false><<<nc, 256>>>(a, dst, 0); }));
    rep("32K shape0 plain ld/nt st", time_ms(runs, [&] { copy32k<0, false, true><<<nc, 256>>>(a, dst, 0); }));
    for (int lds : {35000, 53000, 80000, 160000}) {
      if (lds > 65536) {
        CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&copy32k<0, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&copy32k<1, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
        CHECK_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(&copy32k<2, true, true>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
      }
      char nm[80];
      std::snprintf(nm, sizeof nm, "32K shape0 nt/nt lds=%d", lds);
      rep(nm, time_ms(runs, [&] { copy32k<0, true, true><<<nc, 256, lds>>>(a, dst, 1); }));
      std::snprintf(nm, sizeof nm, "32K shape1 nt/nt lds=%d", lds);
      rep(nm, time_ms(runs, [&] { copy32k<1, true, true><<<nc, 256, lds>>>(a, dst, 1); }));
      std::snprintf(nm, sizeof nm, "32K shape2 nt/nt lds=%d", lds);
      rep(nm, time_ms(runs, [&] { copy32k<2, true, true><<<nc, 256, lds>>>(a, dst, 1); }));
    }
  }
  return 0;
}
