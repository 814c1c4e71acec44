// Fourth streaming micro-benchmark: resident workgroups that draw 32 KiB chunks from an atomic
// ticket counter (so chunks are handed out in address order, like the hardware dispatcher does
// for one-shot workgroups) and prefetch the next chunk while the current one is "computed".
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

template <int SPIN>
__device__ __forceinline__ void fake_compute(f32x4 (&v)[8]) {
#pragma unroll 1
  for (int s = 0; s < SPIN; ++s) {
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = v[u] * 1.0000001f + 1e-9f;
  }
}

// TICKET 0: static grid-stride; 1: one counter; 2: eight counters, one per blockIdx % 8 class
template <int TICKET, int SPIN>
__global__ __launch_bounds__(256) void copy_resident(const f32x4* in, f32x4* out, unsigned nchunks, unsigned* counters) {
  const int tid = threadIdx.x;
  __shared__ unsigned s_ticket;
  auto draw = [&]() -> unsigned {
    if (TICKET == 0) return 0;
    unsigned t = 0;
    if (tid == 0) {
      if (TICKET == 1) t = atomicAdd(&counters[0], 1u);
      else t = atomicAdd(&counters[(blockIdx.x & 7) * 32], 1u) * 8 + (blockIdx.x & 7);
    }
    return t;
  };
  auto bcast = [&](unsigned t) -> unsigned {
    __syncthreads();
    if (tid == 0) s_ticket = t;
    __syncthreads();
    return s_ticket;
  };
  unsigned cur, nxt;
  if (TICKET == 0) {
    cur = blockIdx.x;
    nxt = cur + gridDim.x;
  } else {
    cur = bcast(draw());
    nxt = bcast(draw());
  }
  if (cur >= nchunks) return;
  f32x4 pre[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) pre[u] = ldnt(in + (size_t)cur * 2048 + tid + u * 256);
  for (;;) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = pre[u];
    const bool more = nxt < nchunks;
    unsigned drawn = 0;
    if (more) {
#pragma unroll
      for (int u = 0; u < 8; ++u) pre[u] = ldnt(in + (size_t)nxt * 2048 + tid + u * 256);
      if (TICKET != 0) drawn = draw();  // in flight during the compute
    }
    fake_compute<SPIN>(v);
#pragma unroll
    for (int u = 0; u < 8; ++u) stnt(out + (size_t)cur * 2048 + tid + u * 256, v[u]);
    if (!more) break;
    cur = nxt;
    nxt = (TICKET == 0) ? cur + gridDim.x : bcast(drawn);
  }
}

template <int SPIN>
__global__ __launch_bounds__(256) void copy_oneshot(const f32x4* in, f32x4* out) {
  const int tid = threadIdx.x;
  const size_t c = blockIdx.x;
  f32x4 v[8];
#pragma unroll
  for (int u = 0; u < 8; ++u) v[u] = ldnt(in + c * 2048 + tid + u * 256);
  fake_compute<SPIN>(v);
#pragma unroll
  for (int u = 0; u < 8; ++u) stnt(out + c * 2048 + tid + u * 256, v[u]);
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

template <int SPIN>
void suite(f32x4* a, f32x4* dst, size_t bytes, unsigned* counters, int cus, int runs, const char* tag) {
  const unsigned nc = (unsigned)(bytes / 32768);
  auto rep = [&](const char* name, int g, float ms) {
    std::printf("%-10s spin=%3d %-28s grid=%2dxCU %8.4f ms %8.1f GB/s %5.1f%%\n", tag, SPIN, name, g, ms,
                2.0 * bytes / ms * 1e-6, 2.0 * bytes / ms * 1e-6 / 80.0);
  };
  rep("one-shot", 0, time_ms(runs, [&] { copy_oneshot<SPIN><<<nc, 256>>>(a, dst); }));
  for (int g : {1, 2, 3, 4, 6, 8}) {
    rep("resident static", g, time_ms(runs, [&] { copy_resident<0, SPIN><<<cus * g, 256>>>(a, dst, nc, counters); }));
    rep("resident ticket x1", g, time_ms(runs, [&] {
          CHECK_HIP(hipMemsetAsync(counters, 0, 1024, 0));
          copy_resident<1, SPIN><<<cus * g, 256>>>(a, dst, nc, counters);
        }));
    rep("resident ticket x8", g, time_ms(runs, [&] {
          CHECK_HIP(hipMemsetAsync(counters, 0, 1024, 0));
          copy_resident<2, SPIN><<<cus * g, 256>>>(a, dst, nc, counters);
        }));
  }
}

int main() {
  const size_t bytes = 2048ull << 20;
  const int runs = 10;
  f32x4 *a, *b;
  unsigned* counters;
  CHECK_HIP(hipMalloc(&a, bytes));
  CHECK_HIP(hipMalloc(&b, bytes));
  CHECK_HIP(hipMalloc(&counters, 1024));
  CHECK_HIP(hipMemset(a, 1, bytes));
  CHECK_HIP(hipMemset(b, 2, bytes));
  hipDeviceProp_t prop;
  CHECK_HIP(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  for (int oop = 0; oop < 2; ++oop) {
    f32x4* dst = oop ? b : a;
    const char* tag = oop ? "outofplace" : "inplace";
    suite<0>(a, dst, bytes, counters, cus, runs, tag);
    suite<48>(a, dst, bytes, counters, cus, runs, tag);
  }
  return 0;
}
