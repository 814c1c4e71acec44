// HBM streaming micro-benchmark: what copy rate do the access patterns of the FFT kernels reach
// on this device, next to the plain float4 grid-stride copy? Each variant moves `bytes` in and
// `bytes` out; the table reports (read + write) / time.
//
// Usage: membench [--mib 2048] [--runs 10]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <functional>
#include <string>
#include <vector>

#define CHECK_HIP(call)                                                                   \
  do {                                                                                    \
    hipError_t err_ = (call);                                                             \
    if (err_ != hipSuccess) {                                                             \
      std::fprintf(stderr, "HIP error %s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(err_)); \
      std::exit(EXIT_FAILURE);                                                            \
    }                                                                                     \
  } while (0)

typedef float f32x4 __attribute__((ext_vector_type(4)));
template <bool NT>
__device__ __forceinline__ float4 ld(const float4* p) {
  if constexpr (NT) {
    const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
  }
  return *p;
}
template <bool NT>
__device__ __forceinline__ void st(float4* p, float4 v) {
  if constexpr (NT) {
    f32x4 t = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(t, reinterpret_cast<f32x4*>(p));
  } else {
    *p = v;
  }
}

// (a) classic grid-stride copy, U loads in flight per thread
template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_gridstride(const float4* in, float4* out, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  for (; i + (U - 1) * stride < n; i += U * stride) {
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld<NT>(in + i + u * stride);
#pragma unroll
    for (int u = 0; u < U; ++u) st<NT>(out + i + u * stride, v[u]);
  }
  for (; i < n; i += stride) st<NT>(out + i, ld<NT>(in + i));
}

// (b) one workgroup per contiguous chunk of 256*U float4 (U=8: 32 KiB), chunks walked grid-stride
template <int U, bool NT>
__global__ __launch_bounds__(256) void copy_chunk(const float4* in, float4* out, size_t nchunks) {
  for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    const float4* src = in + c * (256 * U) + threadIdx.x;
    float4* dst = out + c * (256 * U) + threadIdx.x;
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = ld<NT>(src + u * 256);
#pragma unroll
    for (int u = 0; u < U; ++u) st<NT>(dst + u * 256, v[u]);
  }
}

// (c) the N=4096 FFT pattern: per wave-instruction lanes 0-31 read 512 B of row i, lanes 32-63
// 512 B of row i+8; rows are 2 KiB apart; 8 loads per thread; chunk = 32 KiB
template <bool NT, bool PREFETCH>
__global__ __launch_bounds__(256) void copy_fftpattern(const float4* in, float4* out, size_t nchunks) {
  const int tid = threadIdx.x;
  const int w = tid >> 6, l = tid & 63, p = l & 31, h = l >> 5;
  const int base = (h * 8 * 256 + 64 * w + 2 * p) >> 1;  // float4 units; row = 128 float4
  float4 v[8];
  size_t c = blockIdx.x;
  if (c >= nchunks) return;
  if (PREFETCH) {
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = ld<NT>(in + c * 2048 + base + u * 128);
  }
  for (; c < nchunks; c += gridDim.x) {
    float4 cur[8];
    if (PREFETCH) {
#pragma unroll
      for (int u = 0; u < 8; ++u) cur[u] = v[u];
      const size_t nx = c + gridDim.x;
      if (nx < nchunks) {
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = ld<NT>(in + nx * 2048 + base + u * 128);
      }
    } else {
#pragma unroll
      for (int u = 0; u < 8; ++u) cur[u] = ld<NT>(in + c * 2048 + base + u * 128);
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) st<NT>(out + c * 2048 + base + u * 128, cur[u]);
  }
}

// (d) 8-byte-per-lane variant of the FFT pattern: a wave reads 512 contiguous bytes of one row
template <bool NT>
__global__ __launch_bounds__(256) void copy_fftpattern8(const float2* in, float2* out, size_t nchunks) {
  const int tid = threadIdx.x;
  for (size_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
    float2 v[16];
#pragma unroll
    for (int u = 0; u < 16; ++u) v[u] = in[c * 4096 + tid + u * 256];
#pragma unroll
    for (int u = 0; u < 16; ++u) out[c * 4096 + tid + u * 256] = v[u];
  }
}

// read-only / write-only streams
template <int U, bool NT>
__global__ __launch_bounds__(256) void read_only(const float4* in, float4* out, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  float4 acc = make_float4(0, 0, 0, 0);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
      float4 v = ld<NT>(in + i + u * stride);
      acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
    }
  }
  if (acc.x == 1.2345f) out[0] = acc;
}
template <int U, bool NT>
__global__ __launch_bounds__(256) void write_only(const float4* in, float4* out, size_t n) {
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  const float4 v = make_float4(1, 2, 3, 4);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i + (U - 1) * stride < n; i += U * stride) {
#pragma unroll
    for (int u = 0; u < U; ++u) st<NT>(out + i + u * stride, v);
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

int main(int argc, char** argv) {
  size_t mib = 2048;
  int runs = 10;
  for (int i = 1; i + 1 < argc; i += 2) {
    if (std::string(argv[i]) == "--mib") mib = std::stoull(argv[i + 1]);
    if (std::string(argv[i]) == "--runs") runs = std::stoi(argv[i + 1]);
  }
  const size_t bytes = mib << 20, n4 = bytes / 16, nchunks = bytes / 32768;
  float4 *a, *b;
  CHECK_HIP(hipMalloc(&a, bytes));
  CHECK_HIP(hipMalloc(&b, bytes));
  CHECK_HIP(hipMemset(a, 1, bytes));
  CHECK_HIP(hipMemset(b, 2, bytes));
  hipDeviceProp_t prop;
  CHECK_HIP(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  std::printf("device: %s, %d CUs, %zu MiB per buffer\n", prop.name, cus, mib);
  auto report = [&](const char* name, double moved, float ms) {
    std::printf("%-44s %8.4f ms  %8.1f GB/s  %5.1f%% of 8 TB/s\n", name, ms, moved / ms * 1e-6, moved / ms * 1e-6 / 80.0);
  };
  const double rw = 2.0 * bytes;
  for (int oop = 1; oop >= 0; --oop) {
    float4* dst = oop ? b : a;
    std::printf("---- %s\n", oop ? "out of place" : "in place");
    for (int g : {4, 8, 16, 32}) {
      const int grid = cus * g;
      char nm[96];
      std::snprintf(nm, sizeof nm, "gridstride U=4 grid=%dxCU", g);
      report(nm, rw, time_ms(runs, [&] { copy_gridstride<4, false><<<grid, 256>>>(a, dst, n4); }));
      std::snprintf(nm, sizeof nm, "gridstride U=8 nt grid=%dxCU", g);
      report(nm, rw, time_ms(runs, [&] { copy_gridstride<8, true><<<grid, 256>>>(a, dst, n4); }));
    }
    report("gridstride U=1 grid=n/256", rw, time_ms(runs, [&] { copy_gridstride<1, false><<<(unsigned)(n4 / 256), 256>>>(a, dst, n4); }));
    report("gridstride U=1 nt grid=n/256", rw, time_ms(runs, [&] { copy_gridstride<1, true><<<(unsigned)(n4 / 256), 256>>>(a, dst, n4); }));
    for (int g : {2, 4, 8, 16}) {
      const int grid = cus * g;
      char nm[96];
      std::snprintf(nm, sizeof nm, "chunk32K grid=%dxCU", g);
      report(nm, rw, time_ms(runs, [&] { copy_chunk<8, false><<<grid, 256>>>(a, dst, nchunks); }));
      std::snprintf(nm, sizeof nm, "chunk32K nt grid=%dxCU", g);
      report(nm, rw, time_ms(runs, [&] { copy_chunk<8, true><<<grid, 256>>>(a, dst, nchunks); }));
      std::snprintf(nm, sizeof nm, "fftpattern grid=%dxCU", g);
      report(nm, rw, time_ms(runs, [&] { copy_fftpattern<false, false><<<grid, 256>>>(a, dst, nchunks); }));
      std::snprintf(nm, sizeof nm, "fftpattern nt grid=%dxCU", g);
      report(nm, rw, time_ms(runs, [&] { copy_fftpattern<true, false><<<grid, 256>>>(a, dst, nchunks); }));
      std::snprintf(nm, sizeof nm, "fftpattern prefetch grid=%dxCU", g);
      report(nm, rw, time_ms(runs, [&] { copy_fftpattern<false, true><<<grid, 256>>>(a, dst, nchunks); }));
      std::snprintf(nm, sizeof nm, "fftpattern 8B/lane grid=%dxCU", g);
      report(nm, rw, time_ms(runs, [&] { copy_fftpattern8<false><<<grid, 256>>>((const float2*)a, (float2*)dst, nchunks); }));
    }
    report("chunk32K grid=nchunks", rw, time_ms(runs, [&] { copy_chunk<8, false><<<(unsigned)nchunks, 256>>>(a, dst, nchunks); }));
    report("chunk32K nt grid=nchunks", rw, time_ms(runs, [&] { copy_chunk<8, true><<<(unsigned)nchunks, 256>>>(a, dst, nchunks); }));
    report("fftpattern grid=nchunks", rw, time_ms(runs, [&] { copy_fftpattern<false, false><<<(unsigned)nchunks, 256>>>(a, dst, nchunks); }));
    report("fftpattern nt grid=nchunks", rw, time_ms(runs, [&] { copy_fftpattern<true, false><<<(unsigned)nchunks, 256>>>(a, dst, nchunks); }));
  }
  std::printf("---- one-way streams\n");
  report("read only U=8 grid=16xCU", bytes, time_ms(runs, [&] { read_only<8, false><<<cus * 16, 256>>>(a, b, n4); }));
  report("read only U=8 nt grid=16xCU", bytes, time_ms(runs, [&] { read_only<8, true><<<cus * 16, 256>>>(a, b, n4); }));
  report("write only U=8 grid=16xCU", bytes, time_ms(runs, [&] { write_only<8, false><<<cus * 16, 256>>>(a, b, n4); }));
  report("write only U=8 nt grid=16xCU", bytes, time_ms(runs, [&] { write_only<8, true><<<cus * 16, 256>>>(a, b, n4); }));
  report("hipMemcpyAsync D2D", rw, time_ms(runs, [&] { CHECK_HIP(hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0)); }));
  return 0;
}
