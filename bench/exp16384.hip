// Experiments on the fp32 N = 16384 kernel's launch structure (measurement tool, not part of the library):
// the library's one-pass plan as it ships, and resident forms of it whose workgroups that share a CU start half
// a period apart. Prints % of 8 TB/s per variant from interleaved, event-timed launches and checks every variant
// against the first bit for bit.
//
//   exp16384 [rounds] [variant ...]
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../fft_amd/csrc/plans.hpp"

using namespace tfft;

#define OK(call)                                                                         \
  do {                                                                                   \
    hipError_t e_ = (call);                                                              \
    if (e_ != hipSuccess) {                                                              \
      std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));     \
      std::exit(1);                                                                      \
    }                                                                                    \
  } while (0)

using P = Plan16384Split;
using IO = StoreF32;

// Resident walk of the batch; workgroups blockIdx.x >= gridDim.x / 2 (the second workgroup of every CU when the
// grid is two per CU) sleep `stagger` * 8128 cycles before their first load.
template <bool FWD, bool RAWBAR>
__global__ __launch_bounds__(P::WG, P::MINW) void resident_stagger(const IO::raw* in, IO::raw* out,
                                                                   const float2* __restrict__ tw, long long batch,
                                                                   int xcd_log2, int stagger) {
  constexpr int E = P::E, NRAW = P::NRAW;
  __shared__ __attribute__((aligned(16))) char lds_raw[P::LDS_BYTES];
  long long unit = blockIdx.x;
  if (unit >= batch) return;
  if (blockIdx.x >= gridDim.x / 2)
    for (int i = 0; i < stagger; ++i) __builtin_amdgcn_s_sleep(127);
  const BlockFFT<P> f(tw, lds_raw);
  constexpr long long chunk_raw = (long long)P::TOTAL / 2;
  IO::raw pre[NRAW];
  f.template load<IO>(in + xcd_chunk(unit, batch, xcd_log2) * chunk_raw, 1, pre);
  for (;;) {
    const long long chunk = xcd_chunk(unit, batch, xcd_log2);
    const long long next = unit + gridDim.x;
    const bool more = next < batch;
    c32 v[E];
    f.template unpack<IO>(pre, v);
    f.template transform<FWD, true>(v, false);
    f.template store<IO>(out + chunk * chunk_raw, 1, v);
    if (!more) break;
    unit = next;
    if constexpr (RAWBAR) {
      // the next rows are requested behind the stores, and the barrier that hands the exchange buffer to the next
      // transform waits for LDS traffic only: __syncthreads() would also wait for the stores to be acknowledged
      f.template load<IO>(in + xcd_chunk(unit, batch, xcd_log2) * chunk_raw, 1, pre);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    } else {
      __syncthreads();
      f.template load<IO>(in + xcd_chunk(unit, batch, xcd_log2) * chunk_raw, 1, pre);
    }
  }
}

__global__ void fill(float* p, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t stride = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += stride) {
    unsigned h = (unsigned)(i * 2654435761u) ^ (unsigned)(i >> 32);
    h ^= h >> 16;
    h *= 0x7feb352du;
    h ^= h >> 15;
    p[i] = (float)(h >> 8) * (2.0f / (1 << 24)) - 1.0f;
  }
}

struct Variant {
  std::string name;
  int kind;      // 0 = one-shot (library kernel), 1 = resident, 2 = resident with a raw barrier between transforms
  int per_cu;    // resident: workgroups per CU
  int stagger;   // resident: sleep units of the second half of the grid
  int xcd_log2;
  int out_offset_kib = 0;  // the output buffer starts this far into its allocation
};

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? std::atoi(argv[1]) : 20;
  std::vector<Variant> vs;
  for (int i = 2; i < argc; ++i) {
    // name:kind:per_cu:stagger:xcd
    Variant v{"", 0, 2, 0, 5};
    char name[64];
    if (std::sscanf(argv[i], "%63[^:]:%d:%d:%d:%d:%d", name, &v.kind, &v.per_cu, &v.stagger, &v.xcd_log2,
                    &v.out_offset_kib) < 2)
      continue;
    v.name = name;
    vs.push_back(v);
  }
  if (vs.empty()) vs = {{"oneshot", 0, 0, 0, 5}, {"resident", 1, 2, 0, 5}, {"stagger2", 1, 2, 2, 5}};
  constexpr int N = P::N;
  const long long batch = (1ll << 28) / N;
  const size_t floats = 2ull << 28;
  float *in, *out, *ref;
  OK(hipMalloc(&in, floats * 4));
  OK(hipMalloc(&out, floats * 4 + (64u << 20)));
  OK(hipMalloc(&ref, floats * 4));
  fill<<<4096, 256>>>(in, floats);
  std::vector<float2> tw(N);
  for (int k = 0; k < N; ++k) {
    const double a = -2.0 * M_PI * k / N;
    tw[k] = make_float2((float)std::cos(a), (float)std::sin(a));
  }
  float2* d_tw;
  OK(hipMalloc(&d_tw, sizeof(float2) * N));
  OK(hipMemcpy(d_tw, tw.data(), sizeof(float2) * N, hipMemcpyHostToDevice));
  hipDeviceProp_t prop;
  OK(hipGetDeviceProperties(&prop, 0));
  std::printf("in %p out %p ref %p\n", (void*)in, (void*)out, (void*)ref);
  const int cus = prop.multiProcessorCount;

  auto run = [&](const Variant& v, float* dst) {
    if (dst == out) dst = out + (size_t)v.out_offset_kib * 256;
    if (v.kind == 0) {
      fft_lds_kernel<P, IO, true, kModeFFT, false><<<dim3((unsigned)batch), dim3(P::WG)>>>(
          reinterpret_cast<const IO::raw*>(in), reinterpret_cast<IO::raw*>(dst), d_tw, batch, batch, v.xcd_log2);
    } else {
      auto k = v.kind == 1 ? resident_stagger<true, false> : resident_stagger<true, true>;
      k<<<dim3((unsigned)(v.per_cu * cus)), dim3(P::WG)>>>(
          reinterpret_cast<const IO::raw*>(in), reinterpret_cast<IO::raw*>(dst), d_tw, batch, v.xcd_log2, v.stagger);
    }
  };
  // correctness: every variant against the first
  run(vs[0], ref);
  OK(hipDeviceSynchronize());
  std::vector<float> h_ref(1 << 20), h_out(1 << 20);
  for (size_t i = 1; i < vs.size(); ++i) {
    OK(hipMemset(out, 0xff, floats * 4 + (64u << 20)));
    run(vs[i], out);
    OK(hipDeviceSynchronize());
    bool same = true;
    for (size_t off : {size_t(0), floats / 2 - (1 << 19), floats - (1 << 20)}) {
      OK(hipMemcpy(h_ref.data(), ref + off, 4 << 20, hipMemcpyDeviceToHost));
      OK(hipMemcpy(h_out.data(), out + (size_t)vs[i].out_offset_kib * 256 + off, 4 << 20, hipMemcpyDeviceToHost));
      same = same && std::memcmp(h_ref.data(), h_out.data(), 4 << 20) == 0;
    }
    std::printf("%-14s %s\n", vs[i].name.c_str(), same ? "matches" : "DIFFERS");
  }
  std::vector<double> sum(vs.size(), 0.0), best(vs.size(), 1e9);
  hipEvent_t a, b;
  OK(hipEventCreate(&a));
  OK(hipEventCreate(&b));
  for (int r = 0; r < rounds + 2; ++r)
    for (size_t i = 0; i < vs.size(); ++i) {
      OK(hipEventRecord(a));
      run(vs[i], out);
      OK(hipEventRecord(b));
      OK(hipEventSynchronize(b));
      float ms;
      OK(hipEventElapsedTime(&ms, a, b));
      if (r >= 2) {
        sum[i] += ms;
        best[i] = ms < best[i] ? ms : best[i];
      }
    }
  const double bytes = 16.0 * (1ll << 28);
  for (size_t i = 0; i < vs.size(); ++i) {
    const double mean = sum[i] / rounds;
    std::printf("%-14s kind=%d per_cu=%d stagger=%d xcd=%d  mean %.4f ms = %5.1f %%   best %.4f ms = %5.1f %%\n",
                vs[i].name.c_str(), vs[i].kind, vs[i].per_cu, vs[i].stagger, vs[i].xcd_log2, mean,
                bytes / (mean * 1e-3) / 8e12 * 100, best[i], bytes / (best[i] * 1e-3) / 8e12 * 100);
  }
  return 0;
}
