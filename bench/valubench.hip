// VALU issue-rate micro-benchmark: independent v_fma_f32 streams at 1, 2, 4 waves per SIMD.
// Prints shader cycles (s_memtime) per wave instruction and the wall-clock rate, which together
// give the sustained clock under an all-VALU load.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK_HIP(call)                                                                   \
  do {                                                                                    \
    hipError_t err_ = (call);                                                             \
    if (err_ != hipSuccess) {                                                             \
      std::fprintf(stderr, "HIP error %s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(err_)); \
      std::exit(EXIT_FAILURE);                                                            \
    }                                                                                     \
  } while (0)

constexpr int kAcc = 16;     // independent accumulators per thread
constexpr int kIters = 4096;  // loop trips; kAcc FMAs each

__global__ __launch_bounds__(256) void fma_stream(float* out, long long* cycles, float a, float b) {
  float acc[kAcc];
#pragma unroll
  for (int i = 0; i < kAcc; ++i) acc[i] = threadIdx.x * 1e-6f + i;
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int i = 0; i < kAcc; ++i) acc[i] = __builtin_fmaf(acc[i], a, b);
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < kAcc; ++i) s += acc[i];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

// The same stream with the instruction chosen by hand: MODE 0 = v_fma_f32 (one FMA per lane), MODE 1 = v_pk_fma_f32
// (two FMAs per lane on an aligned register pair). kAcc accumulators either way, so MODE 1 does twice the flops.
typedef float f32x2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ __launch_bounds__(256) void asm_stream(float* out, long long* cycles, float a, float b) {
  f32x2 acc[kAcc];
#pragma unroll
  for (int i = 0; i < kAcc; ++i) acc[i] = f32x2{threadIdx.x * 1e-6f + i, threadIdx.x * 2e-6f + i};
  const f32x2 av = {a, a}, bv = {b, b};
  const long long t0 = __builtin_amdgcn_s_memtime();
#pragma unroll 1
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int i = 0; i < kAcc; ++i) {
      if constexpr (MODE == 0) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(acc[i].x) : "v"(a), "v"(b));
      else asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(acc[i]) : "v"(av), "v"(bv));
    }
  }
  const long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < kAcc; ++i) s += acc[i].x + acc[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
}

int main() {
  hipDeviceProp_t prop;
  CHECK_HIP(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  float* d_out;
  long long* d_cyc;
  const int max_blocks = cus * 8;
  CHECK_HIP(hipMalloc(&d_out, sizeof(float) * 256 * max_blocks));
  CHECK_HIP(hipMalloc(&d_cyc, sizeof(long long) * max_blocks));
  hipEvent_t e0, e1;
  CHECK_HIP(hipEventCreate(&e0));
  CHECK_HIP(hipEventCreate(&e1));
  std::printf("CUs=%d clockRate=%d kHz\n", cus, prop.clockRate);
  for (int wps : {1, 2, 4, 8}) {  // waves per SIMD = workgroups (4 waves) per CU
    const int blocks = cus * wps;
    fma_stream<<<blocks, 256>>>(d_out, d_cyc, 1.0000001f, 1e-9f);
    CHECK_HIP(hipDeviceSynchronize());
    CHECK_HIP(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) fma_stream<<<blocks, 256>>>(d_out, d_cyc, 1.0000001f, 1e-9f);
    CHECK_HIP(hipEventRecord(e1));
    CHECK_HIP(hipDeviceSynchronize());
    float ms;
    CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
    ms /= 5;
    long long cyc;
    CHECK_HIP(hipMemcpy(&cyc, d_cyc, sizeof cyc, hipMemcpyDeviceToHost));
    const double instr_per_wave = double(kAcc) * kIters;
    const double instr_per_simd = instr_per_wave * wps;
    // s_memtime counts at a fixed 100 MHz reference on this family; report both views
    std::printf("waves/SIMD=%d  wall=%.4f ms  memtime ticks/wave=%lld  ns per instr per SIMD=%.3f  TFLOP/s=%.1f\n", wps, ms,
                cyc, ms * 1e6 / instr_per_simd, 2.0 * 64 * instr_per_wave * 4 * blocks / (ms * 1e-3) / 1e12);
  }
  for (int mode = 0; mode < 2; ++mode)
    for (int wps : {1, 2, 4}) {
      const int blocks = cus * wps;
      auto k = mode == 0 ? asm_stream<0> : asm_stream<1>;
      k<<<blocks, 256>>>(d_out, d_cyc, 1.0000001f, 1e-9f);
      CHECK_HIP(hipDeviceSynchronize());
      CHECK_HIP(hipEventRecord(e0));
      for (int r = 0; r < 5; ++r) k<<<blocks, 256>>>(d_out, d_cyc, 1.0000001f, 1e-9f);
      CHECK_HIP(hipEventRecord(e1));
      CHECK_HIP(hipDeviceSynchronize());
      float ms;
      CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
      ms /= 5;
      const double instr_per_simd = double(kAcc) * kIters * wps;
      std::printf("%s waves/SIMD=%d  wall=%.4f ms  ns per wave-instruction per SIMD=%.3f  TFLOP/s=%.1f\n",
                  mode == 0 ? "v_fma_f32   " : "v_pk_fma_f32", wps, ms, ms * 1e6 / instr_per_simd,
                  (mode + 1) * 2.0 * 64 * kAcc * kIters * 4 * blocks / (ms * 1e-3) / 1e12);
    }
  return 0;
}
