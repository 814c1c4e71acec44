// Bench harness in the shape of the reference's sample/bench (thunderfft/sample/bench/main.cu:28-83,
// util/helper.h:122-202, baseline/baseline_cufft.cu:28-69): for every length it fills the input,
// runs the library comparator (hipFFT here, cuFFT there), validates ThunderFFT_global against it
// (max-abs complex error, separate validation and timing buffers), then times single launches with
// events: one warm-up, the mean of `runs` launches. On top of the reference's milliseconds it
// reports GFLOP/s (5 N log2 N), GB/s (one read + one write per point) and the fraction of 8 TB/s.
//
// compute_ms is the reference's second figure (util/helper.h:122-141, baseline/thunderfft_bench.h:7-55):
// a kernel that loads once, runs the register-resident transform R times and stores once is timed
// at R and 2R in-kernel repeats, compute_ms = (t(2R) - t(R)) / R -- the cost of one transform pass
// over the batch with HBM taken out. Built on the device-level API; fp32 storage only. The row
// thunderfft_s is the reference's shared-memory variant (baseline/thunderfft_bench.h:139-261): the block's
// signals stay in LDS in natural order and every repeat is smem2reg, transform, reg2smem; its e2e_ms is one
// pass gmem2smem ... smem2gmem (N <= 8192, where the staging buffer and the workspace fit in LDS).
//
// Compile-time defaults in the manner of the reference's profiling build (sample/bench/main.cu:12-26):
// -DTHUNDERFFT_PROFILE_N=1024 -DTHUNDERFFT_PROFILE_BATCH=65536 -DTHUNDERFFT_PROFILE_HALF=1 preset --n,
// --points (N x batch) and --dtype fp16; the command line still overrides them.
//
// Usage: tfft_bench [--n 256,4096] [--points 268435456] [--runs 10] [--input sine|hash]
//                   [--dtype fp32|fp16|bf16] [--inplace 0|1] [--mode 0|1|2] [--grid K] [--extra-lds BYTES] [--csv] [--ndjson]
//                   [--no-hipfft] [--inside-repeats R]   (R = 0 skips compute_ms)
//                   [--xcd L] [--copy-xcd L] [--copy-k K]  runs of 2^L consecutive chunks per XCD in the library's kernels
//                   (default: the kernels' own choice) and in the reference copy (default 8);
//                   TFFT_BENCH_ADDR=1 prints the addresses of the timed buffers
#include <hipfft/hipfft.h>

#include <cmath>
#include <cstring>
#include <sstream>
#include <string>
#include <vector>

#define THUNDERFFT_ASYNC
#include <thunderfft/thunderfft_device.hpp>

#include "stat.hpp"

namespace {

// Same counter hash as fft_amd/synth.py: uniform(-1, 1) from (counter, seed).
inline float hash_uniform(unsigned long long counter, unsigned long long seed) {
  unsigned x = (unsigned)((counter * 0x9E3779B97F4A7C15ull + seed * 0xD1B54A32D192ED03ull) >> 32);
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return (float)(x >> 8) * (2.0f / (1 << 24)) - 1.0f;
}

// Speed-of-light reference measured in the same process: a one-shot nontemporal float4 copy
// (one 16-byte load and store per thread, 4 KiB per workgroup), the fastest streaming copy found
// on MI355X. Boxes differ by several percent, so every table carries its own copy row.
typedef float copy_f32x4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void reference_copy(const copy_f32x4* in, copy_f32x4* out, int xcd_log2) {
  // same XCD-aware order as the library's kernels (runs of 2^xcd_log2 consecutive 4 KiB blocks per XCD)
  const size_t i = (size_t)tfft::xcd_chunk(blockIdx.x, gridDim.x, xcd_log2) * 256 + threadIdx.x;
  __builtin_nontemporal_store(__builtin_nontemporal_load(in + i), out + i);
}

// The reference's ThunderFFT_benchmark_reg (baseline/thunderfft_bench.h:7-55): values overflow after a
// few unnormalised passes, which does not change the instruction timing.
template <int N>
__global__ __launch_bounds__(thunderfft::block_threads<N>) void benchmark_reg_kernel(const float2* __restrict__ in,
                                                                                      float2* __restrict__ out,
                                                                                      const float2* __restrict__ tw,
                                                                                      long long batch, unsigned inside_repeats) {
  constexpr int BPB = thunderfft::batch_per_block<N>;
  __shared__ __attribute__((aligned(16))) char workspace[thunderfft::workspace_bytes<N>];
  thunderfft::DeviceFFT<N> fft(tw, workspace);
  const long long first = (long long)blockIdx.x * BPB;
  const int valid = (int)((batch - first < BPB) ? batch - first : BPB);
  float2 reg[thunderfft::elements_per_thread<N>];
  thunderfft::ThunderFFT_gmem2reg<float, N>(fft, reg, in + first * N, valid);
#pragma unroll 1
  for (unsigned i = 0; i < inside_repeats; ++i) thunderfft::ThunderFFT_kernel_reg<N, true>(fft, reg, true);
  // the result is not meaningful after repeated passes: dump the registers in thread order (one
  // coalesced 8-byte store per register) instead of the transform's natural-order store
  if (valid == BPB) {
#pragma unroll
    for (int e = 0; e < thunderfft::elements_per_thread<N>; ++e)
      out[first * N + (long long)e * thunderfft::block_threads<N> + threadIdx.x] = reg[e];
  }
}

template <int N>
void launch_benchmark_reg(float2* data, long long batch, unsigned repeats, hipStream_t s) {
  constexpr int BPB = thunderfft::batch_per_block<N>;
  const float2* tw = thunderfft::twiddle_table<float>(N);
  benchmark_reg_kernel<N><<<dim3((unsigned)((batch + BPB - 1) / BPB)), dim3(thunderfft::block_threads<N>), 0, s>>>(
      data, data, tw, batch, repeats);
}

// The reference's ThunderFFT_benchmark_smem (baseline/thunderfft_bench.h:139-182).
template <int N>
__global__ __launch_bounds__(thunderfft::block_threads<N>) void benchmark_smem_kernel(const float2* __restrict__ in,
                                                                                       float2* __restrict__ out,
                                                                                       const float2* __restrict__ tw,
                                                                                       long long batch, unsigned inside_repeats) {
  constexpr int BPB = thunderfft::batch_per_block<N>;
  __shared__ __attribute__((aligned(16))) char workspace[thunderfft::workspace_bytes<N>];
  __shared__ __attribute__((aligned(16))) float2 stage[N * BPB];
  thunderfft::DeviceFFT<N> fft(tw, workspace);
  const long long first = (long long)blockIdx.x * BPB;
  const int valid = (int)((batch - first < BPB) ? batch - first : BPB);
  float2 reg[thunderfft::elements_per_thread<N>];
  thunderfft::ThunderFFT_gmem2smem<float, N>(stage, in + first * N, valid);
  __syncthreads();
#pragma unroll 1
  for (unsigned i = 0; i < inside_repeats; ++i) {
    thunderfft::ThunderFFT_smem2reg<float, N>(fft, reg, stage);
    thunderfft::ThunderFFT_kernel_reg<N, true>(fft, reg, true);
    thunderfft::ThunderFFT_reg2smem<float, N>(fft, stage, reg);
    __syncthreads();
  }
  thunderfft::ThunderFFT_smem2gmem<float, N>(out + first * N, stage, valid);
}

template <int N>
void launch_benchmark_smem(float2* data, long long batch, unsigned repeats, hipStream_t s) {
  constexpr int BPB = thunderfft::batch_per_block<N>;
  const float2* tw = thunderfft::twiddle_table<float>(N);
  benchmark_smem_kernel<N><<<dim3((unsigned)((batch + BPB - 1) / BPB)), dim3(thunderfft::block_threads<N>), 0, s>>>(
      data, data, tw, batch, repeats);
}

bool launch_benchmark_smem_n(int n, float2* data, long long batch, unsigned repeats, hipStream_t s) {
  switch (n) {
    case 64: launch_benchmark_smem<64>(data, batch, repeats, s); return true;
    case 128: launch_benchmark_smem<128>(data, batch, repeats, s); return true;
    case 256: launch_benchmark_smem<256>(data, batch, repeats, s); return true;
    case 512: launch_benchmark_smem<512>(data, batch, repeats, s); return true;
    case 1024: launch_benchmark_smem<1024>(data, batch, repeats, s); return true;
    case 2048: launch_benchmark_smem<2048>(data, batch, repeats, s); return true;
    case 4096: launch_benchmark_smem<4096>(data, batch, repeats, s); return true;
    case 8192: launch_benchmark_smem<8192>(data, batch, repeats, s); return true;
    default: return false;
  }
}

bool launch_benchmark_reg_n(int n, float2* data, long long batch, unsigned repeats, hipStream_t s) {
  switch (n) {
    case 64: launch_benchmark_reg<64>(data, batch, repeats, s); return true;
    case 128: launch_benchmark_reg<128>(data, batch, repeats, s); return true;
    case 256: launch_benchmark_reg<256>(data, batch, repeats, s); return true;
    case 512: launch_benchmark_reg<512>(data, batch, repeats, s); return true;
    case 1024: launch_benchmark_reg<1024>(data, batch, repeats, s); return true;
    case 2048: launch_benchmark_reg<2048>(data, batch, repeats, s); return true;
    case 4096: launch_benchmark_reg<4096>(data, batch, repeats, s); return true;
    case 8192: launch_benchmark_reg<8192>(data, batch, repeats, s); return true;
    case 16384: launch_benchmark_reg<16384>(data, batch, repeats, s); return true;
    default: return false;
  }
}

#ifdef THUNDERFFT_PROFILE_N
#ifndef THUNDERFFT_PROFILE_BATCH
#define THUNDERFFT_PROFILE_BATCH 65536
#endif
#endif

// The same copy with K 16-byte loads per thread issued before its K stores: a workgroup of 4K KiB, the
// shape of a transform kernel's life (all rows in, all rows out).
template <int K>
__global__ __launch_bounds__(256) void reference_copy_k(const copy_f32x4* in, copy_f32x4* out, int xcd_log2) {
  const size_t base = (size_t)tfft::xcd_chunk(blockIdx.x, gridDim.x, xcd_log2) * (256 * K) + threadIdx.x;
  copy_f32x4 v[K];
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = __builtin_nontemporal_load(in + base + 256 * k);
#pragma unroll
  for (int k = 0; k < K; ++k) __builtin_nontemporal_store(v[k], out + base + 256 * k);
}

struct Args {
#ifdef THUNDERFFT_PROFILE_N
  std::vector<int> ns = {THUNDERFFT_PROFILE_N};
  long long points = (long long)(THUNDERFFT_PROFILE_N) * (THUNDERFFT_PROFILE_BATCH);
#else
  std::vector<int> ns = {256, 512, 1024, 2048, 4096, 8192, 16384};
  long long points = 1ll << 28;
#endif
  int runs = 10;
  std::string input = "hash";
#if defined(THUNDERFFT_PROFILE_HALF) && THUNDERFFT_PROFILE_HALF
  std::string dtype = "fp16";
#else
  std::string dtype = "fp32";
#endif
  int inplace = 1;
  int mode = 0;
  int grid = 0;
  int extra_lds = 0;
  long long grid_blocks = 0;
  int copy_k = 1;    // 16-byte loads per thread of the reference copy (1, 2, 4, 8, 16, 32: 4 ... 128 KiB per workgroup)
  int copy_xcd = 8;  // log2 of the run of consecutive 4 KiB blocks per XCD in the reference copy (1 MiB runs measured best)
  int xcd = -1;  // >= 0: log2 of the run of consecutive chunks per XCD (library option xcd_run_log2)
  long long skew = 0;  // bytes added to the output buffer's address (out of place only)
  bool csv = false, ndjson = false, hipfft = true;
  int inside_repeats = 0;
};

float time_ms(hipStream_t s, int warm, int runs, const std::function<void()>& fn) {
  hipEvent_t a, b;
  CHECK_HIP(hipEventCreate(&a));
  CHECK_HIP(hipEventCreate(&b));
  for (int i = 0; i < warm; ++i) fn();
  CHECK_HIP(hipStreamSynchronize(s));
  CHECK_HIP(hipEventRecord(a, s));
  for (int i = 0; i < runs; ++i) fn();
  CHECK_HIP(hipEventRecord(b, s));
  CHECK_HIP(hipEventSynchronize(b));
  float ms = 0;
  CHECK_HIP(hipEventElapsedTime(&ms, a, b));
  CHECK_HIP(hipEventDestroy(a));
  CHECK_HIP(hipEventDestroy(b));
  return ms / runs;
}

// Reference check_max_abs_err (util/helper.h:94-120): largest |ref - test|, inf on any non-finite.
double max_abs_err(const float2* ref, const float2* test, size_t n) {
  double m = 0;
  for (size_t i = 0; i < n; ++i) {
    const double dx = (double)ref[i].x - test[i].x, dy = (double)ref[i].y - test[i].y;
    const double e = std::sqrt(dx * dx + dy * dy);
    if (!std::isfinite(test[i].x) || !std::isfinite(test[i].y) || !std::isfinite(e)) return INFINITY;
    if (e > m) m = e;
  }
  return m;
}

void push_row(const std::string& type, int n, const std::string& radix, long long batch, double err, double ms,
              double bytes_per_point, double compute_ms = NAN) {
  const double flops = 5.0 * n * std::log2((double)n) * batch;
  const double bytes = bytes_per_point * (double)n * batch;
  stat::push({type, (unsigned)n, radix, batch, err, compute_ms, ms, flops / ms * 1e-6, bytes / ms * 1e-6,
              bytes / ms * 1e-6 / 8000.0});
}

template <typename T>
void to_storage(const std::vector<float2>& h, std::vector<vec2_t<T>>& out) {
  out.resize(h.size());
  for (size_t i = 0; i < h.size(); ++i) {
    if constexpr (std::is_same_v<T, float>) out[i] = h[i];
    else if constexpr (std::is_same_v<T, __half>) out[i] = __floats2half2_rn(h[i].x, h[i].y);
    else out[i] = __float22bfloat162_rn(h[i]);
  }
}
template <typename T>
float2 from_storage(vec2_t<T> v) {
  if constexpr (std::is_same_v<T, float>) return v;
  else if constexpr (std::is_same_v<T, __half>) return __half22float2(v);
  else return __bfloat1622float2(v);
}

template <typename T>
void run_one(const Args& a, int n) {
  const long long batch = a.points / n;
  const size_t total = (size_t)n * batch;
  // validation runs on a bounded prefix so the host-side check stays short
  const long long vbatch = std::min<long long>(batch, n >= (1 << 20) ? 1 : (1ll << 22) / n + 3);
  const size_t vtotal = (size_t)n * vbatch;
  const bool scaled = !std::is_same_v<T, float>;

  std::vector<float2> h(vtotal);
  for (size_t i = 0; i < vtotal; ++i) {
    if (a.input == "sine") {  // reference input, sample/bench/main.cu:35-41
      h[i] = make_float2(sinf(2 * (float)M_PI * (float)(i % n) / n) / sqrtf((float)n), 0.0f);
    } else {
      const float sc = scaled ? 1.0f / sqrtf((float)n) : 1.0f;
      h[i] = make_float2(hash_uniform(2 * i + 1, 0) * sc, hash_uniform(2 * i + 2, 0) * sc);
    }
  }
  std::vector<vec2_t<T>> hs;
  to_storage<T>(h, hs);
  for (size_t i = 0; i < vtotal; ++i) h[i] = from_storage<T>(hs[i]);  // what the device really holds

  hipStream_t s;
  CHECK_HIP(hipStreamCreate(&s));
  vec2_t<T>*d_val = nullptr, *d_val_out = nullptr, *d_perf = nullptr, *d_perf_out = nullptr;
  CHECK_HIP(hipMalloc(&d_val, sizeof(vec2_t<T>) * vtotal));
  CHECK_HIP(hipMalloc(&d_val_out, sizeof(vec2_t<T>) * vtotal));
  CHECK_HIP(hipMalloc(&d_perf, sizeof(vec2_t<T>) * total));
  vec2_t<T>* d_perf_out_alloc = nullptr;
  if (!a.inplace) {
    CHECK_HIP(hipMalloc(&d_perf_out_alloc, sizeof(vec2_t<T>) * total + (size_t)a.skew));
    d_perf_out = reinterpret_cast<vec2_t<T>*>(reinterpret_cast<char*>(d_perf_out_alloc) + a.skew);
    if (getenv("TFFT_BENCH_ADDR")) fprintf(stderr, "n=%d in=%p out=%p\n", n, (void*)d_perf, (void*)d_perf_out);
  }
  CHECK_HIP(hipMemcpy(d_val, hs.data(), sizeof(vec2_t<T>) * vtotal, hipMemcpyHostToDevice));
  // timing buffer: the validation pattern repeated (values do not matter for timing, but keep them finite)
  for (size_t off = 0; off < total; off += vtotal) {
    const size_t cnt = std::min(vtotal, total - off);
    CHECK_HIP(hipMemcpy(d_perf + off, hs.data(), sizeof(vec2_t<T>) * cnt, hipMemcpyHostToDevice));
  }

  // ---- comparator: hipFFT fp32 on the same (storage-rounded) input
  std::vector<float2> ref(vtotal);
  double hipfft_ms = -1;
  if (a.hipfft) {
    float2 *d_f = nullptr, *d_fo = nullptr;
    CHECK_HIP(hipMalloc(&d_f, sizeof(float2) * vtotal));
    CHECK_HIP(hipMalloc(&d_fo, sizeof(float2) * vtotal));
    CHECK_HIP(hipMemcpy(d_f, h.data(), sizeof(float2) * vtotal, hipMemcpyHostToDevice));
    hipfftHandle plan;
    if (hipfftPlan1d(&plan, n, HIPFFT_C2C, (int)vbatch) != HIPFFT_SUCCESS) { std::fprintf(stderr, "hipfftPlan1d failed\n"); std::exit(1); }
    hipfftSetStream(plan, s);
    hipfftExecC2C(plan, (hipfftComplex*)d_f, (hipfftComplex*)d_fo, HIPFFT_FORWARD);
    CHECK_HIP(hipStreamSynchronize(s));
    CHECK_HIP(hipMemcpy(ref.data(), d_fo, sizeof(float2) * vtotal, hipMemcpyDeviceToHost));
    hipfftDestroy(plan);
    CHECK_HIP(hipFree(d_f));
    CHECK_HIP(hipFree(d_fo));
    if (std::is_same_v<T, float>) {  // full-size hipFFT timing, in place like ours
      hipfftHandle big;
      if (hipfftPlan1d(&big, n, HIPFFT_C2C, (int)batch) == HIPFFT_SUCCESS) {
        hipfftSetStream(big, s);
        float2* buf = (float2*)d_perf;
        float2* dst = a.inplace ? buf : (float2*)d_perf_out;
        hipfft_ms = time_ms(s, 1, a.runs, [&] { hipfftExecC2C(big, (hipfftComplex*)buf, (hipfftComplex*)dst, HIPFFT_FORWARD); });
        hipfftDestroy(big);
        for (size_t off = 0; off < total; off += vtotal) {
          const size_t cnt = std::min(vtotal, total - off);
          CHECK_HIP(hipMemcpy(d_perf + off, hs.data(), sizeof(vec2_t<T>) * cnt, hipMemcpyHostToDevice));
        }
      }
    }
  }

  // ---- ThunderFFT: validate, then time
  thunderfft::ThunderFFTInitialize<T>(n);
  const tfft_plan_info info = thunderfft::plan_info<T>(n);
  std::ostringstream radix;
  radix << info.radix[0] << "x" << info.radix[1];
  if (info.radix[2] > 1) radix << "x" << info.radix[2];
  if (n > 16384) radix << " sweeps";

  thunderfft::check(tfft_exec_c2c(thunderfft::dtype_code<T>(), n, 1, d_val, d_val_out, vbatch, s), "validate");
  CHECK_HIP(hipStreamSynchronize(s));
  std::vector<vec2_t<T>> got(vtotal);
  CHECK_HIP(hipMemcpy(got.data(), d_val_out, sizeof(vec2_t<T>) * vtotal, hipMemcpyDeviceToHost));
  double err = NAN;
  if (a.hipfft) {
    std::vector<float2> gf(vtotal);
    for (size_t i = 0; i < vtotal; ++i) gf[i] = from_storage<T>(got[i]);
    err = max_abs_err(ref.data(), gf.data(), vtotal);
  }

  // --mode 1 / 2 time the measurement builds (loads and stores only / no LDS exchange: meaningless output),
  // every other value is a plan of the transform itself
  tfft_exec_options opt;
  tfft_exec_options_init(&opt);
  opt.plan = (a.mode >= 1 && a.mode <= 2) || a.mode == 13 || a.mode == 14 ? 0 : a.mode;
  opt.grid_blocks_per_cu = a.grid;
  opt.extra_lds = a.extra_lds;
  opt.grid_blocks = a.grid_blocks;
  opt.xcd_run_log2 = a.xcd;
  vec2_t<T>* dst = a.inplace ? d_perf : d_perf_out;
  const float ms = time_ms(s, 1, a.runs, [&] {
    // --mode 1 / 2 and 13 / 14: the measurement builds (patterns 1, 2 and 3, 4 of tfft_bench_access_pattern)
    if (a.mode == 1 || a.mode == 2 || a.mode == 13 || a.mode == 14)
      thunderfft::check(tfft_bench_access_pattern(thunderfft::dtype_code<T>(), n, 1, a.mode > 2 ? a.mode - 10 : a.mode, d_perf,
                                                  dst, batch, s, &opt), "bench");
    else
      thunderfft::check(tfft_exec_c2c_opt(thunderfft::dtype_code<T>(), n, 1, d_perf, dst, batch, s, &opt), "bench");
  });

  double compute_ms = NAN;
  if constexpr (std::is_same_v<T, float>) {
    if (a.inside_repeats > 0 && n <= 16384) {
      const unsigned R = (unsigned)a.inside_repeats;
      float2* buf = reinterpret_cast<float2*>(d_perf);
      const float t_r = time_ms(s, 1, 1, [&] { launch_benchmark_reg_n(n, buf, batch, R, s); });
      const float t_2r = time_ms(s, 1, 1, [&] { launch_benchmark_reg_n(n, buf, batch, 2 * R, s); });
      compute_ms = ((double)t_2r - (double)t_r) / (double)R;
    }
  } else {
    // 16-bit storage: the matrix-core kernels run their passes R and 2R times per tile on chip (the reference
    // times its half-precision kernels the same way, sample/bench/main.cu:52-57); one pass here is registers ->
    // three matrix-core passes -> natural-order LDS image -> registers, the counterpart of its _smem variant
    if (a.inside_repeats > 0 && n >= 256 && n <= 4096) {
      const int R = a.inside_repeats;
      const int dt = thunderfft::dtype_code<T>();
      const float t_r = time_ms(s, 1, 1, [&] {
        thunderfft::check(tfft_bench_onchip_repeats(dt, n, R, d_perf, d_perf, batch, s), "onchip repeats"); });
      const float t_2r = time_ms(s, 1, 1, [&] {
        thunderfft::check(tfft_bench_onchip_repeats(dt, n, 2 * R, d_perf, d_perf, batch, s), "onchip repeats"); });
      compute_ms = ((double)t_2r - (double)t_r) / (double)R;
    }
  }

  const double bpp = 2.0 * sizeof(vec2_t<T>);
  push_row(std::string("thunderfft(") + a.dtype + (a.mode ? ",mode" + std::to_string(a.mode) : "") + ")", n, radix.str(),
           batch, err, ms, bpp, compute_ms);
  if constexpr (std::is_same_v<T, float>) {
    if (a.inside_repeats > 0 && n <= 8192) {
      const unsigned R = (unsigned)a.inside_repeats;
      float2* buf = reinterpret_cast<float2*>(d_perf);
      const float e2e_s = time_ms(s, 1, a.runs, [&] { launch_benchmark_smem_n(n, buf, batch, 1, s); });
      const float t_r = time_ms(s, 1, 1, [&] { launch_benchmark_smem_n(n, buf, batch, R, s); });
      const float t_2r = time_ms(s, 1, 1, [&] { launch_benchmark_smem_n(n, buf, batch, 2 * R, s); });
      push_row("thunderfft_s(fp32)", n, radix.str(), batch, NAN, e2e_s, bpp, ((double)t_2r - (double)t_r) / (double)R);
    }
  }
  if (hipfft_ms > 0) push_row("hipfft(fp32)", n, "-", batch, 0.0, hipfft_ms, 16.0);
  {
    const size_t bytes = sizeof(vec2_t<T>) * total;
    copy_f32x4* cdst = reinterpret_cast<copy_f32x4*>(a.inplace ? d_perf : d_perf_out);
    const float cms = time_ms(s, 1, a.runs, [&] {
      const copy_f32x4* csrc = reinterpret_cast<const copy_f32x4*>(d_perf);
      switch (a.copy_k) {
        case 2: reference_copy_k<2><<<(unsigned)(bytes / 8192), 256, 0, s>>>(csrc, cdst, a.copy_xcd); break;
        case 4: reference_copy_k<4><<<(unsigned)(bytes / 16384), 256, 0, s>>>(csrc, cdst, a.copy_xcd); break;
        case 8: reference_copy_k<8><<<(unsigned)(bytes / 32768), 256, 0, s>>>(csrc, cdst, a.copy_xcd); break;
        case 16: reference_copy_k<16><<<(unsigned)(bytes / 65536), 256, 0, s>>>(csrc, cdst, a.copy_xcd); break;
        case 32: reference_copy_k<32><<<(unsigned)(bytes / 131072), 256, 0, s>>>(csrc, cdst, a.copy_xcd); break;
        default: reference_copy<<<(unsigned)(bytes / 4096), 256, 0, s>>>(csrc, cdst, a.copy_xcd);
      }
    });
    push_row("copy16B(reference)", n, "-", batch, 0.0, cms, bpp);
  }

  CHECK_HIP(hipFree(d_val));
  CHECK_HIP(hipFree(d_val_out));
  CHECK_HIP(hipFree(d_perf));
  if (d_perf_out_alloc) CHECK_HIP(hipFree(d_perf_out_alloc));
  CHECK_HIP(hipStreamDestroy(s));
}

}  // namespace

int main(int argc, char** argv) {
  Args a;
  for (int i = 1; i < argc; ++i) {
    const std::string k = argv[i];
    auto val = [&]() -> std::string { return (i + 1 < argc) ? argv[++i] : ""; };
    if (k == "--n") {
      a.ns.clear();
      std::stringstream ss(val());
      std::string tok;
      while (std::getline(ss, tok, ',')) a.ns.push_back(std::stoi(tok));
    } else if (k == "--points") a.points = std::stoll(val());
    else if (k == "--runs") a.runs = std::stoi(val());
    else if (k == "--input") a.input = val();
    else if (k == "--dtype") a.dtype = val();
    else if (k == "--inplace") a.inplace = std::stoi(val());
    else if (k == "--mode") a.mode = std::stoi(val());
    else if (k == "--grid") a.grid = std::stoi(val());
    else if (k == "--extra-lds") a.extra_lds = std::stoi(val());
    else if (k == "--grid-blocks") a.grid_blocks = std::stoll(val());
    else if (k == "--xcd") a.xcd = std::stoi(val());
    else if (k == "--copy-xcd") a.copy_xcd = std::stoi(val());
    else if (k == "--copy-k") a.copy_k = std::stoi(val());
    else if (k == "--skew") a.skew = std::stoll(val());
    else if (k == "--csv") a.csv = true;
    else if (k == "--ndjson") a.ndjson = true;
    else if (k == "--no-hipfft") a.hipfft = false;
    else if (k == "--inside-repeats") a.inside_repeats = std::stoi(val());
    else { std::fprintf(stderr, "unknown argument %s\n", k.c_str()); return 2; }
  }
  CHECK_HIP(hipSetDevice(0));
  try {
    for (int n : a.ns) {
      if (a.dtype == "fp32") run_one<float>(a, n);
      else if (a.dtype == "fp16") run_one<__half>(a, n);
      else if (a.dtype == "bf16") run_one<__hip_bfloat16>(a, n);
      else { std::fprintf(stderr, "unknown dtype %s\n", a.dtype.c_str()); return 2; }
    }
    thunderfft::ThunderFFTFinalize<float>();
    thunderfft::ThunderFFTFinalize<__half>();
    thunderfft::ThunderFFTFinalize<__hip_bfloat16>();
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 1;
  }
  stat::set_title("ThunderFFT (MI355X) benchmark, e2e = mean of " + std::to_string(a.runs) + " launches");
  if (a.csv) stat::print_csv();
  else if (a.ndjson) stat::print_ndjson();
  else stat::print_table();
  return 0;
}
