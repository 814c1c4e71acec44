// C ABI of libthunderfft_hip.so (declared in include/thunderfft_c.h): plan/twiddle cache and
// launch logic. Host-side counterpart of the reference's TwiddleCache<T> / ThunderFFTInitialize /
// ThunderFFTFinalize (thunderfft/include/thunderfft/detail/thunderfft_impl_twiddle.cuh:23-34,
// 261-311, 342-349) and of its launcher ThunderFFT_global
// (thunderfft/include/thunderfft/detail/thunderfft_impl_kernel.cuh:6-30).
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/thunderfft_c.h"
#include "plans.hpp"

namespace {

thread_local std::string g_err;

int fail(int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  g_err = buf;
  return code;
}

#define TFFT_HIP(call)                                                                          \
  do {                                                                                          \
    hipError_t e_ = (call);                                                                     \
    if (e_ != hipSuccess) return fail(TFFT_ERR_HIP, "%s: %s", #call, hipGetErrorString(e_));    \
  } while (0)

struct DevicePlan {
  float2* tw = nullptr;  // W_n^j = (cos(2 pi j / n), -sin(2 pi j / n)), j < n
  int blocks_per_cu[2] = {0, 0};  // forward, inverse
  int cu_count = 0;
};

struct Options {
  long long grid_blocks_per_cu = 0;
  long long mode = 0;
  long long sync = 0;
  long long extra_lds = 0;
};

std::mutex g_mu;
std::map<std::tuple<int, int, int>, DevicePlan> g_plans;  // (device, dtype, n)
Options g_opt;

bool lookup(int dtype, int n, bool forward, int mode, tfft::KernelEntry* e) {
  switch (dtype) {
    case TFFT_F32:
      return tfft::lookup_f32_a(n, forward, mode, e) || tfft::lookup_f32_b(n, forward, mode, e) ||
             tfft::lookup_f32_c(n, forward, mode, e);
    case TFFT_F16:
      return tfft::lookup_f16(n, forward, mode, e);
    case TFFT_BF16:
      return tfft::lookup_bf16(n, forward, mode, e);
    default:
      return false;
  }
}

const char* dtype_name(int dtype) {
  switch (dtype) {
    case TFFT_F32: return "fp32";
    case TFFT_F16: return "fp16";
    case TFFT_BF16: return "bf16";
    default: return "unknown";
  }
}

}  // namespace

extern "C" {

const char* tfft_last_error(void) { return g_err.c_str(); }

int tfft_is_supported(int dtype, int n, int forward) {
  tfft::KernelEntry e;
  return lookup(dtype, n, forward != 0, 0, &e) ? 1 : 0;
}

int tfft_initialize(int dtype, int n) {
  tfft::KernelEntry fwd, inv;
  if (!lookup(dtype, n, true, 0, &fwd) || !lookup(dtype, n, false, 0, &inv))
    return fail(TFFT_ERR_UNSUPPORTED, "tfft_initialize: no kernel for %s n=%d", dtype_name(dtype), n);
  int dev = 0;
  TFFT_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lock(g_mu);
  auto key = std::make_tuple(dev, dtype, n);
  if (g_plans.count(key)) return TFFT_OK;

  DevicePlan p;
  std::vector<float2> h(n);
  for (int j = 0; j < n; ++j) {
    // exact reduction of the angle: j/n is a dyadic rational, evaluated in fp64 and rounded once
    const double a = -2.0 * M_PI * (double)j / (double)n;
    h[j] = make_float2((float)std::cos(a), (float)std::sin(a));
  }
  TFFT_HIP(hipMalloc(reinterpret_cast<void**>(&p.tw), sizeof(float2) * n));
  hipError_t e = hipMemcpy(p.tw, h.data(), sizeof(float2) * n, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(p.tw);
    return fail(TFFT_ERR_HIP, "hipMemcpy(twiddle): %s", hipGetErrorString(e));
  }
  hipDeviceProp_t prop;
  TFFT_HIP(hipGetDeviceProperties(&prop, dev));
  p.cu_count = prop.multiProcessorCount;
  TFFT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&p.blocks_per_cu[0], fwd.fn, fwd.threads, 0));
  TFFT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&p.blocks_per_cu[1], inv.fn, inv.threads, 0));
  g_plans[key] = p;
  return TFFT_OK;
}

int tfft_finalize(int dtype) {
  std::lock_guard<std::mutex> lock(g_mu);
  int rc = TFFT_OK;
  int cur = 0;
  (void)hipGetDevice(&cur);
  for (auto it = g_plans.begin(); it != g_plans.end();) {
    if (std::get<1>(it->first) == dtype) {
      (void)hipSetDevice(std::get<0>(it->first));
      hipError_t e = hipFree(it->second.tw);
      if (e != hipSuccess) rc = fail(TFFT_ERR_HIP, "hipFree(twiddle): %s", hipGetErrorString(e));
      it = g_plans.erase(it);
    } else {
      ++it;
    }
  }
  (void)hipSetDevice(cur);
  return rc;
}

int tfft_get_plan_info(int dtype, int n, tfft_plan_info* info) {
  if (!info) return fail(TFFT_ERR_INVALID, "tfft_get_plan_info: null info");
  tfft::KernelEntry e;
  if (!lookup(dtype, n, true, 0, &e))
    return fail(TFFT_ERR_UNSUPPORTED, "tfft_get_plan_info: no kernel for %s n=%d", dtype_name(dtype), n);
  info->n = e.n;
  info->threads_per_block = e.threads;
  info->transforms_per_block = e.tpb;
  info->points_per_thread = e.e;
  for (int i = 0; i < 3; ++i) info->radix[i] = e.radix[i];
  info->lds_bytes = e.lds_bytes;
  info->blocks_per_cu = 0;
  int dev = 0;
  if (hipGetDevice(&dev) == hipSuccess) {
    std::lock_guard<std::mutex> lock(g_mu);
    auto it = g_plans.find(std::make_tuple(dev, dtype, n));
    if (it != g_plans.end()) info->blocks_per_cu = it->second.blocks_per_cu[0];
  }
  return TFFT_OK;
}

int tfft_set_option(const char* key, long long value) {
  if (!key) return fail(TFFT_ERR_INVALID, "tfft_set_option: null key");
  std::lock_guard<std::mutex> lock(g_mu);
  if (!strcmp(key, "grid_blocks_per_cu")) g_opt.grid_blocks_per_cu = value;
  else if (!strcmp(key, "mode")) g_opt.mode = value;
  else if (!strcmp(key, "sync")) g_opt.sync = value;
  else if (!strcmp(key, "extra_lds")) g_opt.extra_lds = value;
  else return fail(TFFT_ERR_INVALID, "tfft_set_option: unknown key '%s'", key);
  return TFFT_OK;
}

int tfft_exec_c2c(int dtype, int n, int forward, const void* d_in, void* d_out, long long batch,
                  void* stream) {
  if (batch < 0) return fail(TFFT_ERR_INVALID, "tfft_exec_c2c: negative batch %lld", batch);
  if (batch == 0) return TFFT_OK;
  if (!d_in || !d_out) return fail(TFFT_ERR_INVALID, "tfft_exec_c2c: null device pointer");
  if ((reinterpret_cast<uintptr_t>(d_in) | reinterpret_cast<uintptr_t>(d_out)) & 15u)
    return fail(TFFT_ERR_INVALID, "tfft_exec_c2c: buffers must be 16-byte aligned");
  Options opt;
  DevicePlan plan;
  int dev = 0;
  TFFT_HIP(hipGetDevice(&dev));
  {
    std::lock_guard<std::mutex> lock(g_mu);
    opt = g_opt;
    auto it = g_plans.find(std::make_tuple(dev, dtype, n));
    if (it == g_plans.end()) {
      tfft::KernelEntry probe;
      if (!lookup(dtype, n, forward != 0, 0, &probe))
        return fail(TFFT_ERR_UNSUPPORTED, "tfft_exec_c2c: no kernel for %s n=%d", dtype_name(dtype), n);
      return fail(TFFT_ERR_NOT_INITIALIZED, "tfft_exec_c2c: call tfft_initialize(%s, %d) on device %d first",
                  dtype_name(dtype), n, dev);
    }
    plan = it->second;
  }
  tfft::KernelEntry e;
  if (!lookup(dtype, n, forward != 0, (int)opt.mode, &e))
    return fail(TFFT_ERR_UNSUPPORTED, "tfft_exec_c2c: no kernel for %s n=%d forward=%d mode=%lld",
                dtype_name(dtype), n, forward, opt.mode);

  const long long nchunks = (batch + e.tpb - 1) / e.tpb;
  // Default: one workgroup per chunk. The hardware dispatcher then hands out chunks in address
  // order as workgroups retire, which streams HBM measurably better than a resident grid walking
  // the batch with a stride; "grid_blocks_per_cu" selects the resident (persistent) form instead.
  long long grid = nchunks;
  if (opt.grid_blocks_per_cu > 0) grid = opt.grid_blocks_per_cu * plan.cu_count;
  if (grid > nchunks) grid = nchunks;
  if (grid > 0x7fffffffLL) grid = 0x7fffffffLL;
  hipStream_t s = static_cast<hipStream_t>(stream);
  e.launch(d_in, d_out, plan.tw, batch, nchunks, (unsigned)grid, (unsigned)opt.extra_lds, s);
  TFFT_HIP(hipGetLastError());
  if (opt.sync) TFFT_HIP(hipStreamSynchronize(s));
  return TFFT_OK;
}

}  // extern "C"
