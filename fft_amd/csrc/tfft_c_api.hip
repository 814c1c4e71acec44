// C ABI of libthunderfft_hip.so (declared in include/thunderfft_c.h): plan/twiddle cache and
// launch logic. Host-side counterpart of the reference's TwiddleCache<T> / ThunderFFTInitialize /
// ThunderFFTFinalize (thunderfft/include/thunderfft/detail/thunderfft_impl_twiddle.cuh:23-34,
// 261-311, 342-349) and of its launcher ThunderFFT_global
// (thunderfft/include/thunderfft/detail/thunderfft_impl_kernel.cuh:6-30).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstddef>
#include <cstdio>
#include <cstring>
#include <map>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <tuple>
#include <vector>

#include "../../include/thunderfft_c.h"
#include "conv_plans.hpp"
#include "fft_conv_mfma.hpp"
#include "plans.hpp"
#include "tile_plans.hpp"

// Scratch of tfft_exec_c2c_static: lives in the code object, so every GPU the library runs on holds its own 8 MiB
// copy from module load to process exit and no call ever allocates it. In host code the name is a handle for
// hipGetSymbolAddress only, never an address: each GPU's copy has its own device address, looked up once per
// device (static_scratch_for) and passed to the kernels like any other buffer.
constexpr long long kStaticScratchPoints = 1LL << 20;
static __device__ __align__(256) float4 g_static_scratch[kStaticScratchPoints / 2];

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

// Multi-pass plan of one large transform N = f[0]*f[1][*f[2]] (fp32 only): every factor is a
// tile-kernel length; sweeps 0..nf-2 are strided column sweeps with four-step twiddles of the
// enclosing length m[i], the last sweep reads contiguous rows and writes natural order.
struct BigPlan {
  int nf = 0;
  int f[3] = {0, 0, 0};
  float2* tw_f[3] = {nullptr, nullptr, nullptr};  // W_f^j tables of the factors
  float2* hi[2] = {nullptr, nullptr};             // four-step tables of sweeps 0 and 1
  float2* lo[2] = {nullptr, nullptr};
  int shift[2] = {0, 0};
  unsigned long long m[2] = {0, 0};
  float4* work = nullptr;  // library-owned scratch for `work_batch` transforms
  long long work_batch = 1;
  // The scratch is one buffer per (device, n): a call that uses it waits (on the device) for the previous such
  // call, whatever stream that ran on. `done` is recorded behind the last sweep of every call.
  hipEvent_t done = nullptr;
  bool used = false;
};

// Per-call launch options after validation (tfft_exec_options in the header).
struct ExecOpts {
  int plan = 0;
  int xcd_run_log2 = -1;
  long long grid_blocks_per_cu = 0;
  long long grid_blocks = 0;
  int extra_lds = 0;
  void* workspace = nullptr;
  unsigned long long workspace_bytes = 0;
};

// Readers (every call that looks a plan up and launches with its tables) hold the lock shared, across the launch;
// writers (initialize, finalize, users of the per-plan scratch state, the filter-layout tags) hold it exclusively.
// A finalize therefore never frees a table between a reader's look-up and its launch, and it drains the device
// before freeing what queued kernels still read.
std::shared_mutex g_mu;
using WriteLock = std::unique_lock<std::shared_mutex>;
using ReadLock = std::shared_lock<std::shared_mutex>;
std::map<std::tuple<int, int, int>, DevicePlan> g_plans;  // (device, dtype, n)
std::map<std::tuple<int, int>, BigPlan> g_big;             // (device, n), fp32
std::atomic<int> g_sync{0};                                // "sync": block after every launch like the reference

// Which register order each permuted filter buffer was last written in (tfft_conv_filter_layout), newest last.
struct ConvLayoutTag {
  int device, dtype;
  const char *begin, *end;  // the bytes the layout call wrote
  int layout_id;
};
std::vector<ConvLayoutTag> g_conv_tags;
constexpr size_t kMaxConvTags = 256;

// Drops every tag that overlaps [lo, hi) on `device`. g_mu must be held exclusively.
void forget_conv_layout(int device, const char* lo, const char* hi) {
  for (auto it = g_conv_tags.begin(); it != g_conv_tags.end();)
    it = (it->device == device && lo < it->end && it->begin < hi) ? g_conv_tags.erase(it) : it + 1;
}

std::map<int, float4*> g_static_scratch_addr;  // device -> address of that device's copy of g_static_scratch

constexpr int kMaxLdsN = 16384;       // longest transform one workgroup holds
constexpr int kMaxBigLog2 = 30;       // longest multi-pass transform

int ilog2(long long n) {
  int l = 0;
  while ((1LL << l) < n) ++l;
  return ((1LL << l) == n) ? l : -1;
}

// Factors of 2^lg into 2 or 3 tile lengths (2^6 .. 2^10), as even as possible, longest last
// (the last sweep reads contiguous rows of that length).
bool factor_big(int lg, int* nf, int f[3]) {
  if (lg <= 14 || lg > kMaxBigLog2) return false;
  const int parts = (lg <= 20) ? 2 : 3;
  int e[3] = {0, 0, 0};
  for (int i = 0; i < parts; ++i) e[i] = lg / parts + (i >= parts - lg % parts ? 1 : 0);
  for (int i = 0; i < parts; ++i)
    if (e[i] < 6 || e[i] > 10) return false;
  *nf = parts;
  for (int i = 0; i < 3; ++i) f[i] = (i < parts) ? (1 << e[i]) : 1;
  return true;
}

std::vector<float2> unit_roots(unsigned long long m, unsigned long long count, unsigned long long step) {
  std::vector<float2> h(count);
  for (unsigned long long j = 0; j < count; ++j) {
    const unsigned long long idx = (j * step) & (m - 1);
    const double a = -2.0 * M_PI * (double)idx / (double)m;
    h[j] = make_float2((float)std::cos(a), (float)std::sin(a));
  }
  return h;
}

hipError_t upload(const std::vector<float2>& h, float2** d) {
  hipError_t e = hipMalloc(reinterpret_cast<void**>(d), sizeof(float2) * h.size());
  if (e != hipSuccess) return e;
  return hipMemcpy(*d, h.data(), sizeof(float2) * h.size(), hipMemcpyHostToDevice);
}

void free_big(BigPlan& p) {
  for (int i = 0; i < 3; ++i) if (p.tw_f[i]) (void)hipFree(p.tw_f[i]);
  for (int i = 0; i < 2; ++i) {
    if (p.hi[i]) (void)hipFree(p.hi[i]);
    if (p.lo[i]) (void)hipFree(p.lo[i]);
  }
  if (p.work) (void)hipFree(p.work);
  if (p.done) (void)hipEventDestroy(p.done);
  p = BigPlan();
}

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

// Address of the current device's copy of the static scratch; resolved on first use per device, which
// tfft_initialize does for every multi-pass length. g_mu must be held.
int static_scratch_for(int dev, float4** out) {
  auto it = g_static_scratch_addr.find(dev);
  if (it == g_static_scratch_addr.end()) {
    void* addr = nullptr;
    TFFT_HIP(hipGetSymbolAddress(&addr, HIP_SYMBOL(g_static_scratch)));
    if (!addr) return fail(TFFT_ERR_HIP, "hipGetSymbolAddress(static scratch) returned a null address on device %d", dev);
    it = g_static_scratch_addr.emplace(dev, static_cast<float4*>(addr)).first;
  }
  *out = it->second;
  return TFFT_OK;
}

// Validates a caller's options. Kernel ids 1 and 2 of the lookup tables are the measurement builds with work
// compiled out; they are reachable through tfft_bench_access_pattern only, never as a plan.
int parse_options(const char* who, const tfft_exec_options* in, ExecOpts* out) {
  *out = ExecOpts();
  if (!in) return TFFT_OK;
  if (in->struct_size < (int)offsetof(tfft_exec_options, workspace) || in->struct_size > (int)sizeof(tfft_exec_options))
    return fail(TFFT_ERR_INVALID, "%s: options struct_size %d (expected %d; use tfft_exec_options_init)", who,
                in->struct_size, (int)sizeof(tfft_exec_options));
  if (in->plan < 0 || in->plan == 1 || in->plan == 2)
    return fail(TFFT_ERR_INVALID, "%s: plan %d is not a transform kernel (0 or >= 3)", who, in->plan);
  if (in->grid_blocks_per_cu < 0 || in->grid_blocks < 0 || in->extra_lds < 0 || in->extra_lds > 160 * 1024)
    return fail(TFFT_ERR_INVALID, "%s: negative grid size or extra_lds outside [0, 160 KiB]", who);
  out->plan = in->plan;
  out->xcd_run_log2 = in->xcd_run_log2 > 20 ? 20 : in->xcd_run_log2;
  out->grid_blocks_per_cu = in->grid_blocks_per_cu;
  out->grid_blocks = in->grid_blocks;
  out->extra_lds = in->extra_lds;
  if (in->struct_size >= (int)sizeof(tfft_exec_options)) {
    out->workspace = in->workspace;
    out->workspace_bytes = in->workspace_bytes;
  }
  return TFFT_OK;
}

int finish_launch(hipStream_t s) {
  TFFT_HIP(hipGetLastError());
  if (g_sync.load(std::memory_order_relaxed)) TFFT_HIP(hipStreamSynchronize(s));
  return TFFT_OK;
}

}  // namespace

// error reporting for the translation units that hold only kernels and their argument checks (dist_kernels.hip)
extern "C" int tfft_internal_fail(int code, const char* message) { return fail(code, "%s", message); }

extern "C" int tfft_fourstep_twiddle_launch(void* d_data, long long rows, long long cols, long long row0,
                                            long long col0, long long n, int forward, void* stream);

extern "C" {

int tfft_fourstep_twiddle(int dtype, void* d_data, long long rows, long long cols, long long row0, long long col0,
                          long long n, int forward, void* stream) {
  if (dtype != TFFT_F32) return fail(TFFT_ERR_UNSUPPORTED, "tfft_fourstep_twiddle: fp32 only");
  if (!d_data || rows < 0 || cols < 0 || (cols & 1) || n <= 0 || row0 < 0 || col0 < 0 ||
      (reinterpret_cast<uintptr_t>(d_data) & 15u))
    return fail(TFFT_ERR_INVALID, "tfft_fourstep_twiddle: bad arguments (cols must be even, buffer 16-byte aligned)");
  if (rows == 0 || cols == 0) return TFFT_OK;
  const int e = tfft_fourstep_twiddle_launch(d_data, rows, cols, row0, col0, n, forward, stream);
  if (e != 0) return fail(TFFT_ERR_HIP, "tfft_fourstep_twiddle: %s", hipGetErrorString((hipError_t)e));
  return TFFT_OK;
}

const char* tfft_last_error(void) { return g_err.c_str(); }

void tfft_exec_options_init(tfft_exec_options* opt) {
  if (!opt) return;
  memset(opt, 0, sizeof *opt);
  opt->struct_size = (int)sizeof *opt;
  opt->xcd_run_log2 = -1;
}

int tfft_is_supported(int dtype, int n, int forward) {
  tfft::KernelEntry e;
  if (lookup(dtype, n, forward != 0, 0, &e)) return 1;
  int nf, f[3];
  return (dtype == TFFT_F32 && n > kMaxLdsN && factor_big(ilog2(n), &nf, f)) ? 1 : 0;
}

static int initialize_big(int n) {
  int nf, f[3];
  if (!factor_big(ilog2(n), &nf, f)) return fail(TFFT_ERR_UNSUPPORTED, "tfft_initialize: no multi-pass plan for n=%d", n);
  int dev = 0;
  TFFT_HIP(hipGetDevice(&dev));
  WriteLock lock(g_mu);
  auto key = std::make_tuple(dev, n);
  if (g_big.count(key)) return TFFT_OK;
  BigPlan p;
  p.nf = nf;
  hipError_t e = hipSuccess;
  unsigned long long m = (unsigned long long)n;
  for (int i = 0; i < nf && e == hipSuccess; ++i) {
    p.f[i] = f[i];
    e = upload(unit_roots((unsigned long long)f[i], (unsigned long long)f[i], 1), &p.tw_f[i]);
    if (i < nf - 1 && e == hipSuccess) {
      // sweep i multiplies by W_m^(q*k), m = length of the transform this sweep starts
      p.m[i] = m;
      const int lg = ilog2((long long)m);
      p.shift[i] = (lg + 1) / 2;
      e = upload(unit_roots(m, 1ull << p.shift[i], 1), &p.lo[i]);
      if (e == hipSuccess) e = upload(unit_roots(m, m >> p.shift[i], 1ull << p.shift[i]), &p.hi[i]);
      m /= (unsigned long long)f[i];
    }
  }
  // scratch: at least one transform, and enough (2^26 points, 512 MiB) that batches of shorter
  // transforms go through in few launches
  p.work_batch = std::max<long long>(1, (1LL << 26) / n);
  if (e == hipSuccess) e = hipMalloc(reinterpret_cast<void**>(&p.work), sizeof(float2) * (size_t)n * (size_t)p.work_batch);
  if (e == hipSuccess) e = hipEventCreateWithFlags(&p.done, hipEventDisableTiming);
  if (e != hipSuccess) {
    free_big(p);
    return fail(TFFT_ERR_HIP, "tfft_initialize(n=%d): %s", n, hipGetErrorString(e));
  }
  // the no-allocation entry point finds this device's copy of its static scratch already resolved
  float4* unused = nullptr;
  if (int rc = static_scratch_for(dev, &unused)) {
    free_big(p);
    return rc;
  }
  g_big[key] = p;
  return TFFT_OK;
}

int tfft_initialize(int dtype, int n) {
  if (dtype == TFFT_F32 && n > kMaxLdsN) return initialize_big(n);
  tfft::KernelEntry fwd, inv;
  if (!lookup(dtype, n, true, 0, &fwd) || !lookup(dtype, n, false, 0, &inv))
    return fail(TFFT_ERR_UNSUPPORTED, "tfft_initialize: no kernel for %s n=%d", dtype_name(dtype), n);
  int dev = 0;
  TFFT_HIP(hipGetDevice(&dev));
  WriteLock lock(g_mu);
  auto key = std::make_tuple(dev, dtype, n);
  if (g_plans.count(key)) return TFFT_OK;

  DevicePlan p;
  // everything that can fail without owning memory comes first, so that no error path leaks the table
  hipDeviceProp_t prop;
  TFFT_HIP(hipGetDeviceProperties(&prop, dev));
  p.cu_count = prop.multiProcessorCount;
  TFFT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&p.blocks_per_cu[0], fwd.fn, fwd.threads, 0));
  TFFT_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&p.blocks_per_cu[1], inv.fn, inv.threads, 0));
  // 16-bit storage: the table is followed by the lane-ordered tables of the matrix-core kernels
  // (n = 8192, 16384 run 4096-point passes inside: the tables of n = 4096)
  const bool mfma_tables = dtype != TFFT_F32 && n >= 64;
  // the fused convolution's packed-fp16 pass-1 twiddles (n = 4096), which the fp16 kernels of n = 8192, 16384 read too
  const bool conv_tables = mfma_tables && n >= tfft::Mfma4096::N;
  // fp32, n = 8192: the table of the 4096-point transforms this length is built from follows W_n
  const int sub = (dtype == TFFT_F32 && n == 8192) ? 4096 : 0;
  const size_t entries = (size_t)n + sub + (mfma_tables ? tfft::Mfma4096::TABLES : 0) + (conv_tables ? tfft::ConvMfma4096::TABLE : 0);
  std::vector<float2> h(entries);
  for (int j = 0; j < n; ++j) {
    // exact reduction of the angle: j/n is a dyadic rational, evaluated in fp64 and rounded once
    const double a = -2.0 * M_PI * (double)j / (double)n;
    h[j] = make_float2((float)std::cos(a), (float)std::sin(a));
  }
  for (int j = 0; j < sub; ++j) h[n + j] = h[(size_t)j * (n / sub)];
  if (mfma_tables) tfft::mfma_lane_tables(h.data() + n, n < tfft::Mfma4096::N ? n : tfft::Mfma4096::N);
  if (conv_tables) {
    tfft::conv_mfma4096_tables(h.data() + n + tfft::Mfma4096::TABLES);
  }
  TFFT_HIP(hipMalloc(reinterpret_cast<void**>(&p.tw), sizeof(float2) * entries));
  hipError_t e = hipMemcpy(p.tw, h.data(), sizeof(float2) * entries, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    (void)hipFree(p.tw);
    return fail(TFFT_ERR_HIP, "hipMemcpy(twiddle): %s", hipGetErrorString(e));
  }
  g_plans[key] = p;
  return TFFT_OK;
}

int tfft_finalize(int dtype) {
  WriteLock lock(g_mu);
  int rc = TFFT_OK;
  int cur = 0;
  (void)hipGetDevice(&cur);
  // Kernels already queued may still read the tables: wait for every device that holds one before freeing. (No
  // reader is between its look-up and its launch: they hold the lock shared across both.)
  {
    std::vector<int> devices;
    for (auto& kv : g_plans)
      if (std::get<1>(kv.first) == dtype) devices.push_back(std::get<0>(kv.first));
    if (dtype == TFFT_F32)
      for (auto& kv : g_big) devices.push_back(std::get<0>(kv.first));
    std::sort(devices.begin(), devices.end());
    devices.erase(std::unique(devices.begin(), devices.end()), devices.end());
    for (int d : devices) {
      (void)hipSetDevice(d);
      if (hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError();  // e.g. a capture in progress: the caller's to end
    }
  }
  for (auto it = g_conv_tags.begin(); it != g_conv_tags.end();)
    it = it->dtype == dtype ? g_conv_tags.erase(it) : it + 1;
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
  if (dtype == TFFT_F32) {
    for (auto& kv : g_big) {
      (void)hipSetDevice(std::get<0>(kv.first));
      free_big(kv.second);
    }
    g_big.clear();
  }
  (void)hipSetDevice(cur);
  return rc;
}

int tfft_get_plan_info(int dtype, int n, tfft_plan_info* info) {
  if (!info) return fail(TFFT_ERR_INVALID, "tfft_get_plan_info: null info");
  tfft::KernelEntry e;
  if (dtype == TFFT_F32 && n > kMaxLdsN) {
    int nf, f[3];
    if (!factor_big(ilog2(n), &nf, f))
      return fail(TFFT_ERR_UNSUPPORTED, "tfft_get_plan_info: no multi-pass plan for n=%d", n);
    tfft::TileEntry t;
    tfft::lookup_tile(f[nf - 1], true, tfft::kTileOutCol, false, &t);
    info->n = n;
    info->threads_per_block = t.threads;
    info->transforms_per_block = t.tpb;  // short transforms per tile of the last sweep
    info->points_per_thread = t.n * t.tpb / t.threads;
    for (int i = 0; i < 3; ++i) info->radix[i] = f[i];  // sweep lengths
    info->lds_bytes = t.lds_bytes;
    info->blocks_per_cu = 0;
    return TFFT_OK;
  }
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
    ReadLock lock(g_mu);
    auto it = g_plans.find(std::make_tuple(dev, dtype, n));
    if (it != g_plans.end()) info->blocks_per_cu = it->second.blocks_per_cu[0];
  }
  return TFFT_OK;
}

int tfft_get_twiddle_table(int dtype, int n, const void** d_table) {
  if (!d_table) return fail(TFFT_ERR_INVALID, "tfft_get_twiddle_table: null result pointer");
  int dev = 0;
  TFFT_HIP(hipGetDevice(&dev));
  ReadLock lock(g_mu);
  auto it = g_plans.find(std::make_tuple(dev, dtype, n));
  if (it == g_plans.end())
    return fail(TFFT_ERR_NOT_INITIALIZED, "tfft_get_twiddle_table: call tfft_initialize(%s, %d) on device %d first",
                dtype_name(dtype), n, dev);
  *d_table = it->second.tw;
  return TFFT_OK;
}

int tfft_set_option(const char* key, long long value) {
  if (!key) return fail(TFFT_ERR_INVALID, "tfft_set_option: null key");
  if (!strcmp(key, "sync")) {
    g_sync.store(value != 0, std::memory_order_relaxed);
    return TFFT_OK;
  }
  return fail(TFFT_ERR_INVALID, "tfft_set_option: unknown key '%s' (kernel and launch choices are per call: tfft_exec_options)", key);
}

int tfft_workspace_bytes(int dtype, int n, long long batch, unsigned long long* min_bytes,
                         unsigned long long* best_bytes) {
  if (batch < 0) return fail(TFFT_ERR_INVALID, "tfft_workspace_bytes: negative batch");
  unsigned long long lo = 0, best = 0;
  if (dtype == TFFT_F32 && n > kMaxLdsN) {
    int nf, f[3];
    if (!factor_big(ilog2(n), &nf, f))
      return fail(TFFT_ERR_UNSUPPORTED, "tfft_workspace_bytes: no multi-pass plan for n=%d", n);
    lo = sizeof(float2) * (unsigned long long)n;
    const long long group = std::min<long long>(std::max<long long>(batch, 1), std::max<long long>(1, (1LL << 26) / n));
    best = lo * (unsigned long long)group;
  } else {
    tfft::KernelEntry e;
    if (!lookup(dtype, n, true, 0, &e))
      return fail(TFFT_ERR_UNSUPPORTED, "tfft_workspace_bytes: no kernel for %s n=%d", dtype_name(dtype), n);
  }
  if (min_bytes) *min_bytes = lo;
  if (best_bytes) *best_bytes = best;
  return TFFT_OK;
}

// Plans of a multi-pass length: 0 = 64 KiB tiles; 3, 4, 5 = 128 KiB tiles for the 256- and 512-point sweeps, the
// 512-point sweeps only, the 256-point sweeps only (A/B partners).
static int wide_mask_of(int plan) { return plan == 3 ? 3 : plan == 4 ? 2 : plan == 5 ? 1 : 0; }

// The sweeps of `batch` large transforms through `scratch`, which holds `scratch_batch` transforms of n points.
static int enqueue_big(const BigPlan& p, int n, bool fwd, const void* d_in, void* d_out, long long batch,
                       float4* scratch, long long scratch_batch, int xcd_opt, int wide_mask, hipStream_t s) {
  constexpr int kSweep = tfft::kTileInCol | tfft::kTileOutCol | tfft::kTileTwiddle;
  for (long long b = 0; b < batch; b += scratch_batch) {
    const long long group = std::min<long long>(scratch_batch, batch - b);  // transforms per launch
    const float4* src = static_cast<const float4*>(d_in) + b * (long long)(n / 2);
    float4* dst = static_cast<float4*>(d_out) + b * (long long)(n / 2);
    long long rest = n;  // points of the transform the current sweep starts: f[i] * ... * f[nf-1]
    long long done = 1;  // product of the lengths already transformed
    for (int i = 0; i < p.nf; ++i) {
      const bool last = (i == p.nf - 1);
      tfft::TileEntry t;
      const bool wide = (p.f[i] == 256 && (wide_mask & 1)) || (p.f[i] == 512 && (wide_mask & 2));
      if (!tfft::lookup_tile(p.f[i], fwd, last ? tfft::kTileOutCol : kSweep, wide, &t))
        return fail(TFFT_ERR_UNSUPPORTED, "tfft_exec_c2c: no tile kernel of length %d", p.f[i]);
      tfft::TileArgs a{};
      a.xcd_log2 = xcd_opt >= 0 ? xcd_opt : 4;  // measured: +1 ... 1.5 points on the sweeps
      a.in = (i == 0) ? src : scratch;
      a.out = last ? dst : scratch;
      a.tw = p.tw_f[i];
      a.in_bi = a.out_bi = n;
      const long long cols = rest / p.f[i];  // columns of this sweep inside one enclosing transform
      if (!last) {
        // column sweep: `done` enclosing transforms of `rest` points, each with `cols` columns
        a.tw_hi = p.hi[i];
        a.tw_lo = p.lo[i];
        a.tw_shift = p.shift[i];
        a.tw_mmask = p.m[i] - 1;
        a.na = (int)done;
        a.nb = (int)(cols / t.tpb);
        a.in_a = a.out_a = rest;
        a.in_b = a.out_b = t.tpb;
        a.in_stride = a.out_stride = cols;
        a.q_a = 0;
        a.q_b = t.tpb;
      } else {
        // rows of f[last] contiguous points; row index = (k_0, ..., k_{nf-2}) in sweep order.
        // Tile over adjacent k_0 (row stride n / f[0]); natural position of output k_last of that
        // row is k_0 + f[0]*k_1 + ... + (n / f[last]) * k_last.
        const long long rows_rest = done / p.f[0];  // rows per k_0 value: f[1] * ... * f[nf-2]
        a.na = (int)rows_rest;
        a.nb = p.f[0] / t.tpb;
        a.in_a = p.f[i];
        a.in_b = (long long)t.tpb * (n / p.f[0]);
        a.in_stride = n / p.f[0];
        a.out_a = p.f[0];
        a.out_b = t.tpb;
        a.out_stride = done;
      }
      a.ntiles = group * a.na * a.nb;
      t.launch(a, s);
      TFFT_HIP(hipGetLastError());
      rest = cols;
      done *= p.f[i];
    }
  }
  return TFFT_OK;
}

// One large transform per batch entry: 2 or 3 sweeps of tile kernels through a scratch buffer.
static int exec_big(int n, int forward, const void* d_in, void* d_out, long long batch, hipStream_t s,
                    const ExecOpts& opt) {
  int dev = 0;
  TFFT_HIP(hipGetDevice(&dev));
  if (opt.workspace) {
    // caller-owned scratch: the tables are read-only, so nothing of the plan is shared with other calls
    if (reinterpret_cast<uintptr_t>(opt.workspace) & 15u)
      return fail(TFFT_ERR_INVALID, "tfft_exec_c2c_opt: workspace must be 16-byte aligned");
    const unsigned long long per = sizeof(float2) * (unsigned long long)n;
    if (opt.workspace_bytes < per)
      return fail(TFFT_ERR_INVALID, "tfft_exec_c2c_opt: workspace of %llu bytes, n=%d needs at least %llu",
                  opt.workspace_bytes, n, per);
    ReadLock lock(g_mu);
    auto it = g_big.find(std::make_tuple(dev, n));
    if (it == g_big.end())
      return fail(TFFT_ERR_NOT_INITIALIZED, "tfft_exec_c2c: call tfft_initialize(fp32, %d) on device %d first", n, dev);
    return enqueue_big(it->second, n, forward != 0, d_in, d_out, batch, static_cast<float4*>(opt.workspace),
                       (long long)(opt.workspace_bytes / per), opt.xcd_run_log2, wide_mask_of(opt.plan), s);
  }
  // library-owned scratch: enqueue under the lock, behind the previous user of the scratch
  WriteLock lock(g_mu);
  auto it = g_big.find(std::make_tuple(dev, n));
  if (it == g_big.end())
    return fail(TFFT_ERR_NOT_INITIALIZED, "tfft_exec_c2c: call tfft_initialize(fp32, %d) on device %d first", n, dev);
  BigPlan& p = it->second;
  // A stream under graph capture cannot wait on, or record, an event of the eager world: a captured call is ordered
  // inside its graph only, and keeping its replays apart from other users of the scratch is the caller's business
  // (tfft_exec_c2c_static and a caller workspace exist for exactly that).
  hipStreamCaptureStatus capture = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &capture) != hipSuccess) {
    (void)hipGetLastError();
    capture = hipStreamCaptureStatusNone;
  }
  const bool eager = capture == hipStreamCaptureStatusNone;
  // always behind the previous user, also on "the same" stream: a handle can be that of a destroyed and re-created
  // stream, and a wait on the stream the event was recorded on costs nothing
  if (eager && p.used) TFFT_HIP(hipStreamWaitEvent(s, p.done, 0));
  const int rc = enqueue_big(p, n, forward != 0, d_in, d_out, batch, p.work, p.work_batch, opt.xcd_run_log2, wide_mask_of(opt.plan), s);
  if (eager) {
    // recorded even after a failed sweep: whatever was enqueued still uses the scratch
    TFFT_HIP(hipEventRecord(p.done, s));
    p.used = true;
  }
  return rc;
}

// Shared body of tfft_exec_c2c(_opt) and tfft_bench_access_pattern: `kernel_id` indexes the lookup tables
// (0 default, 1 and 2 the measurement builds, >= 3 the alternative plans).
static int exec_small(const char* who, int dtype, int n, int forward, int kernel_id, const void* d_in, void* d_out,
                      long long batch, hipStream_t s, const ExecOpts& opt) {
  int dev = 0;
  TFFT_HIP(hipGetDevice(&dev));
  ReadLock lock(g_mu);  // held until the kernel that reads the plan's table is enqueued
  auto it = g_plans.find(std::make_tuple(dev, dtype, n));
  if (it == g_plans.end()) {
    tfft::KernelEntry probe;
    if (!lookup(dtype, n, forward != 0, 0, &probe))
      return fail(TFFT_ERR_UNSUPPORTED, "%s: no kernel for %s n=%d", who, dtype_name(dtype), n);
    return fail(TFFT_ERR_NOT_INITIALIZED, "%s: call tfft_initialize(%s, %d) on device %d first", who,
                dtype_name(dtype), n, dev);
  }
  const DevicePlan& plan = it->second;
  tfft::KernelEntry e;
  if (!lookup(dtype, n, forward != 0, kernel_id, &e))
    return fail(TFFT_ERR_UNSUPPORTED, "%s: no kernel for %s n=%d forward=%d plan=%d", who, dtype_name(dtype), n, forward,
                kernel_id);

  // Small batches of N = 512 / 1024 (fp32): the default plans hold 32 points per thread in 64-thread workgroups,
  // which leaves most of the chip idle when there are at most two workgroups per CU; the 16-points-per-thread
  // plans start twice as many waves and finish 10-30 % sooner there (N = 1024 x 1024: 5.0 vs 6.0 us).
  if (kernel_id == 0 && dtype == TFFT_F32 && (n == 512 || n == 1024) &&
      (batch + e.tpb - 1) / e.tpb <= 2LL * plan.cu_count) {
    tfft::KernelEntry small;
    if (lookup(dtype, n, forward != 0, 3 /* the alternative plan */, &small)) e = small;
  }
  const long long nchunks = (batch + e.tpb - 1) / e.tpb;
  if (e.one_workgroup_per_transform) {
    // these kernels take their transform from blockIdx alone: no resident form, and the grid is the batch
    if (opt.grid_blocks_per_cu > 0 || opt.grid_blocks > 0)
      return fail(TFFT_ERR_UNSUPPORTED, "%s: the kernel of %s n=%d runs one workgroup per transform and has no resident "
                  "grid (grid_blocks, grid_blocks_per_cu)", who, dtype_name(dtype), n);
    if (batch > 0x7fffffffLL)
      return fail(TFFT_ERR_INVALID, "%s: batch %lld exceeds one launch (2^31 - 1 transforms) for %s n=%d", who, batch,
                  dtype_name(dtype), n);
  }
  // Default: one workgroup per chunk. The hardware dispatcher then hands out chunks in address
  // order as workgroups retire, which streams HBM measurably better than a resident grid walking
  // the batch with a stride; grid_blocks_per_cu selects the resident (persistent) form instead.
  long long grid = nchunks;
  if (e.resident_per_cu > 0) grid = (long long)e.resident_per_cu * plan.cu_count;
  if (opt.grid_blocks_per_cu > 0) grid = opt.grid_blocks_per_cu * plan.cu_count;
  if (opt.grid_blocks > 0) grid = opt.grid_blocks;
  if (grid > nchunks) grid = nchunks;
  if (grid > 0x7fffffffLL) grid = 0x7fffffffLL;
  tfft::LaunchShape shape;
  shape.grid = (unsigned)grid;
  shape.extra_lds = (unsigned)opt.extra_lds;
  shape.xcd_log2 = opt.xcd_run_log2 >= 0 ? opt.xcd_run_log2
                                         : (d_in == d_out && e.xcd_log2_in_place >= 0 ? e.xcd_log2_in_place : e.xcd_log2);
  e.launch(d_in, d_out, plan.tw, batch, nchunks, shape, s);
  return finish_launch(s);
}

static int check_io(const char* who, const void* d_in, void* d_out, long long batch) {
  if (batch < 0) return fail(TFFT_ERR_INVALID, "%s: negative batch %lld", who, batch);
  if (batch == 0) return TFFT_OK;
  if (!d_in || !d_out) return fail(TFFT_ERR_INVALID, "%s: null device pointer", who);
  if ((reinterpret_cast<uintptr_t>(d_in) | reinterpret_cast<uintptr_t>(d_out)) & 15u)
    return fail(TFFT_ERR_INVALID, "%s: buffers must be 16-byte aligned", who);
  return TFFT_OK;
}

int tfft_exec_c2c_opt(int dtype, int n, int forward, const void* d_in, void* d_out, long long batch, void* stream,
                      const tfft_exec_options* options) {
  ExecOpts opt;
  if (int rc = parse_options("tfft_exec_c2c_opt", options, &opt)) return rc;
  if (int rc = check_io("tfft_exec_c2c", d_in, d_out, batch)) return rc;
  if (batch == 0) return TFFT_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  if (dtype == TFFT_F32 && n > kMaxLdsN) {
    if (opt.plan != 0 && !(opt.plan >= 3 && opt.plan <= 5))
      return fail(TFFT_ERR_UNSUPPORTED, "tfft_exec_c2c_opt: multi-pass lengths have plans 0, 3, 4 and 5");
    if (int rc = exec_big(n, forward, d_in, d_out, batch, s, opt)) return rc;
    return finish_launch(s);
  }
  return exec_small("tfft_exec_c2c", dtype, n, forward, opt.plan, d_in, d_out, batch, s, opt);
}

int tfft_exec_c2c(int dtype, int n, int forward, const void* d_in, void* d_out, long long batch,
                  void* stream) {
  return tfft_exec_c2c_opt(dtype, n, forward, d_in, d_out, batch, stream, nullptr);
}

int tfft_exec_c2c_static(int dtype, int n, int forward, const void* d_in, void* d_out, long long batch,
                         void* stream) {
  if (int rc = check_io("tfft_exec_c2c_static", d_in, d_out, batch)) return rc;
  if (batch == 0) return TFFT_OK;
  hipStream_t s = static_cast<hipStream_t>(stream);
  // one-kernel lengths use no scratch at all: the ordinary path already allocates nothing
  if (!(dtype == TFFT_F32 && n > kMaxLdsN))
    return exec_small("tfft_exec_c2c_static", dtype, n, forward, 0, d_in, d_out, batch, s, ExecOpts());
  if ((long long)n > kStaticScratchPoints)
    return fail(TFFT_ERR_UNSUPPORTED,
                "tfft_exec_c2c_static: n=%d exceeds the static scratch of %lld points (2^15 ... 2^20 supported); use "
                "tfft_exec_c2c or a caller workspace", n, kStaticScratchPoints);
  int dev = 0;
  TFFT_HIP(hipGetDevice(&dev));
  float4* scratch = nullptr;
  {
    // tfft_initialize has filled the address table for this device; only a call on a device that never saw an
    // initialize of its own takes the exclusive lock
    ReadLock lock(g_mu);
    auto known = g_static_scratch_addr.find(dev);
    if (known != g_static_scratch_addr.end()) scratch = known->second;
  }
  if (!scratch) {
    WriteLock lock(g_mu);
    if (int rc = static_scratch_for(dev, &scratch)) return rc;
  }
  ReadLock lock(g_mu);
  auto it = g_big.find(std::make_tuple(dev, n));
  if (it == g_big.end())
    return fail(TFFT_ERR_NOT_INITIALIZED, "tfft_exec_c2c_static: call tfft_initialize(fp32, %d) on device %d first", n, dev);
  // Every sweep writes the part of the scratch it later reads, so nothing needs clearing between calls.
  if (int rc = enqueue_big(it->second, n, forward != 0, d_in, d_out, batch, scratch, kStaticScratchPoints / n, -1, 0, s)) return rc;
  return finish_launch(s);
}

static int exec_half_traffic(int dtype, int n, int pattern, const void* d_in, void* d_out, long long batch, hipStream_t s);

int tfft_bench_access_pattern(int dtype, int n, int forward, int pattern, const void* d_in, void* d_out,
                              long long batch, void* stream, const tfft_exec_options* options) {
  ExecOpts opt;
  if (int rc = parse_options("tfft_bench_access_pattern", options, &opt)) return rc;
  if (pattern < 1 || pattern > 4)
    return fail(TFFT_ERR_INVALID,
                "tfft_bench_access_pattern: pattern %d (1 = loads and stores only, 2 = no LDS exchange, 3 = no stores, "
                "4 = no loads)", pattern);
  if (opt.plan != 0) return fail(TFFT_ERR_INVALID, "tfft_bench_access_pattern: measures the default kernel only");
  if (int rc = check_io("tfft_bench_access_pattern", d_in, d_out, batch)) return rc;
  if (batch == 0) return TFFT_OK;
  if (pattern >= 3) {
    if (dtype != TFFT_F16 && dtype != TFFT_BF16)
      return fail(TFFT_ERR_UNSUPPORTED, "tfft_bench_access_pattern: patterns 3 and 4 are builds of the 16-bit matrix-core kernels");
    return exec_half_traffic(dtype, n, pattern, d_in, d_out, batch, static_cast<hipStream_t>(stream));
  }
  return exec_small("tfft_bench_access_pattern", dtype, n, forward, pattern, d_in, d_out, batch,
                    static_cast<hipStream_t>(stream), opt);
}

int tfft_bench_onchip_repeats(int dtype, int n, int repeats, const void* d_in, void* d_out, long long batch,
                              void* stream) {
  if (int rc = check_io("tfft_bench_onchip_repeats", d_in, d_out, batch)) return rc;
  if (repeats < 1 || repeats > (1 << 20)) return fail(TFFT_ERR_INVALID, "tfft_bench_onchip_repeats: repeats in [1, 2^20]");
  if (batch == 0) return TFFT_OK;
  if (dtype != TFFT_F16 && dtype != TFFT_BF16)
    return fail(TFFT_ERR_UNSUPPORTED, "tfft_bench_onchip_repeats: the 16-bit matrix-core kernels only");
  int dev = 0;
  TFFT_HIP(hipGetDevice(&dev));
  ReadLock lock(g_mu);
  auto it = g_plans.find(std::make_tuple(dev, dtype, n));
  if (it == g_plans.end())
    return fail(TFFT_ERR_NOT_INITIALIZED, "tfft_bench_onchip_repeats: call tfft_initialize(%s, %d) on device %d first",
                dtype_name(dtype), n, dev);
  const DevicePlan& plan = it->second;
  hipStream_t s = static_cast<hipStream_t>(stream);
  const bool ok = dtype == TFFT_F16 ? tfft::launch_repeat_f16(n, d_in, d_out, plan.tw, batch, (unsigned)repeats, 3u, s)
                                    : tfft::launch_repeat_bf16(n, d_in, d_out, plan.tw, batch, (unsigned)repeats, 3u, s);
  if (!ok) return fail(TFFT_ERR_UNSUPPORTED, "tfft_bench_onchip_repeats: n = 256 ... 4096");
  return finish_launch(s);
}

// patterns 3 and 4 of tfft_bench_access_pattern: the matrix-core kernel of a 16-bit length without its stores / loads
static int exec_half_traffic(int dtype, int n, int pattern, const void* d_in, void* d_out, long long batch, hipStream_t s) {
  int dev = 0;
  TFFT_HIP(hipGetDevice(&dev));
  ReadLock lock(g_mu);
  auto it = g_plans.find(std::make_tuple(dev, dtype, n));
  if (it == g_plans.end())
    return fail(TFFT_ERR_NOT_INITIALIZED, "tfft_bench_access_pattern: call tfft_initialize(%s, %d) on device %d first",
                dtype_name(dtype), n, dev);
  const DevicePlan& plan = it->second;
  const unsigned io_mask = pattern == 3 ? 1u : 2u;
  const bool ok = dtype == TFFT_F16 ? tfft::launch_repeat_f16(n, d_in, d_out, plan.tw, batch, 1u, io_mask, s)
                                    : tfft::launch_repeat_bf16(n, d_in, d_out, plan.tw, batch, 1u, io_mask, s);
  if (!ok) return fail(TFFT_ERR_UNSUPPORTED, "tfft_bench_access_pattern: patterns 3 and 4 cover n = 256 ... 4096");
  return finish_launch(s);
}

static int conv_common(const char* who, int dtype, int n, int kind, int plan, tfft::ConvEntry* e) {
  if (kind < TFFT_CONV_COMPLEX || kind > TFFT_CONV_REAL_IN_OUT)
    return fail(TFFT_ERR_INVALID, "%s: unknown kind %d", who, kind);
  if (!tfft::lookup_conv(dtype, n, kind, plan, e))
    return fail(TFFT_ERR_UNSUPPORTED, "%s: no fused convolution kernel for %s n=%d kind=%d plan=%d", who,
                dtype_name(dtype), n, kind, plan);
  return TFFT_OK;
}

int tfft_conv_filter_layout_opt(int dtype, int n, const void* d_h_freq, void* d_h_perm, int D, void* stream,
                                const tfft_exec_options* options) {
  ExecOpts opt;
  if (int rc = parse_options("tfft_conv_filter_layout", options, &opt)) return rc;
  tfft::ConvEntry e;
  if (int rc = conv_common("tfft_conv_filter_layout", dtype, n, TFFT_CONV_COMPLEX, opt.plan, &e)) return rc;
  if (D < 0) return fail(TFFT_ERR_INVALID, "tfft_conv_filter_layout: negative D");
  if (D == 0) return TFFT_OK;
  if (!d_h_freq || !d_h_perm || d_h_freq == d_h_perm)
    return fail(TFFT_ERR_INVALID, "tfft_conv_filter_layout: null or aliased device pointers");
  int dev = 0;
  TFFT_HIP(hipGetDevice(&dev));
  e.layout(d_h_freq, d_h_perm, D, static_cast<hipStream_t>(stream));
  TFFT_HIP(hipGetLastError());
  {
    // remember the order this buffer now holds, for the check in tfft_conv_exec
    WriteLock lock(g_mu);
    const char* lo = static_cast<const char*>(d_h_perm);
    const char* hi = lo + (size_t)D * n * (dtype == TFFT_F32 ? 8 : 4);
    forget_conv_layout(dev, lo, hi);  // whatever these bytes held before is overwritten
    if (g_conv_tags.size() >= kMaxConvTags) g_conv_tags.erase(g_conv_tags.begin());
    g_conv_tags.push_back(ConvLayoutTag{dev, dtype, lo, hi, e.layout_id});
  }
  return TFFT_OK;
}

int tfft_conv_filter_spectra(int dtype, int n, const void* d_taps, void* d_h_perm, int D, void* stream) {
  tfft::ConvEntry e;
  if (int rc = conv_common("tfft_conv_filter_spectra", dtype, n, TFFT_CONV_COMPLEX, 0, &e)) return rc;
  if (!e.spectra)
    return fail(TFFT_ERR_UNSUPPORTED,
                "tfft_conv_filter_spectra: %s n=%d has no one-kernel form: call tfft_exec_c2c and tfft_conv_filter_layout",
                dtype_name(dtype), n);
  if (D < 0) return fail(TFFT_ERR_INVALID, "tfft_conv_filter_spectra: negative D");
  if (D == 0) return TFFT_OK;
  if (!d_taps || !d_h_perm || d_taps == d_h_perm)
    return fail(TFFT_ERR_INVALID, "tfft_conv_filter_spectra: null or aliased device pointers");
  if ((reinterpret_cast<uintptr_t>(d_taps) | reinterpret_cast<uintptr_t>(d_h_perm)) & 15u)
    return fail(TFFT_ERR_INVALID, "tfft_conv_filter_spectra: buffers must be 16-byte aligned");
  int dev = 0;
  TFFT_HIP(hipGetDevice(&dev));
  hipStream_t s = static_cast<hipStream_t>(stream);
  {
    ReadLock lock(g_mu);  // held from the look-up to the launch, like every transform call
    auto it = g_plans.find(std::make_tuple(dev, dtype, n));
    if (it == g_plans.end())
      return fail(TFFT_ERR_NOT_INITIALIZED, "tfft_conv_filter_spectra: call tfft_initialize(%s, %d) on device %d first",
                  dtype_name(dtype), n, dev);
    e.spectra(d_taps, d_h_perm, it->second.tw, D, s);
    TFFT_HIP(hipGetLastError());
  }
  {
    WriteLock lock(g_mu);
    const char* lo = static_cast<const char*>(d_h_perm);
    const char* hi = lo + (size_t)D * n * 4;
    forget_conv_layout(dev, lo, hi);
    if (g_conv_tags.size() >= kMaxConvTags) g_conv_tags.erase(g_conv_tags.begin());
    g_conv_tags.push_back(ConvLayoutTag{dev, dtype, lo, hi, e.layout_id});
  }
  return finish_launch(s);
}

int tfft_conv_forget_layout(const void* d_buffer, unsigned long long bytes) {
  if (!d_buffer) return fail(TFFT_ERR_INVALID, "tfft_conv_forget_layout: null pointer");
  int dev = 0;
  TFFT_HIP(hipGetDevice(&dev));
  WriteLock lock(g_mu);
  const char* lo = static_cast<const char*>(d_buffer);
  forget_conv_layout(dev, lo, lo + (bytes ? bytes : 1));
  return TFFT_OK;
}

int tfft_conv_filter_layout(int dtype, int n, const void* d_h_freq, void* d_h_perm, int D, void* stream) {
  return tfft_conv_filter_layout_opt(dtype, n, d_h_freq, d_h_perm, D, stream, nullptr);
}

int tfft_conv_exec_opt(int dtype, int n, int kind, const void* d_x, const void* d_h_perm, void* d_y, int B, int D,
                       void* stream, const tfft_exec_options* options) {
  ExecOpts opt;
  if (int rc = parse_options("tfft_conv_exec", options, &opt)) return rc;
  tfft::ConvEntry e;
  if (int rc = conv_common("tfft_conv_exec", dtype, n, kind, opt.plan, &e)) return rc;
  if (B < 0 || D < 0) return fail(TFFT_ERR_INVALID, "tfft_conv_exec: negative B or D");
  if (B == 0 || D == 0) return TFFT_OK;
  if ((long long)B * D > 0x7fffffffLL) return fail(TFFT_ERR_INVALID, "tfft_conv_exec: B * D exceeds 2^31 - 1");
  if (!d_x || !d_h_perm || !d_y) return fail(TFFT_ERR_INVALID, "tfft_conv_exec: null device pointer");
  if ((reinterpret_cast<uintptr_t>(d_x) | reinterpret_cast<uintptr_t>(d_y) | reinterpret_cast<uintptr_t>(d_h_perm)) & 15u)
    return fail(TFFT_ERR_INVALID, "tfft_conv_exec: buffers must be 16-byte aligned");
  int dev = 0;
  TFFT_HIP(hipGetDevice(&dev));
  ReadLock lock(g_mu);
  auto it = g_plans.find(std::make_tuple(dev, dtype, n));
  if (it == g_plans.end())
    return fail(TFFT_ERR_NOT_INITIALIZED, "tfft_conv_exec: call tfft_initialize(%s, %d) on device %d first",
                dtype_name(dtype), n, dev);
  const DevicePlan& plan = it->second;
  // a filter buffer this library permuted for the other kernel's register order would convolve silently wrong;
  // the D spectra this call reads are matched against the byte ranges the layout calls wrote
  const char* h_lo = static_cast<const char*>(d_h_perm);
  const char* h_hi = h_lo + (size_t)D * n * (dtype == TFFT_F32 ? 8 : 4);
  for (const ConvLayoutTag& t : g_conv_tags)
    if (t.device == dev && h_lo < t.end && t.begin < h_hi && t.layout_id != e.layout_id)
      return fail(TFFT_ERR_INVALID,
                  "tfft_conv_exec: d_h_perm was written by tfft_conv_filter_layout for the other kernel "
                  "(plan 0 and plan 3 use different filter orders); permute it again with this call's plan");
  e.launch(d_x, d_h_perm, d_y, plan.tw, B, D, (int)opt.grid_blocks_per_cu, static_cast<hipStream_t>(stream));
  return finish_launch(static_cast<hipStream_t>(stream));
}

int tfft_conv_exec(int dtype, int n, int kind, const void* d_x, const void* d_h_perm, void* d_y, int B, int D,
                   void* stream) {
  return tfft_conv_exec_opt(dtype, n, kind, d_x, d_h_perm, d_y, B, D, stream, nullptr);
}

int tfft_conv_tiles_exec(int dtype, int n, const void* d_in, const void* d_hf, void* d_out, long long input_len,
                         long long output_len, int valid, void* stream) {
  tfft::ConvTilesEntry e;
  if (!tfft::lookup_conv_tiles(dtype, n, &e))
    return fail(TFFT_ERR_UNSUPPORTED, "tfft_conv_tiles_exec: no tiled convolution kernel for %s n=%d", dtype_name(dtype), n);
  if (input_len < 0 || output_len < 0 || valid < 1 || valid > n)
    return fail(TFFT_ERR_INVALID, "tfft_conv_tiles_exec: need 1 <= valid <= n and non-negative lengths");
  if (output_len == 0) return TFFT_OK;
  if (!d_in || !d_hf || !d_out) return fail(TFFT_ERR_INVALID, "tfft_conv_tiles_exec: null device pointer");
  const long long ntiles = (output_len + valid - 1) / valid;
  if ((ntiles + e.tiles_per_block - 1) / e.tiles_per_block > 0x7fffffffLL)
    return fail(TFFT_ERR_INVALID, "tfft_conv_tiles_exec: too many tiles for one launch");
  int dev = 0;
  TFFT_HIP(hipGetDevice(&dev));
  ReadLock lock(g_mu);
  auto it = g_plans.find(std::make_tuple(dev, dtype, n));
  if (it == g_plans.end())
    return fail(TFFT_ERR_NOT_INITIALIZED, "tfft_conv_tiles_exec: call tfft_initialize(%s, %d) on device %d first",
                dtype_name(dtype), n, dev);
  const DevicePlan& plan = it->second;
  e.launch(d_in, d_hf, d_out, plan.tw, input_len, output_len, valid, static_cast<hipStream_t>(stream));
  return finish_launch(static_cast<hipStream_t>(stream));
}

}  // extern "C"
