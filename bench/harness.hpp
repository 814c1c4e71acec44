// Host-side plumbing shared by the sample binaries (thff_conv1d_fullfft4096, thff_conv1d, thff_fft):
// positional arguments, device buffers, synthetic data, event timing and fp64 error sums. The binaries
// keep the reference samples' command lines and report lines (what its drivers parse); everything
// between argument parsing and printing is this build's own.
#pragma once
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <cmath>
#include <complex>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <initializer_list>
#include <string>
#include <utility>
#include <vector>

#include <thunderfft/thunderfft.cuh>

namespace harness {

using cplx = std::complex<double>;

[[noreturn]] inline void die(const std::string& msg) {
  std::fprintf(stderr, "%s\n", msg.c_str());
  std::exit(EXIT_FAILURE);
}

inline void hip_ok(hipError_t e, const char* what) {
  if (e != hipSuccess) die(std::string(what) + ": " + hipGetErrorString(e));
}

inline void tfft_ok(int rc, const char* what) {
  if (rc != TFFT_OK) die(std::string(what) + ": " + tfft_last_error());
}

// Positional integer arguments: slot i of `slots` takes argv[i + 1] when present, else keeps its default.
inline void positional(int argc, char** argv, std::initializer_list<int*> slots) {
  int i = 1;
  for (int* slot : slots) {
    if (i < argc) *slot = std::atoi(argv[i]);
    ++i;
  }
}

// Device allocation that frees itself; upload / download move whole std::vectors.
template <class T>
class DeviceArray {
 public:
  DeviceArray() = default;
  explicit DeviceArray(size_t count) { resize(count); }
  DeviceArray(const DeviceArray&) = delete;
  DeviceArray& operator=(const DeviceArray&) = delete;
  DeviceArray(DeviceArray&& o) noexcept : ptr_(o.ptr_), count_(o.count_) {
    o.ptr_ = nullptr;
    o.count_ = 0;
  }
  ~DeviceArray() {
    if (ptr_) (void)hipFree(ptr_);
  }
  void resize(size_t count) {
    if (ptr_) hip_ok(hipFree(ptr_), "hipFree");
    ptr_ = nullptr;
    count_ = count;
    if (count) hip_ok(hipMalloc(reinterpret_cast<void**>(&ptr_), sizeof(T) * count), "hipMalloc");
  }
  void upload(const std::vector<T>& host) {
    if (host.size() != count_) resize(host.size());
    if (count_) hip_ok(hipMemcpy(ptr_, host.data(), sizeof(T) * count_, hipMemcpyHostToDevice), "hipMemcpy H2D");
  }
  std::vector<T> download() const {
    std::vector<T> host(count_);
    if (count_) hip_ok(hipMemcpy(host.data(), ptr_, sizeof(T) * count_, hipMemcpyDeviceToHost), "hipMemcpy D2H");
    return host;
  }
  T* get() const { return ptr_; }
  size_t size() const { return count_; }

 private:
  T* ptr_ = nullptr;
  size_t count_ = 0;
};

// Counter-based uniform numbers in [0, 1): value k of stream `seed` depends on (k, seed) only, so any
// element can be regenerated on its own (the validation below re-reads inputs that way).
inline float uniform01(uint64_t k, uint64_t seed) {
  uint32_t x = (uint32_t)(((k + 1) * 0x9E3779B97F4A7C15ull + seed * 0xD1B54A32D192ED03ull) >> 32);
  x ^= x >> 16;
  x *= 0x7FEB352Du;
  x ^= x >> 15;
  x *= 0x846CA68Bu;
  x ^= x >> 16;
  return (float)(x >> 8) * (1.0f / (1 << 24));
}

inline cplx to_cplx(__half2 v) {
  const float2 f = __half22float2(v);
  return cplx(f.x, f.y);
}
inline cplx to_cplx(float2 v) { return cplx(v.x, v.y); }

// Mean duration in ms of `runs` back-to-back calls of fn on the null stream, after `warmup` untimed ones.
template <class F>
float mean_ms(F&& fn, int warmup, int runs) {
  hipEvent_t t0, t1;
  hip_ok(hipEventCreate(&t0), "hipEventCreate");
  hip_ok(hipEventCreate(&t1), "hipEventCreate");
  for (int i = 0; i < warmup; ++i) fn();
  hip_ok(hipDeviceSynchronize(), "hipDeviceSynchronize");
  hip_ok(hipEventRecord(t0), "hipEventRecord");
  for (int i = 0; i < runs; ++i) fn();
  hip_ok(hipEventRecord(t1), "hipEventRecord");
  hip_ok(hipEventSynchronize(t1), "hipEventSynchronize");
  float ms = 0.f;
  hip_ok(hipEventElapsedTime(&ms, t0, t1), "hipEventElapsedTime");
  (void)hipEventDestroy(t0);
  (void)hipEventDestroy(t1);
  return ms / (float)(runs > 0 ? runs : 1);
}

// Running error of a computed sequence against its expected values, all sums in fp64.
struct ErrorSums {
  double diff2 = 0, ref2 = 0, max_diff = 0, max_ref = 0;
  void add(cplx got, cplx want) {
    const double d = std::abs(got - want), r = std::abs(want);
    diff2 += d * d;
    ref2 += r * r;
    if (d > max_diff) max_diff = d;
    if (r > max_ref) max_ref = r;
  }
  double l2_rel() const { return ref2 > 0 ? std::sqrt(diff2 / ref2) : std::sqrt(diff2); }
  double linf_rel() const { return max_ref > 0 ? max_diff / max_ref : max_diff; }
};

// fp32 complex -> fp16 complex on the device (filter spectra that were transformed in fp32)
__global__ void narrow_to_half2(const float2* src, __half2* dst, size_t count) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < count) dst[i] = __floats2half2_rn(src[i].x, src[i].y);
}
inline void narrow(const float2* src, __half2* dst, size_t count, hipStream_t s = nullptr) {
  if (!count) return;
  narrow_to_half2<<<dim3((unsigned)((count + 255) / 256)), dim3(256), 0, s>>>(src, dst, count);
  hip_ok(hipGetLastError(), "narrow_to_half2");
}

}  // namespace harness
