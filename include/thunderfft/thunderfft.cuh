// ThunderFFT host API for MI355X (gfx950): the reference's entry points
// (thunderfft/include/thunderfft/thunderfft.cuh:39-44, 98-102) as thin templates over the
// C ABI of libthunderfft_hip.so (include/thunderfft_c.h). Include path and names are the
// reference's; the CUDA runtime types are replaced by their HIP counterparts.
//
//   thunderfft::ThunderFFTInitialize<float>(N);
//   thunderfft::ThunderFFT_global<float, N, /*forward=*/true>(d_in, d_out, batch, stream);
//   thunderfft::ThunderFFTFinalize<float>();
//
// T is float, __half or __hip_bfloat16 (storage type; arithmetic is fp32). Data is `batch`
// contiguous transforms of N interleaved complex points, natural order, unnormalised in both
// directions, in place allowed. Unlike the reference, inverse transforms exist for every N.
#pragma once
#include <hip/hip_bf16.h>
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

#include "../thunderfft_c.h"

// Interleaved complex element of storage type T (reference: detail/util_cuda.h:46-49).
template <typename T>
struct thunderfft_vec2 {
  using type = __half2;
};
template <>
struct thunderfft_vec2<float> {
  using type = float2;
};
template <>
struct thunderfft_vec2<__hip_bfloat16> {
  using type = __hip_bfloat162;
};
template <typename T>
using vec2_t = typename thunderfft_vec2<T>::type;

// Reference: CHECK_CUDA prints and exits (detail/util_cuda.h:10-18).
#define CHECK_HIP(call)                                                                   \
  do {                                                                                    \
    hipError_t err_ = (call);                                                             \
    if (err_ != hipSuccess) {                                                             \
      std::fprintf(stderr, "HIP error %s:%d: %s\n", __FILE__, __LINE__, hipGetErrorString(err_)); \
      std::exit(EXIT_FAILURE);                                                            \
    }                                                                                     \
  } while (0)
// The name the reference's samples check runtime calls with (detail/util_cuda.h:10-18): same print-and-exit.
#define CHECK_CUDA(call) CHECK_HIP(call)

namespace thunderfft {

template <typename T>
constexpr int dtype_code() {
  static_assert(std::is_same_v<T, float> || std::is_same_v<T, __half> || std::is_same_v<T, __hip_bfloat16>,
                "ThunderFFT: supported T are float, __half and __hip_bfloat16");
  return std::is_same_v<T, float> ? TFFT_F32 : (std::is_same_v<T, __half> ? TFFT_F16 : TFFT_BF16);
}

// Reference: tf_check_cuda throws std::runtime_error("[ThunderFFT] ...") (detail/util_cuda.h:20-24).
inline void check(int rc, const char* what) {
  if (rc != TFFT_OK) throw std::runtime_error(std::string("[ThunderFFT] ") + what + ": " + tfft_last_error());
}

namespace detail {
// The reference publishes its table addresses to device code through module-scope __constant__ pointer symbols
// (detail/thunderfft_impl_twiddle.cuh:42-50, 298-301). Device symbols are per translation unit here, so every
// unit that includes the device-level API (thunderfft_device.hpp) registers a function that writes the unit's own
// symbol, and Initialize / Finalize call them all: (dtype, n) after a table was built, (dtype, 0) after the
// tables of dtype were freed.
using table_publisher = void (*)(int dtype, int n);
inline std::vector<table_publisher>& table_publishers() {
  static std::vector<table_publisher> list;
  return list;
}
}  // namespace detail

// Persistent twiddle management (reference thunderfft.cuh:39-44).
template <typename T>
inline void ThunderFFTInitialize(int N) {
  check(tfft_initialize(dtype_code<T>(), N), "ThunderFFTInitialize");
  for (auto publish : detail::table_publishers()) publish(dtype_code<T>(), N);
}

template <typename T>
inline void ThunderFFTFinalize() {
  check(tfft_finalize(dtype_code<T>()), "ThunderFFTFinalize");
  for (auto publish : detail::table_publishers()) publish(dtype_code<T>(), 0);
}

// Launch wrapper (reference thunderfft.cuh:98-102, detail/thunderfft_impl_kernel.cuh:6-30).
// Like the reference it returns after the stream has drained; define THUNDERFFT_ASYNC before
// including this header to get a stream-ordered launch without the host synchronisation.
// The reference's fp32 path works without a prior Initialize (it passes W = nullptr,
// detail/kernel/global_kernel.cuh:43); the same call order is accepted here by initialising
// on first use.
template <typename T, int N, bool forward = true>
inline void ThunderFFT_global(vec2_t<T>* d_input, vec2_t<T>* d_output, int batch, hipStream_t stream = 0) {
  static_assert(N >= 64 && N <= 16384 && (N & (N - 1)) == 0, "ThunderFFT_global: N must be a power of two in 64..16384");
  int rc = tfft_exec_c2c(dtype_code<T>(), N, forward ? 1 : 0, d_input, d_output, batch, stream);
  if (rc == TFFT_ERR_NOT_INITIALIZED) {
    ThunderFFTInitialize<T>(N);
    rc = tfft_exec_c2c(dtype_code<T>(), N, forward ? 1 : 0, d_input, d_output, batch, stream);
  }
  check(rc, "ThunderFFT_global");
#ifndef THUNDERFFT_ASYNC
  CHECK_HIP(hipStreamSynchronize(stream));
#endif
}

// Launch geometry (reference: batch_per_block<N>, warp_per_block<N>, detail/thunderfft_impl.cuh:14-32).
// The values are properties of the compiled kernels, so they are queried, not constexpr.
template <typename T>
inline tfft_plan_info plan_info(int N) {
  tfft_plan_info info{};
  check(tfft_get_plan_info(dtype_code<T>(), N, &info), "plan_info");
  return info;
}

}  // namespace thunderfft

// Like the reference's header (thunderfft.cuh:135 pulls in its device code), one include gives a HIP translation
// unit the device-level calls as well. Define THUNDERFFT_HOST_API_ONLY to keep a unit to the three host calls.
#if defined(__HIPCC__) && !defined(THUNDERFFT_HOST_API_ONLY)
#include "thunderfft_device.hpp"
#endif
