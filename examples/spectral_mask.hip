// Example user kernel on the device-level API: band-stop filter in one pass over HBM.
//   out = IDFT( mask(DFT(in)) ) / N,  mask zeroes bins lo <= k < hi
// Forward transform, the mask and the inverse transform all happen on registers
// (thunderfft_device.hpp); the reference's conv samples are built the same way from its device
// pipeline (thunderfft/sample/conv1d/conv1d_fullfft_l4096_thunder.cu:148-194).
#include <thunderfft/thunderfft_device.hpp>

template <typename T, int N>
__global__ __launch_bounds__(thunderfft::block_threads<N>) void spectral_mask_kernel(const vec2_t<T>* __restrict__ in,
                                                                                      vec2_t<T>* __restrict__ out,
                                                                                      const float2* __restrict__ tw,
                                                                                      long long batch, int lo, int hi) {
  static_assert(thunderfft::is_chainable<N>, "pick a length whose transforms chain in registers");
  constexpr int E = thunderfft::elements_per_thread<N>, BPB = thunderfft::batch_per_block<N>;
  __shared__ __attribute__((aligned(16))) char workspace[thunderfft::workspace_bytes<N>];
  thunderfft::DeviceFFT<N> fft(tw, workspace);
  const long long first = (long long)blockIdx.x * BPB;
  const int valid = (int)((batch - first < BPB) ? batch - first : BPB);
  float2 reg[E];
  thunderfft::ThunderFFT_gmem2reg<T, N>(fft, reg, in + first * N, valid);
  thunderfft::ThunderFFT_kernel_reg<N, true>(fft, reg, false);
#pragma unroll
  for (int e = 0; e < E; ++e) {
    const int k = fft.index_out(e);
    const float keep = (k >= lo && k < hi) ? 0.f : 1.0f / N;
    reg[e] = make_float2(reg[e].x * keep, reg[e].y * keep);
  }
  thunderfft::ThunderFFT_kernel_reg<N, false>(fft, reg);
  thunderfft::ThunderFFT_reg2gmem<T, N>(fft, out + first * N, reg, valid);
}

// The same filter through the staged pipeline: the spectrum visits LDS in natural order, where the mask
// is applied by all threads in index order. Works for every length, chainable or not.
template <typename T, int N>
__global__ __launch_bounds__(thunderfft::block_threads<N>) void spectral_mask_staged_kernel(
    const vec2_t<T>* __restrict__ in, vec2_t<T>* __restrict__ out, const float2* __restrict__ tw, long long batch, int lo,
    int hi) {
  constexpr int E = thunderfft::elements_per_thread<N>, BPB = thunderfft::batch_per_block<N>;
  __shared__ __attribute__((aligned(16))) char workspace[thunderfft::workspace_bytes<N>];
  __shared__ __attribute__((aligned(16))) vec2_t<T> stage[N * BPB];
  thunderfft::DeviceFFT<N> fft(tw, workspace);
  const long long first = (long long)blockIdx.x * BPB;
  const int valid = (int)((batch - first < BPB) ? batch - first : BPB);
  float2 reg[E];
  thunderfft::ThunderFFT_gmem2smem<T, N>(stage, in + first * N, valid);
  __syncthreads();
  thunderfft::ThunderFFT_smem2reg<T, N>(fft, reg, stage);
  thunderfft::ThunderFFT_kernel_reg<N, true>(fft, reg, false);
  thunderfft::ThunderFFT_reg2smem<T, N>(fft, stage, reg);
  __syncthreads();
  for (int q = threadIdx.x; q < N * BPB; q += thunderfft::block_threads<N>) {
    const int k = q % N;
    const float keep = (k >= lo && k < hi) ? 0.f : 1.0f / N;
    const float2 v = thunderfft::detail::widen(stage[q]);
    stage[q] = thunderfft::detail::narrow<T>(make_float2(v.x * keep, v.y * keep));
  }
  __syncthreads();
  thunderfft::ThunderFFT_smem2reg<T, N>(fft, reg, stage);
  thunderfft::ThunderFFT_kernel_reg<N, false>(fft, reg);
  thunderfft::ThunderFFT_reg2smem<T, N>(fft, stage, reg);
  __syncthreads();
  thunderfft::ThunderFFT_smem2gmem<T, N>(out + first * N, stage, valid);
}

// Once more with a padded LDS layout (one unused element after every 64) through the layout-aware calls.
template <typename T, int N>
__global__ __launch_bounds__(thunderfft::block_threads<N>) void spectral_mask_layout_kernel(
    const vec2_t<T>* __restrict__ in, vec2_t<T>* __restrict__ out, const float2* __restrict__ tw, long long batch, int lo,
    int hi) {
  constexpr int E = thunderfft::elements_per_thread<N>, BPB = thunderfft::batch_per_block<N>;
  using L = thunderfft::layout_t<N, BPB, 1, N, 64, 1>;
  __shared__ __attribute__((aligned(16))) char workspace[thunderfft::workspace_bytes<N>];
  __shared__ __attribute__((aligned(16))) vec2_t<T> stage[L::slots];
  thunderfft::DeviceFFT<N> fft(tw, workspace);
  const long long first = (long long)blockIdx.x * BPB;
  const int valid = (int)((batch - first < BPB) ? batch - first : BPB);
  float2 reg[E];
  thunderfft::ThunderFFT_gmem2smem<T, L>(stage, in + first * N, valid);
  __syncthreads();
  thunderfft::ThunderFFT_smem2reg<T, L>(fft, reg, stage);
  thunderfft::ThunderFFT_kernel_reg<N, true>(fft, reg, false);
  thunderfft::ThunderFFT_reg2smem<T, L>(fft, stage, reg);
  __syncthreads();
  for (int q = threadIdx.x; q < N * BPB; q += thunderfft::block_threads<N>) {
    const int k = q % N, slot = L::slot(k, q / N);
    const float keep = (k >= lo && k < hi) ? 0.f : 1.0f / N;
    const float2 v = thunderfft::detail::widen(stage[slot]);
    stage[slot] = thunderfft::detail::narrow<T>(make_float2(v.x * keep, v.y * keep));
  }
  __syncthreads();
  thunderfft::ThunderFFT_smem2reg<T, L>(fft, reg, stage);
  thunderfft::ThunderFFT_kernel_reg<N, false>(fft, reg);
  thunderfft::ThunderFFT_reg2smem<T, L>(fft, stage, reg);
  __syncthreads();
  thunderfft::ThunderFFT_smem2gmem<T, L>(out + first * N, stage, valid);
}

template <typename T, int N>
static int launch_layout(const void* in, void* out, long long batch, int lo, int hi, hipStream_t stream) {
  const float2* tw = thunderfft::twiddle_table<T>(N);
  constexpr int BPB = thunderfft::batch_per_block<N>;
  const long long blocks = (batch + BPB - 1) / BPB;
  spectral_mask_layout_kernel<T, N><<<dim3((unsigned)blocks), dim3(thunderfft::block_threads<N>), 0, stream>>>(
      static_cast<const vec2_t<T>*>(in), static_cast<vec2_t<T>*>(out), tw, batch, lo, hi);
  return (int)hipGetLastError();
}

template <typename T, int N>
static int launch_staged(const void* in, void* out, long long batch, int lo, int hi, hipStream_t stream) {
  const float2* tw = thunderfft::twiddle_table<T>(N);
  constexpr int BPB = thunderfft::batch_per_block<N>;
  const long long blocks = (batch + BPB - 1) / BPB;
  spectral_mask_staged_kernel<T, N><<<dim3((unsigned)blocks), dim3(thunderfft::block_threads<N>), 0, stream>>>(
      static_cast<const vec2_t<T>*>(in), static_cast<vec2_t<T>*>(out), tw, batch, lo, hi);
  return (int)hipGetLastError();
}

template <typename T, int N>
static int launch(const void* in, void* out, long long batch, int lo, int hi, hipStream_t stream) {
  const float2* tw = thunderfft::twiddle_table<T>(N);
  constexpr int BPB = thunderfft::batch_per_block<N>;
  const long long blocks = (batch + BPB - 1) / BPB;
  spectral_mask_kernel<T, N><<<dim3((unsigned)blocks), dim3(thunderfft::block_threads<N>), 0, stream>>>(
      static_cast<const vec2_t<T>*>(in), static_cast<vec2_t<T>*>(out), tw, batch, lo, hi);
  return (int)hipGetLastError();
}

// dtype: TFFT_F32 or TFFT_F16; n: 64, 256, 1024 or 4096. Returns 0, a hipError_t, or -1 for an
// unsupported (dtype, n). tfft_initialize(dtype, n) must have been called.
extern "C" int spectral_mask(int dtype, int n, const void* in, void* out, long long batch, int lo, int hi, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  try {
    if (dtype == TFFT_F32) {
      switch (n) {
        case 64: return launch<float, 64>(in, out, batch, lo, hi, s);
        case 256: return launch<float, 256>(in, out, batch, lo, hi, s);
        case 1024: return launch<float, 1024>(in, out, batch, lo, hi, s);
        case 4096: return launch<float, 4096>(in, out, batch, lo, hi, s);
      }
    } else if (dtype == TFFT_F16) {
      switch (n) {
        case 256: return launch<__half, 256>(in, out, batch, lo, hi, s);
        case 4096: return launch<__half, 4096>(in, out, batch, lo, hi, s);
      }
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return -2;
  }
  return -1;
}

// The staged pipeline (gmem2smem -> smem2reg -> kernel_reg -> reg2smem -> smem2gmem). dtype: TFFT_F32 with
// n = 64, 512, 2048 or 4096, TFFT_F16 with n = 1024; TFFT_F32 with n = 128 or 1024 and TFFT_F16 with n = 256 go
// through a padded layout_t; same return codes.
extern "C" int spectral_mask_staged(int dtype, int n, const void* in, void* out, long long batch, int lo, int hi,
                                    void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  try {
    if (dtype == TFFT_F32) {
      switch (n) {
        case 64: return launch_staged<float, 64>(in, out, batch, lo, hi, s);
        case 128: return launch_layout<float, 128>(in, out, batch, lo, hi, s);
        case 1024: return launch_layout<float, 1024>(in, out, batch, lo, hi, s);
        case 512: return launch_staged<float, 512>(in, out, batch, lo, hi, s);
        case 2048: return launch_staged<float, 2048>(in, out, batch, lo, hi, s);
        case 4096: return launch_staged<float, 4096>(in, out, batch, lo, hi, s);
      }
    } else if (dtype == TFFT_F16) {
      switch (n) {
        case 256: return launch_layout<__half, 256>(in, out, batch, lo, hi, s);
        case 1024: return launch_staged<__half, 1024>(in, out, batch, lo, hi, s);
      }
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return -2;
  }
  return -1;
}
