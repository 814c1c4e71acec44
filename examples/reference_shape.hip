// User kernels written against the reference's device-level signatures
// (thunderfft/include/thunderfft/thunderfft.cuh:55-93, 128-130): no transform object, registers of the storage
// type, the block's shared memory as staging array and workspace at once, blocks of (threads_per_warp,
// warp_per_block<N>) threads. The call sequences are those of the reference's bench and convolution samples
// (sample/bench/baseline/thunderfft_bench.h:57-96; sample/conv1d/conv1d_fullfft_l4096_thunder.cu:148-194;
// sample/conv1d/main.cu:38-129); the kernels themselves are new. __launch_bounds__ is the one addition a HIP
// build wants: without it the compiler budgets registers for 1024 threads per block and the 32-point-per-thread
// lengths spill a few values.
#include <thunderfft/thunderfft.cuh>
#include <thunderfft/thunderfft_layout_tuned.h>

namespace {

template <typename T, int N>
struct shape {
  static constexpr int BPB = thunderfft::batch_per_block<N>;
  static constexpr int WPB = thunderfft::warp_per_block<N>;
  static constexpr int ept = N * BPB / (thunderfft::threads_per_warp * WPB);
  static constexpr size_t smem_bytes = sizeof(vec2_t<T>) * (N + pad_h(N)) * BPB;
  static dim3 block() { return dim3(thunderfft::threads_per_warp, WPB); }
};

// (1) One transform per call chain, straight between global memory and registers.
template <typename T, int N, bool forward>
__global__ __launch_bounds__(thunderfft::block_threads<N>) void direct_kernel(const vec2_t<T>* __restrict__ d_input, vec2_t<T>* __restrict__ d_output) {
  constexpr int BPB = shape<T, N>::BPB, ept = shape<T, N>::ept;
  extern __shared__ __align__(16) unsigned char _smem[];
  vec2_t<T> reg[ept];
  thunderfft::ThunderFFT_gmem2reg<T, N, BPB>(reg, d_input + (size_t)blockIdx.x * BPB * N);
  thunderfft::ThunderFFT_kernel_reg<T, N, BPB, forward>(reg, nullptr, _smem);
  thunderfft::ThunderFFT_reg2gmem<T, N, BPB>(d_output + (size_t)blockIdx.x * BPB * N, reg);
}

// (2) The five staged calls of the reference's end-to-end bench kernel.
template <typename T, int N, bool forward>
__global__ __launch_bounds__(thunderfft::block_threads<N>) void staged_kernel(const vec2_t<T>* __restrict__ d_input, vec2_t<T>* __restrict__ d_output) {
  constexpr int BPB = shape<T, N>::BPB, ept = shape<T, N>::ept;
  extern __shared__ __align__(16) unsigned char _smem[];
  vec2_t<T>* s_in = reinterpret_cast<vec2_t<T>*>(_smem);
  vec2_t<T> reg[ept];
  using L_in = typename thunderfft::bench_layout<T, N, BPB>::L_in;
  using L_out = typename thunderfft::bench_layout<T, N, BPB>::L_out;

  vec2_t<T> W[36];
  if constexpr (std::is_same_v<T, __half>) thunderfft::unit_fp16::make_reg_b_precompute<N, forward>(W);

  thunderfft::ThunderFFT_gmem2smem<T, L_in>(s_in, d_input + (size_t)blockIdx.x * BPB * N);
  __syncthreads();
  thunderfft::ThunderFFT_smem2reg<T, L_in>(reg, s_in);
  __syncthreads();
  thunderfft::ThunderFFT_kernel_reg<T, N, BPB, forward>(reg, W, s_in);
  __syncthreads();
  thunderfft::ThunderFFT_reg2smem<T, L_out>(s_in, reg);
  __syncthreads();
  thunderfft::ThunderFFT_smem2gmem<T, L_out>(d_output + (size_t)blockIdx.x * BPB * N, s_in);
}

__device__ __forceinline__ float2 cprod(float2 a, float2 b) { return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ __half2 cprod(__half2 a, __half2 b) {
  return __float22half2_rn(cprod(__half22float2(a), __half22float2(b)));
}

// (3) Circular convolution through the frequency domain in one kernel: forward transform, product with the
// spectrum of filter (signal index % D) in register order, inverse transform. Signals [batch][N], spectra [D][N]
// in natural order; unnormalised, so the caller scales the filter by 1 / N.
template <typename T, int N>
__global__ __launch_bounds__(thunderfft::block_threads<N>) void filter_kernel(const vec2_t<T>* __restrict__ d_input, const vec2_t<T>* __restrict__ d_H,
                              vec2_t<T>* __restrict__ d_output, int D) {
  static_assert(thunderfft::is_chainable<N>, "forward output must be valid inverse input");
  constexpr int BPB = shape<T, N>::BPB, ept = shape<T, N>::ept;
  extern __shared__ __align__(16) unsigned char _smem[];
  vec2_t<T>* s_in = reinterpret_cast<vec2_t<T>*>(_smem);
  vec2_t<T> reg[ept];
  using L_in = typename thunderfft::bench_layout<T, N, BPB>::L_in;
  using L_out = typename thunderfft::bench_layout<T, N, BPB>::L_out;

  vec2_t<T> W_fwd[36], W_inv[36];
  if constexpr (std::is_same_v<T, __half>) {
    thunderfft::unit_fp16::make_reg_b_precompute<N, true>(W_fwd);
    thunderfft::unit_fp16::make_reg_b_precompute<N, false>(W_inv);
  }

  thunderfft::ThunderFFT_gmem2smem<T, L_in>(s_in, d_input + (size_t)blockIdx.x * BPB * N);
  __syncthreads();
  thunderfft::ThunderFFT_smem2reg<T, L_in>(reg, s_in);
  __syncthreads();
  thunderfft::ThunderFFT_kernel_reg<T, N, BPB, true>(reg, W_fwd, s_in);
  __syncthreads();

#pragma unroll
  for (int e = 0; e < ept; ++e) {
    const long long signal = (long long)blockIdx.x * BPB + thunderfft::transform_out<N>(e);
    const int d = (int)(signal % D);
    reg[e] = cprod(reg[e], d_H[(size_t)d * N + thunderfft::index_out<N>(e)]);
  }

  thunderfft::ThunderFFT_kernel_reg<T, N, BPB, false>(reg, W_inv, s_in);
  __syncthreads();
  thunderfft::ThunderFFT_reg2smem<T, L_out>(s_in, reg);
  __syncthreads();
  thunderfft::ThunderFFT_smem2gmem<T, L_out>(d_output + (size_t)blockIdx.x * BPB * N, s_in);
}

// (4) Overlap-save tiles of 64 points, fp16: the block gathers its tiles into a padded, digit-reversed layout by
// hand, as the reference's tiled convolution does, and hands only the registers to the library.
//   out[t * valid + j] = IDFT( DFT(in[t * valid ... + 64)) * hf )[j],  j < valid
__global__ __launch_bounds__(thunderfft::block_threads<64>) void tiles64_kernel(const __half2* __restrict__ input, const __half2* __restrict__ hf,
                               __half2* __restrict__ output, int input_len, int output_len, int num_tiles, int valid) {
  constexpr int N = 64;
  constexpr int BPB = shape<__half, N>::BPB, ept = shape<__half, N>::ept;
  constexpr int threads = thunderfft::threads_per_warp * shape<__half, N>::WPB;
  extern __shared__ __align__(16) unsigned char _smem[];
  __half2* s_in = reinterpret_cast<__half2*>(_smem);
  __half2 reg[ept];
  using L_in = thunderfft::layout_t<N, BPB, 1, N, 64, 4, true>;
  using L_out = thunderfft::layout_t<N, BPB, 1, N, 16, 1, false>;
  const int tidx = threadIdx.x + threadIdx.y * blockDim.x;
  const int tile0 = blockIdx.x * BPB;

  __half2 W_fwd[28], W_inv[28];
  thunderfft::unit_fp16::make_reg_b_precompute<N, true>(W_fwd);
  thunderfft::unit_fp16::make_reg_b_precompute<N, false>(W_inv);

  __half2 filter_reg[ept];
#pragma unroll
  for (int e = 0; e < ept; ++e) filter_reg[e] = hf[thunderfft::index_out<N>(e)];

  for (int q = tidx; q < N * BPB; q += threads) {
    const int i = q / N, j = q % N, tile = tile0 + i;
    int slot = i * N + j;
    slot += (slot / 64) * 4;
    const long long at = (long long)tile * valid + reverse_bit_groups<2, 6>(j);
    s_in[slot] = (tile < num_tiles && at < input_len) ? input[at] : __half2{};
  }
  __syncthreads();
  thunderfft::ThunderFFT_smem2reg<__half, L_in>(reg, s_in);
  __syncthreads();
  thunderfft::ThunderFFT_kernel_reg<__half, N, BPB, true>(reg, W_fwd, s_in);
  __syncthreads();
#pragma unroll
  for (int e = 0; e < ept; ++e) reg[e] = cprod(reg[e], filter_reg[e]);
  thunderfft::ThunderFFT_kernel_reg<__half, N, BPB, false>(reg, W_inv, s_in);
  __syncthreads();
  thunderfft::ThunderFFT_reg2smem<__half, L_out>(s_in, reg);
  __syncthreads();
  for (int q = tidx; q < valid * BPB; q += threads) {
    const int i = q / valid, j = q % valid, tile = tile0 + i;
    const long long at = (long long)tile * valid + j;
    int slot = i * N + j;
    slot += slot / 16;
    if (tile < num_tiles && at < output_len) output[at] = s_in[slot];
  }
}

template <typename T>
void prepare(int n, bool publish) {
  if (publish)
    thunderfft::ThunderFFTInitialize<T>(n);  // builds the table and publishes it to this unit's kernels
}

template <typename Kernel, typename... Args>
int launch(Kernel kernel, unsigned blocks, dim3 block, size_t smem, hipStream_t stream, Args... args) {
  hipError_t rc = hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
  if (rc != hipSuccess) return (int)rc;
  kernel<<<dim3(blocks), block, smem, stream>>>(args...);
  return (int)hipGetLastError();
}

template <typename T, int N>
int run_fft(int staged, int forward, const void* in, void* out, long long batch, hipStream_t s) {
  using S = shape<T, N>;
  if (batch % S::BPB != 0) return -3;
  const unsigned blocks = (unsigned)(batch / S::BPB);
  auto* x = static_cast<const vec2_t<T>*>(in);
  auto* y = static_cast<vec2_t<T>*>(out);
  if (staged)
    return forward ? launch(staged_kernel<T, N, true>, blocks, S::block(), S::smem_bytes, s, x, y)
                   : launch(staged_kernel<T, N, false>, blocks, S::block(), S::smem_bytes, s, x, y);
  return forward ? launch(direct_kernel<T, N, true>, blocks, S::block(), S::smem_bytes, s, x, y)
                 : launch(direct_kernel<T, N, false>, blocks, S::block(), S::smem_bytes, s, x, y);
}

template <typename T>
int run_fft_n(int n, int staged, int forward, const void* in, void* out, long long batch, hipStream_t s) {
  switch (n) {
    case 64: return run_fft<T, 64>(staged, forward, in, out, batch, s);
    case 128: return run_fft<T, 128>(staged, forward, in, out, batch, s);
    case 256: return run_fft<T, 256>(staged, forward, in, out, batch, s);
    case 512: return run_fft<T, 512>(staged, forward, in, out, batch, s);
    case 1024: return run_fft<T, 1024>(staged, forward, in, out, batch, s);
    case 2048: return run_fft<T, 2048>(staged, forward, in, out, batch, s);
    case 4096: return run_fft<T, 4096>(staged, forward, in, out, batch, s);
    case 8192: return run_fft<T, 8192>(staged, forward, in, out, batch, s);
    case 16384: return run_fft<T, 16384>(staged, forward, in, out, batch, s);
    default: return -1;
  }
}

template <typename T, int N>
int run_filter(const void* in, const void* H, void* out, long long batch, int D, hipStream_t s) {
  using S = shape<T, N>;
  if (batch % S::BPB != 0 || D <= 0) return -3;
  return launch(filter_kernel<T, N>, (unsigned)(batch / S::BPB), S::block(), S::smem_bytes, s,
                static_cast<const vec2_t<T>*>(in), static_cast<const vec2_t<T>*>(H), static_cast<vec2_t<T>*>(out), D);
}

}  // namespace

// dtype: TFFT_F32, TFFT_F16 or TFFT_BF16; n = 64 ... 16384; batch a multiple of batch_per_block<n>. publish = 0
// leaves ThunderFFTInitialize out (the kernels then evaluate their twiddle bases themselves). Returns 0, a
// hipError_t, -1 (unsupported), -2 (library error), -3 (bad batch).
extern "C" int refshape_fft(int dtype, int n, int staged, int forward, int publish, const void* in, void* out,
                            long long batch, void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  try {
    switch (dtype) {
      case TFFT_F32: prepare<float>(n, publish); return run_fft_n<float>(n, staged, forward, in, out, batch, s);
      case TFFT_F16: prepare<__half>(n, publish); return run_fft_n<__half>(n, staged, forward, in, out, batch, s);
      case TFFT_BF16:
        prepare<__hip_bfloat16>(n, publish);
        return run_fft_n<__hip_bfloat16>(n, staged, forward, in, out, batch, s);
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return -2;
  }
  return -1;
}

// Forgets the tables published to this unit (the library keeps them), so that the next launches run without.
extern "C" void refshape_unpublish(void) {
  for (int dtype = 0; dtype < 3; ++dtype) thunderfft::detail::publish_tables(dtype, 0);
}

// y[b] = IDFT(DFT(x[b]) * H[b % D]); n = 64, 128, 256, 1024 or 4096 (fp32, fp16).
extern "C" int refshape_filter(int dtype, int n, const void* in, const void* H, void* out, long long batch, int D,
                               void* stream) {
  hipStream_t s = static_cast<hipStream_t>(stream);
  try {
    if (dtype == TFFT_F32) {
      prepare<float>(n, true);
      switch (n) {
        case 64: return run_filter<float, 64>(in, H, out, batch, D, s);
        case 128: return run_filter<float, 128>(in, H, out, batch, D, s);
        case 256: return run_filter<float, 256>(in, H, out, batch, D, s);
        case 1024: return run_filter<float, 1024>(in, H, out, batch, D, s);
        case 4096: return run_filter<float, 4096>(in, H, out, batch, D, s);
      }
    } else if (dtype == TFFT_F16) {
      prepare<__half>(n, true);
      switch (n) {
        case 64: return run_filter<__half, 64>(in, H, out, batch, D, s);
        case 256: return run_filter<__half, 256>(in, H, out, batch, D, s);
        case 4096: return run_filter<__half, 4096>(in, H, out, batch, D, s);
      }
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return -2;
  }
  return -1;
}

// Overlap-save tiles of 64 points on fp16 data: same arguments and result as tfft_conv_tiles_exec(TFFT_F16, 64, ...).
extern "C" int refshape_tiles64(const void* in, const void* hf, void* out, long long input_len, long long output_len,
                                int valid, void* stream) {
  using S = shape<__half, 64>;
  if (valid <= 0 || valid > 64 || input_len <= 0 || output_len <= 0 || input_len > 0x7fffffff) return -3;
  try {
    prepare<__half>(64, true);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return -2;
  }
  const int num_tiles = (int)((output_len + valid - 1) / valid);
  return launch(tiles64_kernel, (unsigned)((num_tiles + S::BPB - 1) / S::BPB), S::block(), S::smem_bytes,
                static_cast<hipStream_t>(stream), static_cast<const __half2*>(in), static_cast<const __half2*>(hf),
                static_cast<__half2*>(out), (int)input_len, (int)output_len, num_tiles, valid);
}
