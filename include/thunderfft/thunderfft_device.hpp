// ThunderFFT device-level API for MI355X (gfx950): the pieces of the batched kernel as device
// functions, so that a user kernel can put its own work between a forward and an inverse transform
// without a round trip through HBM.
//
// Counterpart of the reference's composable device pipeline (thunderfft/include/thunderfft/
// thunderfft.cuh:55-130):
//     ThunderFFT_gmem2smem -> ThunderFFT_smem2reg -> ThunderFFT_kernel_reg -> ThunderFFT_reg2smem ->
//     ThunderFFT_smem2gmem      (and the direct ThunderFFT_gmem2reg / ThunderFFT_reg2gmem, :55-65)
// The reference stages rows through shared memory into mma.m16n8k8 fragments. On wave64 hardware a
// thread's points go from global memory straight into VGPRs in the order its first butterfly pass
// wants them, so gmem2reg -> kernel_reg -> reg2gmem is the fast pipeline and LDS is only the
// transform's internal exchange buffer (`workspace`). The staged calls exist as well, for kernels that
// want the block's signals or spectra in natural order in LDS (work across bins or across transforms,
// lengths whose transforms do not chain in registers): `smem` is [batch_per_block<N>][N] elements of
// vec2_t<T>, natural order, smem_bytes<T, N> bytes, separate from `workspace`.
//
//   template <int N> __global__ __launch_bounds__(thunderfft::block_threads<N>)
//   void my_kernel(const float2* in, float2* out, const float2* tw, int batch) {
//     __shared__ __attribute__((aligned(16))) char workspace[thunderfft::workspace_bytes<N>];
//     thunderfft::DeviceFFT<N> fft(tw, workspace);
//     float2 reg[thunderfft::elements_per_thread<N>];
//     const long long first = (long long)blockIdx.x * thunderfft::batch_per_block<N>;
//     const int valid = min((long long)thunderfft::batch_per_block<N>, batch - first);
//     thunderfft::ThunderFFT_gmem2reg<float, N>(fft, reg, in + first * N, valid);
//     thunderfft::ThunderFFT_kernel_reg<N, true>(fft, reg, /*lead=*/false);
//     for (int e = 0; e < thunderfft::elements_per_thread<N>; ++e)   // reg[e] is bin fft.index_out(e)
//       reg[e] = my_filter(reg[e], fft.index_out(e));                //   of transform fft.transform_out(e)
//     thunderfft::ThunderFFT_kernel_reg<N, false>(fft, reg);
//     thunderfft::ThunderFFT_reg2gmem<float, N>(fft, out + first * N, reg, valid);
//   }
//
// The same seven calls also exist with exactly the reference's template and argument lists
// (thunderfft.cuh:55-93, 128-130), for kernels written against the reference:
//     ThunderFFT_gmem2smem<T, sL>(smem, gmem)      ThunderFFT_smem2gmem<T, sL>(gmem, smem)
//     ThunderFFT_smem2reg<T, sL>(reg, smem)        ThunderFFT_reg2smem<T, sL>(smem, reg)
//     ThunderFFT_gmem2reg<T, N, batch>(reg, gmem)  ThunderFFT_reg2gmem<T, N, batch>(gmem, reg)
//     ThunderFFT_kernel_reg<T, N, batch, forward>(reg, W, workspace)
// with reg = vec2_t<T>[N * batch / (threads_per_warp * warp_per_block<N>)], workspace = the block's
// shared memory of sizeof(vec2_t<T>) * (N + pad_h(N)) * batch bytes (it may be the staging array, as in the
// reference's samples) and W = the array unit_fp16::make_reg_b_precompute<N, forward>(W) filled. The block
// may be one- or two-dimensional (the reference launches (threads_per_warp, warp_per_block<N>)). See the
// section "the reference's signatures" below and examples/reference_shape.hip.
//
// `tw` is the table tfft_initialize built; tfft_get_twiddle_table returns its device address.
// Chaining a forward and an inverse transform in registers needs thunderfft::is_chainable<N>
// (the register order after a transform equals the order the next one expects): N = 64, 128, 256,
// 1024, 4096. Compile user kernels with hipcc --offload-arch=gfx950 -fno-slp-vectorize and
// -I<repo>/include; this header pulls the kernel sources from <repo>/fft_amd/csrc.
#pragma once
#include "thunderfft.cuh"

#include "../../fft_amd/csrc/fft_block.hpp"
#include "../../fft_amd/csrc/plans.hpp"

// Index x of N bits with its r-bit digits in reverse order (base-2^r digit reversal); the N % r bits above the
// last whole digit follow, lowest first (reference: reverse_bit_groups, detail/util_arith.h:17-30; global
// namespace like there).
template <int r, int N>
__host__ __device__ constexpr int reverse_bit_groups(int x) {
  constexpr int digits = N / r;
  int y = 0;
  for (int d = 0; d < digits; ++d) y |= ((x >> (r * d)) & ((1 << r) - 1)) << (r * (digits - 1 - d));
  for (int b = r * digits; b < N; ++b) y = (y << 1) | ((x >> b) & 1);
  return y;
}
// The inverse map for r = 2: reverse_bit_groups<2, BITS>(thunderfft_unreverse_base4<BITS>(y)) == y. For an even
// number of bits the reversal is its own inverse; for an odd number the spare top bit of x sits at the bottom of y.
template <int BITS>
__host__ __device__ constexpr int thunderfft_unreverse_base4(int y) {
  if constexpr (BITS % 2 == 0) return reverse_bit_groups<2, BITS>(y);
  return reverse_bit_groups<2, BITS - 1>(y >> 1) | ((y & 1) << (BITS - 1));
}

namespace thunderfft {

template <int N>
struct device_plan;
template <> struct device_plan<64> { using type = tfft::Plan64; };
template <> struct device_plan<128> { using type = tfft::Plan128; };
template <> struct device_plan<256> { using type = tfft::Plan256; };
template <> struct device_plan<512> { using type = tfft::Plan512; };
template <> struct device_plan<1024> { using type = tfft::Plan1024; };
template <> struct device_plan<2048> { using type = tfft::Plan2048; };
template <> struct device_plan<4096> { using type = tfft::Plan4096; };
template <> struct device_plan<8192> { using type = tfft::Plan8192; };
template <> struct device_plan<16384> { using type = tfft::Plan16384; };

template <typename T>
struct device_io;
template <> struct device_io<float> { using type = tfft::StoreF32; };
template <> struct device_io<__half> { using type = tfft::StoreF16; };
template <> struct device_io<__hip_bfloat16> { using type = tfft::StoreBF16; };

using tfft::cmul;  // complex product of two float2 (reference: cmul, detail/util_arith.h:39-50)

// One workgroup's transform state: per-thread index maps, twiddle bases, the LDS workspace.
template <int N>
using DeviceFFT = tfft::BlockFFT<typename device_plan<N>::type>;

// Launch geometry of a user kernel (reference: threads_per_warp * warp_per_block<N>,
// batch_per_block<N>, the `ept` of its samples and the shared-memory size,
// detail/thunderfft_impl.cuh:14-32, sample/conv1d/conv1d_fullfft_l4096_thunder.cu:25-27, 394).
inline constexpr int threads_per_warp = 64;  // wave64
template <int N> inline constexpr int block_threads = device_plan<N>::type::WG;
template <int N> inline constexpr int warp_per_block = device_plan<N>::type::WG / threads_per_warp;
template <int N> inline constexpr int batch_per_block = device_plan<N>::type::TPB;
template <int N> inline constexpr int elements_per_thread = device_plan<N>::type::E;
template <int N> inline constexpr int workspace_bytes = device_plan<N>::type::LDS_BYTES;
template <int N> inline constexpr bool is_chainable = DeviceFFT<N>::kChainable;

// Global memory -> register file (reference: ThunderFFT_gmem2reg, thunderfft.cuh:55-58). `gmem`
// addresses this workgroup's batch_per_block<N> transforms ([batch][N], 16-byte aligned); transforms
// numbered `valid` and above read as zero. reg[e] becomes point fft.index_in(e) of transform
// fft.transform_in(e), widened to fp32.
template <typename T, int N>
__device__ __forceinline__ void ThunderFFT_gmem2reg(const DeviceFFT<N>& fft, float2 (&reg)[elements_per_thread<N>],
                                                    const vec2_t<T>* __restrict__ gmem,
                                                    int valid = batch_per_block<N>) {
  using IO = typename device_io<T>::type;
  typename IO::raw raw[elements_per_thread<N> / 2];
  fft.template load<IO>(reinterpret_cast<const typename IO::raw*>(gmem), valid, raw);
  fft.template unpack<IO>(raw, reg);
}

// Register file -> global memory (reference: ThunderFFT_reg2gmem, thunderfft.cuh:61-65): reg[e] is
// stored as point fft.index_out(e) of transform fft.transform_out(e), rounded to T.
template <typename T, int N>
__device__ __forceinline__ void ThunderFFT_reg2gmem(const DeviceFFT<N>& fft, vec2_t<T>* __restrict__ gmem,
                                                    const float2 (&reg)[elements_per_thread<N>],
                                                    int valid = batch_per_block<N>) {
  using IO = typename device_io<T>::type;
  float2 tmp[elements_per_thread<N>];
#pragma unroll
  for (int e = 0; e < elements_per_thread<N>; ++e) tmp[e] = reg[e];
  fft.template store<IO>(reinterpret_cast<typename IO::raw*>(gmem), valid, tmp);
}

// Register-resident transform (reference: ThunderFFT_kernel_reg<T, N, batch, forward>(reg, W,
// workspace), thunderfft.cuh:127-130; W and workspace were given to the DeviceFFT constructor).
// Unnormalised in both directions. `lead`: the workspace may still be read by an earlier
// ThunderFFT_kernel_reg call or by user code, so begin with a barrier; pass false for the first use
// in a kernel.
template <int N, bool forward>
__device__ __forceinline__ void ThunderFFT_kernel_reg(const DeviceFFT<N>& fft, float2 (&reg)[elements_per_thread<N>],
                                                      bool lead = true) {
  fft.template transform<forward>(reg, lead);
}

// ---- staged pipeline -------------------------------------------------------------------------
template <typename T, int N> inline constexpr int smem_bytes = (int)sizeof(vec2_t<T>) * N * batch_per_block<N>;

namespace detail {
__device__ __forceinline__ float2 widen(float2 v) { return v; }
__device__ __forceinline__ float2 widen(__half2 v) { return __half22float2(v); }
__device__ __forceinline__ float2 widen(__hip_bfloat162 v) { return __bfloat1622float2(v); }
template <typename T> __device__ __forceinline__ vec2_t<T> narrow(float2 v);
template <> __device__ __forceinline__ float2 narrow<float>(float2 v) { return v; }
template <> __device__ __forceinline__ __half2 narrow<__half>(float2 v) { return __float22half2_rn(v); }
template <> __device__ __forceinline__ __hip_bfloat162 narrow<__hip_bfloat16>(float2 v) { return __float22bfloat162_rn(v); }
}  // namespace detail

// Global memory -> LDS, all threads of the block (reference: ThunderFFT_gmem2smem, thunderfft.cuh:68-71):
// the block's batch_per_block<N> transforms in natural order, 16 bytes per thread and step; transforms
// numbered `valid` and above are filled with zeros. Follow with __syncthreads().
template <typename T, int N>
__device__ __forceinline__ void ThunderFFT_gmem2smem(vec2_t<T>* smem, const vec2_t<T>* __restrict__ gmem,
                                                     int valid = batch_per_block<N>) {
  constexpr int PER = 16 / (int)sizeof(vec2_t<T>);                 // elements per 16-byte granule
  constexpr int GRANULES = N * batch_per_block<N> / PER;
  const tfft::u32x4_t* src = reinterpret_cast<const tfft::u32x4_t*>(gmem);
  tfft::u32x4_t* dst = reinterpret_cast<tfft::u32x4_t*>(smem);
  for (int q = threadIdx.x; q < GRANULES; q += block_threads<N>)
    dst[q] = (q * PER < valid * N) ? __builtin_nontemporal_load(src + q) : tfft::u32x4_t{0u, 0u, 0u, 0u};
}

// LDS -> global memory (reference: ThunderFFT_smem2gmem, thunderfft.cuh:74-77); transforms numbered
// `valid` and above are not written. Call after a __syncthreads() that follows the last LDS write.
template <typename T, int N>
__device__ __forceinline__ void ThunderFFT_smem2gmem(vec2_t<T>* __restrict__ gmem, const vec2_t<T>* smem,
                                                     int valid = batch_per_block<N>) {
  constexpr int PER = 16 / (int)sizeof(vec2_t<T>);
  constexpr int GRANULES = N * batch_per_block<N> / PER;
  const tfft::u32x4_t* src = reinterpret_cast<const tfft::u32x4_t*>(smem);
  tfft::u32x4_t* dst = reinterpret_cast<tfft::u32x4_t*>(gmem);
  for (int q = threadIdx.x; q < GRANULES; q += block_threads<N>)
    if (q * PER < valid * N) __builtin_nontemporal_store(src[q], dst + q);
}

// LDS -> register file in the order the transform wants (reference: ThunderFFT_smem2reg,
// thunderfft.cuh:80-83): reg[e] = point fft.index_in(e) of transform fft.transform_in(e).
template <typename T, int N>
__device__ __forceinline__ void ThunderFFT_smem2reg(const DeviceFFT<N>& fft, float2 (&reg)[elements_per_thread<N>],
                                                    const vec2_t<T>* smem) {
#pragma unroll
  for (int e = 0; e < elements_per_thread<N>; ++e) reg[e] = detail::widen(smem[fft.transform_in(e) * N + fft.index_in(e)]);
}

// Register file -> LDS in natural order (reference: ThunderFFT_reg2smem, thunderfft.cuh:86-89):
// reg[e] is bin fft.index_out(e) of transform fft.transform_out(e). Every thread must be past its
// last read of `smem` (a __syncthreads() after smem2reg; ThunderFFT_kernel_reg contains one).
template <typename T, int N>
__device__ __forceinline__ void ThunderFFT_reg2smem(const DeviceFFT<N>& fft, vec2_t<T>* smem,
                                                    const float2 (&reg)[elements_per_thread<N>]) {
#pragma unroll
  for (int e = 0; e < elements_per_thread<N>; ++e)
    smem[fft.transform_out(e) * N + fft.index_out(e)] = detail::narrow<T>(reg[e]);
}

// ---- layout-aware staging (reference: layout_t, thunderfft.cuh:11-33, and the sL parameter of its staging calls)
// Element i of signal j lives at idx = i * elem_stride + j * batch_stride, and every pad_period elements are
// followed by `pad` unused ones: slot = idx + (idx / pad_period) * pad. `reversed`: the array holds every signal
// in base-4 digit-reversed order, logical element i = sample reverse_bit_groups<2, log2 N>(i) -- what the
// reference's gather produces (detail/thunderfft_impl_helper.cuh:21-22) and what sample/conv1d/main.cu:54, 87
// writes by hand. Input layouts only; the loaders below undo the order on the way to the registers.
template <int N_, int Batch_, int ElemStride_, int BatchStride_, int PadPeriod_, int Pad_, bool Reversed_ = false>
struct layout_t {
  static constexpr int N = N_;
  static constexpr int batch = Batch_;
  static constexpr int elem_stride = ElemStride_;
  static constexpr int batch_stride = BatchStride_;
  static constexpr int pad_period = PadPeriod_;
  static constexpr int pad = Pad_;
  static constexpr bool reversed = Reversed_;
  static_assert(PadPeriod_ > 0, "pad_period");
  static_assert((N_ & (N_ - 1)) == 0, "N must be a power of two");
  // slot of logical element i of signal j
  __host__ __device__ static constexpr int slot(int i, int j) {
    const int idx = i * elem_stride + j * batch_stride;
    return idx + (idx / pad_period) * pad;
  }
  // logical element that holds sample n, and the sample logical element i holds
  __host__ __device__ static constexpr int element_of_sample(int n) {
    return Reversed_ ? ::thunderfft_unreverse_base4<__builtin_ctz(N_)>(n) : n;
  }
  __host__ __device__ static constexpr int sample_of_element(int i) {
    return Reversed_ ? ::reverse_bit_groups<2, __builtin_ctz(N_)>(i) : i;
  }
  // elements the array needs
  static constexpr int slots = slot(N_ - 1, Batch_ - 1) + 1;
};
template <int N>
using natural_layout = layout_t<N, batch_per_block<N>, 1, N, N * batch_per_block<N>, 0, false>;
// The layouts the bench uses (reference: bench_layout, thunderfft_layout_tuned.h:6-97, tuned pads per length).
// The register maps here read and write an unpadded natural-order array without bank conflicts (a thread's
// points are N / R apart, neighbouring lanes hold neighbouring points), so both are the natural layout.
template <typename T, int N, int BPB = batch_per_block<N>>
struct bench_layout {
  using L_in = layout_t<N, BPB, 1, N, N * BPB, 0, false>;
  using L_out = L_in;
};

namespace detail {
// Thread id within the block; blocks may be (threads) or (threads_per_warp, warps) like the reference's launches.
__device__ __forceinline__ int linear_tid() { return threadIdx.x + threadIdx.y * blockDim.x; }
}  // namespace detail

// Global memory -> LDS under layout sL (reference: ThunderFFT_gmem2smem<T, sL>(smem, gmem), thunderfft.cuh:82-86,
// body detail/thunderfft_impl_helper.cuh:2-35; same template and argument list, `valid` is an addition).
// A reversed layout gathers sample reverse_bit_groups(i) into element i, as the reference does.
template <typename T, typename sL>
__device__ __forceinline__ void ThunderFFT_gmem2smem(vec2_t<T>* __restrict__ smem, const vec2_t<T>* __restrict__ gmem,
                                                     int valid = sL::batch) {
  static_assert(sL::batch == batch_per_block<sL::N>, "one layout per workgroup tile");
  for (int q = detail::linear_tid(); q < sL::N * sL::batch; q += block_threads<sL::N>) {
    const int i = q % sL::N, j = q / sL::N;
    smem[sL::slot(i, j)] =
        q < valid * sL::N ? gmem[j * sL::N + sL::sample_of_element(i)] : detail::narrow<T>(make_float2(0.f, 0.f));
  }
}
// LDS -> global memory (reference: ThunderFFT_smem2gmem<T, sL>(gmem, smem), thunderfft.cuh:89-93). Output layouts
// are never reversed (the reference's reversed branch is empty, detail/thunderfft_impl_helper.cuh:56-58).
template <typename T, typename sL>
__device__ __forceinline__ void ThunderFFT_smem2gmem(vec2_t<T>* __restrict__ gmem, const vec2_t<T>* __restrict__ smem,
                                                     int valid = sL::batch) {
  static_assert(!sL::reversed, "output layouts hold natural order");
  for (int q = detail::linear_tid(); q < sL::N * sL::batch; q += block_threads<sL::N>)
    if (q < valid * sL::N) gmem[q] = smem[sL::slot(q % sL::N, q / sL::N)];
}
template <typename T, typename sL>
__device__ __forceinline__ void ThunderFFT_smem2reg(const DeviceFFT<sL::N>& fft, float2 (&reg)[elements_per_thread<sL::N>],
                                                    const vec2_t<T>* smem) {
#pragma unroll
  for (int e = 0; e < elements_per_thread<sL::N>; ++e)
    reg[e] = detail::widen(smem[sL::slot(sL::element_of_sample(fft.index_in(e)), fft.transform_in(e))]);
}
template <typename T, typename sL>
__device__ __forceinline__ void ThunderFFT_reg2smem(const DeviceFFT<sL::N>& fft, vec2_t<T>* smem,
                                                    const float2 (&reg)[elements_per_thread<sL::N>]) {
  static_assert(!sL::reversed, "output layouts hold natural order");
#pragma unroll
  for (int e = 0; e < elements_per_thread<sL::N>; ++e)
    smem[sL::slot(fft.index_out(e), fft.transform_out(e))] = detail::narrow<T>(reg[e]);
}

// ---- the reference's signatures --------------------------------------------------------------------------------
// The calls below take exactly the template and argument lists of thunderfft.cuh:55-93, 128-130: no transform
// object, registers of the storage type vec2_t<T>. What the reference keeps in module-scope symbols lives in
// module-scope symbols here too: the table addresses ThunderFFTInitialize<T>(N) published to this translation unit
// (detail::tu_twiddle_table; reference: the __constant__ pointers of detail/thunderfft_impl_twiddle.cuh:42-50). A
// length that was never published -- Initialize ran before this unit was loaded, or not at all -- has its few
// twiddle bases evaluated in the kernel instead (the reference evaluates all of its twiddles that way,
// detail/util_arith.h:85-91), so the calls are correct either way. Arithmetic is fp32 for every T; 16-bit registers
// are widened on entry and rounded once on exit of each call.
//
// Register e of a thread holds sample index_in<N>(e) of the block's transform transform_in<N>(e) before
// ThunderFFT_kernel_reg and bin index_out<N>(e) of transform transform_out<N>(e) after it; where
// is_chainable<N>, the two maps agree and a forward result is valid inverse input as it stands (the reference
// needs reverse_data / swap_thread_data there, sample/conv1d/conv1d_fullfft_l4096_thunder.cu:187).
namespace detail {
template <class P>
using split_of = tfft::Plan<P::N, P::E, P::R1, P::R2, P::R3, P::WG, P::PS1, P::PW1, P::PS2, P::PW2, P::MINW, tfft::kFlagSplit>;
// 16-bit T: the LDS exchanges move real parts, then imaginary parts, through N floats per transform, so that the
// workspace is as large as the caller's staging array of vec2_t<T> (4 bytes per point) and may be that array.
template <typename T, int N> struct ref_plan { using type = split_of<typename device_plan<N>::type>; };
template <int N> struct ref_plan<float, N> { using type = typename device_plan<N>::type; };
template <typename T, int N>
using RefFFT = tfft::BlockFFT<typename ref_plan<T, N>::type, true>;

namespace {  // one copy per translation unit
__device__ const float2* tu_twiddle_table[3][16];  // [dtype][log2 n]; null = not published

inline void publish_tables(int dtype, int n) {
  if (dtype < 0 || dtype > 2) return;
  constexpr size_t kRow = sizeof(const float2*) * 16;
  hipError_t rc = hipSuccess;
  if (n == 0) {
    const float2* none[16] = {};
    rc = hipMemcpyToSymbol(HIP_SYMBOL(tu_twiddle_table), none, kRow, dtype * kRow);
  } else if (n >= 64 && n <= 16384 && (n & (n - 1)) == 0) {
    const void* table = nullptr;
    if (tfft_get_twiddle_table(dtype, n, &table) != TFFT_OK) return;
    rc = hipMemcpyToSymbol(HIP_SYMBOL(tu_twiddle_table), &table, sizeof table,
                           dtype * kRow + __builtin_ctz(n) * sizeof table);
  }
  // a unit whose kernels never use the symbol may not carry it: those kernels evaluate their twiddles
  if (rc != hipSuccess) (void)hipGetLastError();
}
[[maybe_unused]] const bool tu_publisher_registered = (table_publishers().push_back(&publish_tables), true);
}  // namespace

template <typename T, int N>
__device__ __forceinline__ RefFFT<T, N> ref_fft(void* workspace, bool with_table) {
  RefFFT<T, N> fft(with_table ? tu_twiddle_table[dtype_code<T>()][__builtin_ctz(N)] : nullptr,
                   static_cast<char*>(workspace));
  fft.tid = linear_tid();
  fft.hi = (fft.tid >> 5) & 1;
  return fft;
}
}  // namespace detail

// Bytes of LDS ThunderFFT_kernel_reg<T, N, ...> exchanges through; sizeof(vec2_t<T>) * (N + pad_h(N)) *
// batch_per_block<N> (the reference's shared-memory size, detail/thunderfft_impl_kernel.cuh:18) covers it.
template <typename T, int N> inline constexpr int kernel_workspace_bytes = detail::ref_plan<T, N>::type::LDS_BYTES;

// Which sample / bin a register holds (see above). All threads of the block, any block shape.
template <int N> __device__ __forceinline__ int index_in(int e) { return detail::ref_fft<float, N>(nullptr, false).index_in(e); }
template <int N> __device__ __forceinline__ int transform_in(int e) { return detail::ref_fft<float, N>(nullptr, false).transform_in(e); }
template <int N> __device__ __forceinline__ int index_out(int e) { return detail::ref_fft<float, N>(nullptr, false).index_out(e); }
template <int N> __device__ __forceinline__ int transform_out(int e) { return detail::ref_fft<float, N>(nullptr, false).transform_out(e); }

// Global memory -> register file (reference: ThunderFFT_gmem2reg<T, N, batch>(reg, gmem), thunderfft.cuh:55-58).
// gmem: the block's batch whole transforms, [batch][N], 16-byte aligned. 16-byte loads, no LDS.
template <typename T, int N, int batch>
__device__ __forceinline__ void ThunderFFT_gmem2reg(vec2_t<T>* __restrict__ reg, const vec2_t<T>* __restrict__ gmem) {
  static_assert(batch == batch_per_block<N>, "batch is batch_per_block<N>");
  using IO = typename device_io<T>::type;
  const auto fft = detail::ref_fft<T, N>(nullptr, false);
  typename IO::raw raw[elements_per_thread<N> / 2];
  float2 v[elements_per_thread<N>];
  fft.template load<IO>(reinterpret_cast<const typename IO::raw*>(gmem), batch, raw);
  fft.template unpack<IO>(raw, v);
#pragma unroll
  for (int e = 0; e < elements_per_thread<N>; ++e) reg[e] = detail::narrow<T>(v[e]);
}

// Register file -> global memory (reference: ThunderFFT_reg2gmem<T, N, batch>(gmem, reg), thunderfft.cuh:61-65).
template <typename T, int N, int batch>
__device__ __forceinline__ void ThunderFFT_reg2gmem(vec2_t<T>* __restrict__ gmem, const vec2_t<T>* __restrict__ reg) {
  static_assert(batch == batch_per_block<N>, "batch is batch_per_block<N>");
  using IO = typename device_io<T>::type;
  const auto fft = detail::ref_fft<T, N>(nullptr, false);
  float2 v[elements_per_thread<N>];
#pragma unroll
  for (int e = 0; e < elements_per_thread<N>; ++e) v[e] = detail::widen(reg[e]);
  fft.template store<IO>(reinterpret_cast<typename IO::raw*>(gmem), batch, v);
}

// Shared memory -> register file (reference: ThunderFFT_smem2reg<T, sL>(reg, smem), thunderfft.cuh:68-72).
template <typename T, typename sL>
__device__ __forceinline__ void ThunderFFT_smem2reg(vec2_t<T>* __restrict__ reg, const vec2_t<T>* __restrict__ smem) {
  static_assert(sL::batch == batch_per_block<sL::N>, "one layout per workgroup tile");
  const auto fft = detail::ref_fft<T, sL::N>(nullptr, false);
#pragma unroll
  for (int e = 0; e < elements_per_thread<sL::N>; ++e)
    reg[e] = smem[sL::slot(sL::element_of_sample(fft.index_in(e)), fft.transform_in(e))];
}

// Register file -> shared memory (reference: ThunderFFT_reg2smem<T, sL>(smem, reg), thunderfft.cuh:75-79).
template <typename T, typename sL>
__device__ __forceinline__ void ThunderFFT_reg2smem(vec2_t<T>* __restrict__ smem, const vec2_t<T>* __restrict__ reg) {
  static_assert(sL::batch == batch_per_block<sL::N>, "one layout per workgroup tile");
  static_assert(!sL::reversed, "output layouts hold natural order");
  const auto fft = detail::ref_fft<T, sL::N>(nullptr, false);
#pragma unroll
  for (int e = 0; e < elements_per_thread<sL::N>; ++e)
    smem[sL::slot(fft.index_out(e), fft.transform_out(e))] = reg[e];
}

// Register-resident transform (reference: ThunderFFT_kernel_reg<T, N, batch, forward>(reg, W, workspace),
// thunderfft.cuh:127-130). workspace: kernel_workspace_bytes<T, N> bytes of the block's LDS, 16-byte aligned; it
// may alias the staging array the registers were read from (every thread passes a barrier before the first
// write). W is accepted for the reference's argument list and not read: the reference's fp32 kernels ignore it
// as well (its bench passes an unset array, sample/bench/baseline/thunderfft_bench.h:26-29, 43), and the per-lane
// matrix fragments it carries for fp16 have no counterpart in this transform. Unnormalised in both directions.
// Follow with __syncthreads() before the workspace is written by anything else, as the reference's samples do.
template <typename T, int N, int batch, bool forward>
__device__ __forceinline__ void ThunderFFT_kernel_reg(vec2_t<T>* __restrict__ reg, vec2_t<T>* __restrict__ W,
                                                      void* workspace) {
  static_assert(batch == batch_per_block<N>, "batch is batch_per_block<N>");
  (void)W;
  const auto fft = detail::ref_fft<T, N>(workspace, true);
  float2 v[elements_per_thread<N>];
#pragma unroll
  for (int e = 0; e < elements_per_thread<N>; ++e) v[e] = detail::widen(reg[e]);
  fft.template transform<forward>(v, true);
#pragma unroll
  for (int e = 0; e < elements_per_thread<N>; ++e) reg[e] = detail::narrow<T>(v[e]);
}

// Reference: unit_fp16::make_reg_b_precompute<N, forward>(W) loads a lane's 28 (N <= 128) or 36 matrix fragments
// for its tensor-core stages (detail/kernel/unit_kernel_fp16.cuh:97-108). The array has no reader here (see
// ThunderFFT_kernel_reg); it is set to zero so that kernels written against the reference pass defined values.
namespace unit_fp16 {
template <int N, bool forward>
__device__ __forceinline__ void make_reg_b_precompute(vec2_t<__half>* W) {
  constexpr int entries = (N == 64 || N == 128) ? 28 : 36;
#pragma unroll
  for (int i = 0; i < entries; ++i) W[i] = __half2{};
}
}  // namespace unit_fp16

// Device address of the table ThunderFFTInitialize<T>(N) built on the current device.
template <typename T>
inline const float2* twiddle_table(int N) {
  const void* p = nullptr;
  check(tfft_get_twiddle_table(dtype_code<T>(), N, &p), "twiddle_table");
  return static_cast<const float2*>(p);
}

}  // namespace thunderfft

// ---- global-namespace helpers the reference's samples use beside the device API -------------------------------
// (detail/util_arith.h:16-30, 123-135, 148-167). They describe the reference's own register and shared-memory
// layouts; here they keep their arithmetic meaning so that sample code that calls them still compiles and computes
// the same indices.

// Shared-memory padding, in points, behind one N-point transform: the reference sizes its dynamic shared memory as
// sizeof(vec2_t<T>) * (N + pad_h(N)) * batch_per_block<N> (detail/thunderfft_impl_kernel.cuh:18, detail/util_arith.h:
// 123-135). Here the same formula covers, for every T, both a natural-order staging array and the workspace of
// ThunderFFT_kernel_reg<T, N, ...> (one exchange slot per point: 8 bytes for float, 4 for the 16-bit types).
constexpr int pad_h(int N) {
  constexpr auto points = [](int slots, int n, int bpb) { return (slots + bpb - 1) / bpb - n; };
  using namespace thunderfft;
  switch (N) {
    case 64: return points(device_plan<64>::type::LDS_ELEMS, 64, batch_per_block<64>);
    case 128: return points(device_plan<128>::type::LDS_ELEMS, 128, batch_per_block<128>);
    case 256: return points(device_plan<256>::type::LDS_ELEMS, 256, batch_per_block<256>);
    case 512: return points(device_plan<512>::type::LDS_ELEMS, 512, batch_per_block<512>);
    case 1024: return points(device_plan<1024>::type::LDS_ELEMS, 1024, batch_per_block<1024>);
    case 2048: return points(device_plan<2048>::type::LDS_ELEMS, 2048, batch_per_block<2048>);
    case 4096: return points(device_plan<4096>::type::LDS_ELEMS, 4096, batch_per_block<4096>);
    case 8192: return points(device_plan<8192>::type::LDS_ELEMS, 8192, batch_per_block<8192>);
    case 16384: return points(device_plan<16384>::type::LDS_ELEMS, 16384, batch_per_block<16384>);
    default: return -1;
  }
}

// Transposes the 4 x 4 digit grid of a thread's registers in place, in both halves of its 32 registers:
// element 4a + b trades places with 4b + a, and 16 + 4a + b with 16 + 4b + a.
template <typename T>
__host__ __device__ inline void swap_thread_data(T* thread_data) {
  for (int half = 0; half < 32; half += 16)
    for (int a = 0; a < 4; ++a)
      for (int b = a + 1; b < 4; ++b) {
        const T t = thread_data[half + 4 * a + b];
        thread_data[half + 4 * a + b] = thread_data[half + 4 * b + a];
        thread_data[half + 4 * b + a] = t;
      }
}
