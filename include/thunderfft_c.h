/* C ABI of libthunderfft_hip.so: batched 1D complex-to-complex FFT on MI355X (gfx950).
 *
 * The reference library is header-only CUDA C++ (thunderfft/include/thunderfft/thunderfft.cuh);
 * it has no FFI layer, so the symbols below are what a binding for its three host entry
 * points would call. The C++ templates in include/thunderfft/thunderfft.cuh forward to them.
 *
 * Data contract, identical to the reference (detail/kernel/global_kernel.cuh:37,49 and
 * detail/thunderfft_impl_kernel.cuh:6-30): device pointers, `batch` contiguous transforms of
 * n interleaved (re, im) points each, natural order in and out, unnormalised in both
 * directions, `in == out` allowed.
 *
 * Every function returns 0 on success or a TFFT_ERR_* code; tfft_last_error() returns the
 * message of the calling thread's last failure.
 */
#ifndef THUNDERFFT_C_H_
#define THUNDERFFT_C_H_

#ifdef __cplusplus
extern "C" {
#endif

/* Element type of the interleaved complex data in device memory. Arithmetic is fp32 for all. */
enum { TFFT_F32 = 0, TFFT_F16 = 1, TFFT_BF16 = 2 };

enum {
  TFFT_OK = 0,
  TFFT_ERR_UNSUPPORTED = 1,     /* n, dtype or direction has no kernel */
  TFFT_ERR_NOT_INITIALIZED = 2, /* tfft_initialize(dtype, n) was not called on this device */
  TFFT_ERR_INVALID = 3,         /* null pointer, negative batch, misaligned buffer */
  TFFT_ERR_HIP = 4              /* a HIP runtime call failed; message holds hipGetErrorString */
};

/* Replaces thunderfft::ThunderFFTInitialize<T>(int N) (thunderfft.cuh:39-40; body
 * detail/thunderfft_impl_twiddle.cuh:261-311): builds the twiddle table of length n in fp64
 * on the host, rounds once to fp32 and uploads it to the current device. Tables are cached
 * per (device, dtype, n); calling again with the same arguments is a no-op. Allocates.
 * Thread safety: every entry point may be called from any thread. Calls that launch hold the plan cache shared
 * from their look-up to their launch; tfft_initialize and tfft_finalize hold it exclusively. */
int tfft_initialize(int dtype, int n);

/* Replaces thunderfft::ThunderFFTFinalize<T>() (thunderfft.cuh:43-44; body
 * thunderfft_impl_twiddle.cuh:342-349): frees every table of `dtype` on every device. It first waits for the
 * work already queued on those devices (kernels launched earlier may still read the tables), so it may run
 * concurrently with tfft_exec_* calls of other threads: each of those either launched before the tables went away
 * or returns TFFT_ERR_NOT_INITIALIZED. Do not call it while a stream of an affected device is being captured. */
int tfft_finalize(int dtype);

/* Replaces thunderfft::ThunderFFT_global<T, N, forward>(d_input, d_output, batch, stream)
 * (thunderfft.cuh:98-102; body detail/thunderfft_impl_kernel.cuh:6-30). n is a run-time
 * argument here. The launch is asynchronous on `stream` (a hipStream_t, may be NULL); the
 * reference's trailing cudaStreamSynchronize is left to the caller (the C++ wrapper does it).
 * n is a power of two: 64 ... 16384 run one LDS-resident kernel per launch (all three element
 * types); 2^15 ... 2^30 (fp32) run 2 or 3 strided sweeps through a scratch buffer that
 * tfft_initialize allocated (four-step; the reference has no transform beyond 4096). That scratch
 * is one buffer per (device, n): multi-pass calls that share it are ordered on the device by an
 * event, so calls from several streams or threads stay correct but do not overlap; pass a
 * workspace of your own (tfft_exec_c2c_opt) for calls that should. A multi-pass call made under stream capture
 * is ordered inside its graph only: replays of that graph are not ordered against other users of the
 * library-owned scratch (eager calls, other graphs) -- captured callers use tfft_exec_c2c_static or a workspace.
 * Non-finite inputs: a NaN or infinity stays inside its own transform, with two exceptions, both for 16-bit storage
 * on the default plan. At 512 <= n <= 2048, 2 ... 8 transforms go through one matrix product per 4096-point tile, with
 * explicit zeros between them, so a non-finite value also turns the other transforms of its tile (the 4096 / n
 * consecutive transforms starting at a multiple of 4096 / n) non-finite. At n = 64 and 128 the same holds for the
 * 256 / n consecutive transforms starting at a multiple of 256 / n (one 256-point row of the tile). tfft_exec_options.plan = 3
 * (the fp32-VALU kernels) keeps every transform independent. */
int tfft_exec_c2c(int dtype, int n, int forward, const void* d_in, void* d_out, long long batch,
                  void* stream);

/* 1 if tfft_exec_c2c has a kernel for (dtype, n, forward), else 0. The reference expresses the
 * same through missing template specialisations (no inverse for N = 512 and 2048,
 * detail/kernel/shared_kernel_n2048_b1.cuh:121-145). */
int tfft_is_supported(int dtype, int n, int forward);

/* Launch geometry of the kernel behind (dtype, n): the counterpart of the reference's
 * batch_per_block<N>, warp_per_block<N> and shared-memory size (detail/thunderfft_impl.cuh:14-32,
 * detail/thunderfft_impl_kernel.cuh:18). */
typedef struct tfft_plan_info {
  int n;
  int threads_per_block;
  int transforms_per_block;
  int points_per_thread;
  int radix[3];
  int lds_bytes;
  int blocks_per_cu; /* occupancy query result on the current device, 0 if not initialised */
} tfft_plan_info;
int tfft_get_plan_info(int dtype, int n, tfft_plan_info* info);

/* Per-call launch options (no reference counterpart: the reference fixes its launch shape at compile time,
 * detail/thunderfft_impl.cuh:14-32). Every combination gives correct results; the fields choose between kernels
 * and launch shapes kept for A/B timing. Fill with tfft_exec_options_init, change what is needed, pass to a
 * *_opt entry point; NULL there means the defaults. Nothing here is process-wide state.
 *   plan               : 0 = the default kernel of (dtype, n); 3 and up = its alternative kernels (another
 *                        factorisation, workgroup size or LDS exchange form). 16-bit storage runs on
 *                        the matrix cores by default at every length (n = 64 and 128: plans 0 and 12; 3 and 4 the VALU kernels); there 3 (and 4) are the fp32-VALU kernels and 5-12 other
 *                        forms of the matrix-core kernel: 6 = one workgroup per tile with the twiddle table,
 *                        10 = one workgroup per tile with incrementally built twiddles (fp16 at n = 8192, 16384:
 *                        the other of the two pass-1 twiddle forms, fp32 chains or an fp16 table), 11 = resident grid,
 *                        12 = plan 10 with one LDS image (n <= 4096); 5, 7 = one LDS image (n = 4096); 8 = split
 *                        bf16 matrix (bf16, n = 4096). A plan a length does not have returns
 *                        TFFT_ERR_UNSUPPORTED. Multi-pass lengths (fp32, n > 16384): 3, 4, 5 = 128 KiB tiles
 *                        (twice the transforms per tile) for the 256- and 512-point sweeps, the 512-point sweeps
 *                        only, the 256-point sweeps only; same results bit for bit. For the convolution entry points (fp16 storage): 0 = one workgroup
 *                        per run of 2 ... 8 signals of one filter, the next signal travelling into LDS under the
 *                        current one's passes (small problems fall back to plan 4); 3 = the fp32-VALU kernels;
 *                        4 = one workgroup per signal; 5, 6 = the plan-0 kernel with runs of 2 / 8 whatever the size;
 *                        7 = runs of 4 with per-wave staging (two workgroup barriers fewer per signal).
 *   xcd_run_log2       : -1 = the kernel's default; L >= 0 = every XCD takes runs of 2^L consecutive chunks of
 *                        the batch (0 = chunk order equals launch order); results do not depend on it
 *   grid_blocks_per_cu : resident grid of value * CU count workgroups that walk the batch (0 = default: one
 *                        workgroup per chunk of transforms, or the resident size of a kernel that is resident
 *                        by default). For the fp16 convolution: workgroups per CU of the resident grid of its
 *                        one-workgroup-per-signal kernel (plans 0 and 4; 0 = no resident grid).
 *   grid_blocks        : resident grid of exactly value workgroups (overrides the per-CU form)
 *   extra_lds          : bytes of unused dynamic LDS per workgroup, to lower occupancy
 *   workspace(+_bytes) : multi-pass lengths (fp32, n > 16384): caller-owned scratch of at least n points
 *                        (8 n bytes; tfft_workspace_bytes gives the minimum and the size that takes a batch in
 *                        one round). With it the call touches no library-owned scratch, so calls on different
 *                        streams with different workspaces run concurrently. NULL = the scratch
 *                        tfft_initialize allocated, shared per (device, n): calls that use it are ordered one
 *                        after the other on the device, whatever their streams. */
typedef struct tfft_exec_options {
  int struct_size; /* sizeof(tfft_exec_options); set by tfft_exec_options_init */
  int plan;
  int xcd_run_log2;
  int grid_blocks_per_cu;
  long long grid_blocks;
  int extra_lds;
  int reserved;
  void* workspace;
  unsigned long long workspace_bytes;
} tfft_exec_options;
void tfft_exec_options_init(tfft_exec_options* opt);

/* tfft_exec_c2c with per-call options (opt == NULL: exactly tfft_exec_c2c). */
int tfft_exec_c2c_opt(int dtype, int n, int forward, const void* d_in, void* d_out, long long batch,
                      void* stream, const tfft_exec_options* opt);

/* tfft_exec_c2c for callers that cannot allocate device memory (code under HIP graph capture, tight serving
 * loops, harnesses that forbid hipMalloc): same arguments, same results bit for bit, and neither this call nor
 * anything it launches allocates or frees. It has no counterpart in the reference. Multi-pass lengths
 * (fp32, n = 2^15 ... 2^20) run their sweeps through a scratch of 2^20 points (8 MiB) that is part of the
 * library's code object: every GPU the library runs on holds its own copy until the process ends. Longer
 * transforms return TFFT_ERR_UNSUPPORTED (use tfft_exec_c2c or a workspace of your own); batches of any size
 * are folded through the scratch 2^20 / n transforms at a time. n <= 16384 (any dtype) needs no scratch and is
 * forwarded to the ordinary kernels. tfft_initialize(dtype, n) must have run on the device (it builds the
 * tables). All calls on one GPU share that GPU's scratch: calls that overlap in time on one GPU (different
 * streams, or threads) race; order them with events. */
int tfft_exec_c2c_static(int dtype, int n, int forward, const void* d_in, void* d_out, long long batch,
                         void* stream);

/* Scratch a multi-pass transform needs: *min_bytes for one transform at a time, *best_bytes to take `batch`
 * transforms in as few rounds as the library-owned scratch would (both 0 for n <= 16384). Either pointer may
 * be NULL. No reference counterpart. */
int tfft_workspace_bytes(int dtype, int n, long long batch, unsigned long long* min_bytes,
                         unsigned long long* best_bytes);

/* Process-wide switches (no reference counterpart). The only key is
 *   "sync" : 1 = synchronise the stream after every launch, the reference's behaviour
 *            (detail/thunderfft_impl_kernel.cuh:29); 0 (default) = asynchronous.
 * Kernel and launch-shape choices are per call (tfft_exec_options), never process-wide. */
int tfft_set_option(const char* key, long long value);

/* MEASUREMENT ONLY -- the output is meaningless. Runs the default kernel of (dtype, n) with part of its work
 * compiled out, to price the rest: pattern 1 = the global loads and stores alone (the rows are read and written
 * in the transform's access pattern), pattern 2 = loads, butterflies and stores without the LDS exchanges
 * (fp32, n <= 16384, lengths that have the build; for fp32 n = 8192 the builds are those of the one-pass
 * 32 x 16 x 16 kernel, plan 3, not of the default two-transform kernel); patterns 3 and 4 = the matrix-core kernel of a 16-bit length
 * (fp16 / bf16, n = 256 ... 4096) without its stores / without its loads: what the read stream and the write
 * stream of that kernel cost each on its own. Kept apart from tfft_exec_c2c so that no option can make a
 * transform call return wrong data. No reference counterpart (its bench prices on-chip time with in-kernel
 * repeats instead, sample/bench/util/helper.h:130-136). */
int tfft_bench_access_pattern(int dtype, int n, int forward, int pattern, const void* d_in, void* d_out,
                              long long batch, void* stream, const tfft_exec_options* opt);

/* Twiddle step between the two halves of a four-step transform that run as separate batched
 * calls (distributed transforms: local FFTs, all-to-all, local FFTs): element (r, c) of the
 * row-major [rows][cols] block is multiplied by W_n^((row0 + r) * (col0 + c)), W_n = exp(-2 pi i / n),
 * or by its conjugate when forward == 0. cols must be even. No counterpart in the reference,
 * whose longest transform fits one block (detail/thunderfft_impl.cuh:14-32). fp32 only. */
int tfft_fourstep_twiddle(int dtype, void* d_data, long long rows, long long cols, long long row0,
                          long long col0, long long n, int forward, void* stream);

/* MEASUREMENT ONLY -- the output is meaningless (values overflow). The default matrix-core kernel of a 16-bit
 * length (fp16 / bf16, n = 256 ... 4096) with its three passes run `repeats` times per tile on chip (registers ->
 * passes -> natural-order LDS image -> registers) between one load and one store of the tile. Two launches with R
 * and 2R repeats give the on-chip time per pass, (t(2R) - t(R)) / R: the reference's compute_ms protocol for its
 * half-precision kernels (sample/bench/main.cu:52-57, baseline/thunderfft_bench.h:139-182, util/helper.h:130-136). */
int tfft_bench_onchip_repeats(int dtype, int n, int repeats, const void* d_in, void* d_out, long long batch,
                              void* stream);

/* Local passes of the distributed four-step transform (one large transform over the ranks of a process group:
 * local transforms, all-to-all exchanges, and between them the two passes below). No counterpart in the
 * reference, which is single-GPU throughout (sample/fft1d/main.cu:80). fp32 interleaved complex, offsets and
 * strides in points. Asynchronous on `stream`; nothing is allocated.
 *
 * tfft_dist_permute: for b1 < nb1, b2 < nb2, i < ni, j < nj
 *     transpose = 1:  out[b1 out_b1 + b2 out_b2 + j out_j + i] = w * in[b1 in_b1 + b2 in_b2 + i in_i + j]
 *     transpose = 0:  out[b1 out_b1 + b2 out_b2 + i out_j + j] = w * in[b1 in_b1 + b2 in_b2 + i in_i + j]
 * with w = 1, or when twiddle = 1 the four-step factor w = W_n^(p q), W_n = exp(-2 pi i / tw_n) (conjugated when
 * forward == 0), p = tw_p0 + tw_pi i, q = tw_q0 + tw_qb1 b1 + tw_qb2 b2 + tw_qj j, taken from the two-level table
 * W_n^e = hi[e >> tw_shift] * lo[e & (2^tw_shift - 1)] (hi: n >> tw_shift entries W_n^(k << tw_shift); lo:
 * 2^tw_shift entries W_n^k; float2 (cos, -sin), built by the caller). Extents and strides must be even (16-byte
 * accesses); in and out must not overlap. */
typedef struct tfft_dist_permute_args {
  long long nb1, nb2, ni, nj;
  long long in_b1, in_b2, in_i;
  long long out_b1, out_b2, out_j;
  int transpose, twiddle;
  long long tw_n;
  int tw_shift, reserved;
  long long tw_p0, tw_pi, tw_q0, tw_qb1, tw_qb2, tw_qj;
} tfft_dist_permute_args;
int tfft_dist_permute(const void* d_in, void* d_out, const tfft_dist_permute_args* args, const void* d_tw_hi,
                      const void* d_tw_lo, int forward, void* stream);

/* tfft_dist_radix: a rank that holds `parts` chunks of `rows` rows of row_len points (chunk s = the part of every
 * row that rank s contributed) turns them into `parts` batches of rows whose own row_len-point transforms
 * complete the (parts * row_len)-point transform of every row:
 *     out[k out_chunk + r out_pitch + t] = W_M^(t k) * sum_s W_parts^(s k) in[s in_chunk + r in_pitch + t],
 * k < parts, r < rows, t < row_len, M = parts * row_len, d_table = W_M^e, e < M (float2 (cos, -sin)); conjugated
 * when forward == 0. parts = 2, 4 or 8; row_len a power of two; strides even. In place (d_in == d_out with equal
 * strides) is allowed. */
typedef struct tfft_dist_radix_args {
  int parts, reserved;
  long long rows, row_len;
  long long in_chunk, in_pitch, out_chunk, out_pitch;
} tfft_dist_radix_args;
int tfft_dist_radix(const void* d_in, void* d_out, const tfft_dist_radix_args* args, const void* d_table, int forward,
                    void* stream);

/* Fused circular convolution of length n through the frequency domain, one kernel per call:
 *   y[b][d] = IDFT_n( DFT_n(x[b][d]) * H[d] ),  b < B, d < D,  unnormalised (scale h by 1/n beforehand,
 * as the reference does, sample/conv1d/conv1d_fullfft_l4096_thunder.cu:352-361).
 * Replaces thunder_conv4096_block_kernel, thunder_conv4096_block_kernel_realin and
 * thunder_conv4096_block_kernel_realin_realout (same file, :148-292): signals are [B][D][n]
 * contiguous and signal (b, d) uses filter d, i.e. filter = signal index % D (:171).
 *   kind TFFT_CONV_COMPLEX     : x and y interleaved complex of `dtype`
 *   kind TFFT_CONV_REAL_IN     : x real (n values per signal), y interleaved complex
 *   kind TFFT_CONV_REAL_IN_OUT : x real, y real; the filter taps must be real (two signals then share
 *                                one complex transform; complex taps give wrong results)
 * d_h_perm holds the D filter spectra in the kernel's register order, made by
 * tfft_conv_filter_layout. Supported: n = 4096; fp16 (all kinds), fp32 (complex kind).
 * Needs tfft_initialize(dtype, n). Asynchronous on `stream`; allocates nothing. */
enum { TFFT_CONV_COMPLEX = 0, TFFT_CONV_REAL_IN = 1, TFFT_CONV_REAL_IN_OUT = 2 };
int tfft_conv_exec(int dtype, int n, int kind, const void* d_x, const void* d_h_perm, void* d_y, int B, int D,
                   void* stream);

/* Replaces permute_filter_layout4096 (sample/conv1d/conv1d_fullfft_l4096_thunder.cu:120-146):
 * reorders D natural-order filter spectra ([D][n] interleaved complex of `dtype`, e.g. the output of
 * tfft_exec_c2c on the zero-padded filters) into the order tfft_conv_exec reads. d_h_perm must not
 * alias d_h_freq. */
int tfft_conv_filter_layout(int dtype, int n, const void* d_h_freq, void* d_h_perm, int D, void* stream);

/* The two steps of the reference's filter_fft_once (sample/conv1d/conv1d_fullfft_l4096_thunder.cu:412-426: forward
 * transform of the zero-padded taps, then permute_filter_layout4096) in ONE kernel: d_h_perm <- DFT_n(d_taps) in the
 * order tfft_conv_exec reads, bit for bit what tfft_exec_c2c followed by tfft_conv_filter_layout writes. d_taps:
 * D zero-padded filters [D][n] interleaved complex of `dtype`; must not alias d_h_perm. Exists where the default
 * convolution kernel runs on the matrix cores (fp16, n = 4096); TFFT_ERR_UNSUPPORTED otherwise (use the two calls).
 * Needs tfft_initialize(dtype, n). Asynchronous on `stream`; allocates nothing. */
int tfft_conv_filter_spectra(int dtype, int n, const void* d_taps, void* d_h_perm, int D, void* stream);

/* The two convolution calls with per-call options (opt == NULL: exactly the calls above). The matrix-core kernels
 * (plans 0, 4, 5, 6, 7) and the fp32-VALU kernels (plan 3) read the permuted filters in different orders: the library remembers which order each
 * d_h_perm buffer was last written in and tfft_conv_exec(_opt) returns TFFT_ERR_INVALID when asked to read a
 * buffer with the other kernel. */
int tfft_conv_exec_opt(int dtype, int n, int kind, const void* d_x, const void* d_h_perm, void* d_y, int B, int D,
                       void* stream, const tfft_exec_options* opt);
int tfft_conv_filter_layout_opt(int dtype, int n, const void* d_h_freq, void* d_h_perm, int D, void* stream,
                                const tfft_exec_options* opt);

/* The order check above knows only what tfft_conv_filter_layout(_opt) wrote: it keeps the byte range of each of
 * its last 256 outputs per process and compares any d_h_perm that overlaps one. A range whose contents the caller
 * replaced by other means (a copy of a saved buffer, an allocator handing the address out again) is stale:
 * tfft_conv_forget_layout drops what is recorded for [d_buffer, d_buffer + bytes) on the current device (bytes = 0:
 * any record that contains d_buffer). tfft_finalize(dtype) drops all records of dtype. No reference counterpart. */
int tfft_conv_forget_layout(const void* d_buffer, unsigned long long bytes);

/* Overlap-save tiled convolution with transforms of length n: for every tile t,
 *   out[t*valid + j] = IDFT_n( DFT_n(in[t*valid .. t*valid + n)) * hf )[j],   j < valid, t*valid + j < output_len,
 * input points at or past input_len read as zero, unnormalised. Replaces conv1d_direct_fft_kernel
 * and its launch (sample/conv1d/main.cu:38-129, 294-299), where valid = n - filter_len + 1,
 * output_len = input_len - filter_len + 1 and hf is the spectrum of the taps stored at
 * (n - i) % n and scaled by 1/n (:145-154), which makes the result the valid cross-correlation
 * sum_k in[i + k] * filter[k] (:156-174). in, out: interleaved complex of `dtype` (fp32 or fp16);
 * hf: n points of the same type, natural order. n = 64. Needs tfft_initialize(dtype, n).
 * Asynchronous on `stream`; allocates nothing. in and out must not overlap. */
int tfft_conv_tiles_exec(int dtype, int n, const void* d_in, const void* d_hf, void* d_out, long long input_len,
                         long long output_len, int valid, void* stream);

/* Device address of the twiddle table tfft_initialize(dtype, n) built on the current device:
 * n float2 entries W_n^k = (cos, -sin)(2 pi k / n). For user kernels built on the device-level API
 * (include/thunderfft/thunderfft_device.hpp), which take the table as an argument the way the
 * reference's ThunderFFT_kernel_reg takes W (thunderfft.cuh:127-130). n <= 16384. The table stays
 * valid until tfft_finalize(dtype). */
int tfft_get_twiddle_table(int dtype, int n, const void** d_table);

const char* tfft_last_error(void);

#ifdef __cplusplus
}
#endif
#endif /* THUNDERFFT_C_H_ */
