// Fused FFT-convolution benchmark, N = 4096, fp16 storage: the MI355X build of the reference's
// thff_conv1d_fullfft4096 (thunderfft/sample/conv1d/conv1d_fullfft_l4096_thunder.cu:304-537).
//
// Same 11 positional arguments in the same order (:305-327)
//   K device B D warmup runs include_filter_fft validate filter_fft_mode real_input real_filter
// and the same first stdout line (:452-455), which Figure5_1dFFT/benchmark_figure5_1dfft.py:27-34
// parses; the validation block prints the reference's three lines (:516-522). Inputs come from
// std::srand(42) / std::rand() in the reference's order (:335-361), so the data are the same.
// filter_fft_mode 0 uses hipFFT (fp32) for the filter spectra where the reference uses cuFFT,
// mode 1 the library's own fp16-storage transform. Everything on the device goes through the C ABI
// (tfft_exec_c2c, tfft_conv_filter_layout, tfft_conv_exec).
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>

#include <cmath>
#include <cstdlib>
#include <iomanip>
#include <iostream>
#include <vector>

#include <thunderfft/thunderfft.cuh>

using cpx_h = __half2;

namespace {

constexpr int kN = 4096;

#define CHECK_HIPFFT(call)                                                                  \
  do {                                                                                      \
    hipfftResult r_ = (call);                                                               \
    if (r_ != HIPFFT_SUCCESS) {                                                             \
      std::fprintf(stderr, "hipFFT error %s:%d: %d\n", __FILE__, __LINE__, (int)r_);        \
      std::exit(EXIT_FAILURE);                                                              \
    }                                                                                       \
  } while (0)

__global__ void f32_to_h2(const float2* __restrict__ in, __half2* __restrict__ out, int n) {
  const int i = int(blockIdx.x) * int(blockDim.x) + int(threadIdx.x);
  if (i < n) out[i] = __floats2half2_rn(in[i].x, in[i].y);
}

// the reference's measure_execution_ms (sample/utils.h:33-57): mean over `runs` back-to-back calls
template <class F>
float measure_execution_ms(F&& fn, unsigned warmup, unsigned runs) {
  hipEvent_t start, stop;
  CHECK_HIP(hipEventCreate(&start));
  CHECK_HIP(hipEventCreate(&stop));
  CHECK_HIP(hipDeviceSynchronize());
  for (unsigned i = 0; i < warmup; ++i) fn();
  CHECK_HIP(hipDeviceSynchronize());
  CHECK_HIP(hipEventRecord(start));
  for (unsigned i = 0; i < runs; ++i) fn();
  CHECK_HIP(hipEventRecord(stop));
  CHECK_HIP(hipDeviceSynchronize());
  float ms = 0.f;
  CHECK_HIP(hipEventElapsedTime(&ms, start, stop));
  CHECK_HIP(hipEventDestroy(start));
  CHECK_HIP(hipEventDestroy(stop));
  return ms / float(runs);
}

void check_tfft(int rc, const char* what) {
  if (rc != TFFT_OK) {
    std::fprintf(stderr, "%s: %s\n", what, tfft_last_error());
    std::exit(EXIT_FAILURE);
  }
}

}  // namespace

int main(int argc, char** argv) {
  int K = 9, device = 0, B = 1, D = 1, warmup = 5, runs = 30;
  int include_filter_fft = 1, validate = 1, filter_fft_mode = 1, real_input = 0, real_filter = 0;
  if (argc >= 2) K = std::atoi(argv[1]);
  if (argc >= 3) device = std::atoi(argv[2]);
  if (argc >= 4) B = std::atoi(argv[3]);
  if (argc >= 5) D = std::atoi(argv[4]);
  if (argc >= 6) warmup = std::atoi(argv[5]);
  if (argc >= 7) runs = std::atoi(argv[6]);
  if (argc >= 8) include_filter_fft = std::atoi(argv[7]);
  if (argc >= 9) validate = std::atoi(argv[8]);
  if (argc >= 10) filter_fft_mode = std::atoi(argv[9]);
  if (argc >= 11) real_input = std::atoi(argv[10]);
  if (argc >= 12) real_filter = std::atoi(argv[11]);

  if (K < 1 || K > kN) {
    std::cerr << "K must be in [1, " << kN << "]\n";
    return EXIT_FAILURE;
  }
  if (B < 1 || D < 1 || runs < 1 || warmup < 0) {
    std::cerr << "B, D and runs must be positive\n";
    return EXIT_FAILURE;
  }

  CHECK_HIP(hipSetDevice(device));
  // tuning knobs outside the reference's argument list
  if (const char* e = std::getenv("TFFT_CONV_RESIDENT_PER_CU")) tfft_set_option("conv_resident_per_cu", std::atoi(e));
  if (const char* e = std::getenv("TFFT_MODE")) tfft_set_option("mode", std::atoi(e));
  std::srand(42);

  const int batch_count = B * D;
  const size_t sig_elems = size_t(batch_count) * kN;
  const size_t filt_elems = size_t(D) * kN;

  std::vector<cpx_h> h_x(sig_elems);
  std::vector<__half> h_x_real(sig_elems, __float2half(0.0f));
  std::vector<float2> h_h_time(filt_elems, make_float2(0.f, 0.f));
  std::vector<cpx_h> h_h_half(filt_elems, __float2half2_rn(0.0f));
  for (size_t i = 0; i < sig_elems; ++i) {
    const float re = std::rand() / float(RAND_MAX);
    const float im = real_input ? 0.0f : (std::rand() / float(RAND_MAX));
    h_x[i] = __floats2half2_rn(re, im);
    h_x_real[i] = __float2half(re);
  }
  for (int d = 0; d < D; ++d) {
    for (int k = 0; k < K; ++k) {
      // taps scaled by 1/N: the unnormalised inverse then returns linear amplitude, and the
      // product in the frequency domain stays inside the fp16 range
      const float re = (std::rand() / float(RAND_MAX)) / float(kN);
      const float im = real_filter ? 0.0f : ((std::rand() / float(RAND_MAX)) / float(kN));
      h_h_time[size_t(d) * kN + k] = make_float2(re, im);
      h_h_half[size_t(d) * kN + k] = __floats2half2_rn(re, im);
    }
  }

  cpx_h *d_x = nullptr, *d_out = nullptr, *d_H = nullptr, *d_H_perm = nullptr, *d_h_half = nullptr;
  __half *d_x_real = nullptr, *d_out_real = nullptr;
  float2 *d_h_time = nullptr, *d_h_freq = nullptr;
  CHECK_HIP(hipMalloc(&d_x, sizeof(cpx_h) * sig_elems));
  CHECK_HIP(hipMalloc(&d_out, sizeof(cpx_h) * sig_elems));
  CHECK_HIP(hipMalloc(&d_H, sizeof(cpx_h) * filt_elems));
  CHECK_HIP(hipMalloc(&d_H_perm, sizeof(cpx_h) * filt_elems));
  CHECK_HIP(hipMalloc(&d_h_half, sizeof(cpx_h) * filt_elems));
  CHECK_HIP(hipMalloc(&d_h_time, sizeof(float2) * filt_elems));
  CHECK_HIP(hipMalloc(&d_h_freq, sizeof(float2) * filt_elems));
  if (real_input) {
    CHECK_HIP(hipMalloc(&d_x_real, sizeof(__half) * sig_elems));
    CHECK_HIP(hipMemcpy(d_x_real, h_x_real.data(), sizeof(__half) * sig_elems, hipMemcpyHostToDevice));
  }
  if (real_input && real_filter) CHECK_HIP(hipMalloc(&d_out_real, sizeof(__half) * sig_elems));
  CHECK_HIP(hipMemcpy(d_x, h_x.data(), sizeof(cpx_h) * sig_elems, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(d_h_time, h_h_time.data(), sizeof(float2) * filt_elems, hipMemcpyHostToDevice));
  CHECK_HIP(hipMemcpy(d_h_half, h_h_half.data(), sizeof(cpx_h) * filt_elems, hipMemcpyHostToDevice));

  hipfftHandle plan_h;
  CHECK_HIPFFT(hipfftPlan1d(&plan_h, kN, HIPFFT_C2C, D));
  thunderfft::ThunderFFTInitialize<__half>(kN);

  auto filter_fft_once = [&]() {
    if (filter_fft_mode == 0) {
      CHECK_HIPFFT(hipfftExecC2C(plan_h, reinterpret_cast<hipfftComplex*>(d_h_time),
                                 reinterpret_cast<hipfftComplex*>(d_h_freq), HIPFFT_FORWARD));
      f32_to_h2<<<dim3((unsigned)((filt_elems + 255) / 256)), dim3(256)>>>(d_h_freq, d_H, (int)filt_elems);
      CHECK_HIP(hipGetLastError());
    } else {
      check_tfft(tfft_exec_c2c(TFFT_F16, kN, 1, d_h_half, d_H, D, nullptr), "filter transform");
    }
    check_tfft(tfft_conv_filter_layout(TFFT_F16, kN, d_H, d_H_perm, D, nullptr), "filter layout");
  };

  if (!include_filter_fft) {
    filter_fft_once();
    CHECK_HIP(hipDeviceSynchronize());
  }

  const int kind = real_input ? (real_filter ? TFFT_CONV_REAL_IN_OUT : TFFT_CONV_REAL_IN) : TFFT_CONV_COMPLEX;
  const void* x_arg = real_input ? static_cast<const void*>(d_x_real) : static_cast<const void*>(d_x);
  void* y_arg = (kind == TFFT_CONV_REAL_IN_OUT) ? static_cast<void*>(d_out_real) : static_cast<void*>(d_out);
  auto run_once = [&]() {
    if (include_filter_fft) filter_fft_once();
    check_tfft(tfft_conv_exec(TFFT_F16, kN, kind, x_arg, d_H_perm, y_arg, B, D, nullptr), "fused convolution");
  };

  const float ms = measure_execution_ms(run_once, unsigned(warmup), unsigned(runs));
  std::cout << std::fixed << std::setprecision(6) << "ThunderFFT full FFT-Conv (N=4096, include_filter_fft="
            << include_filter_fft << ", filter_fft_mode=" << filter_fft_mode << ") time: " << ms << " ms\n";
  {
    // not in the reference's output: compulsory HBM bytes of the signal pass and the rate they moved at
    const double in_b = real_input ? 2.0 : 4.0, out_b = (kind == TFFT_CONV_REAL_IN_OUT) ? 2.0 : 4.0;
    const double bytes = double(sig_elems) * (in_b + out_b) + double(filt_elems) * 4.0;
    const double flops = 2.0 * 5.0 * kN * 12.0 * batch_count;
    std::cout << std::setprecision(3) << "  signals=" << batch_count << " bytes=" << bytes / 1e6 << " MB  "
              << bytes / (ms * 1e-3) / 1e9 << " GB/s (" << 100.0 * bytes / (ms * 1e-3) / 8e12 << "% of 8 TB/s)  "
              << flops / (ms * 1e-3) / 1e12 << " TFLOP/s (2 x 5NlogN per signal)\n";
  }

  if (validate) {
    std::vector<cpx_h> h_out(sig_elems);
    if (kind == TFFT_CONV_REAL_IN_OUT) {
      std::vector<__half> h_out_real(sig_elems);
      CHECK_HIP(hipMemcpy(h_out_real.data(), d_out_real, sizeof(__half) * sig_elems, hipMemcpyDeviceToHost));
      for (size_t i = 0; i < sig_elems; ++i) h_out[i] = __halves2half2(h_out_real[i], __float2half(0.0f));
    } else {
      CHECK_HIP(hipMemcpy(h_out.data(), d_out, sizeof(cpx_h) * sig_elems, hipMemcpyDeviceToHost));
    }
    double max_raw = 0.0, max_scaled = 0.0, max_corr = 0.0;
    double num_raw = 0.0, num_scaled = 0.0, den = 0.0, num_corr = 0.0, den_corr = 0.0;
    for (int bd = 0; bd < batch_count; ++bd) {
      const int d = bd % D;
      for (int n = 0; n < kN; ++n) {
        float ref_re = 0.f, ref_im = 0.f, refc_re = 0.f, refc_im = 0.f;
        for (int kk = 0; kk < K; ++kk) {
          const float2 hv = h_h_time[size_t(d) * kN + kk];
          const int xi = (n - kk >= 0) ? n - kk : n - kk + kN;
          const float2 xv = __half22float2(h_x[size_t(bd) * kN + xi]);
          ref_re += xv.x * hv.x - xv.y * hv.y;
          ref_im += xv.x * hv.y + xv.y * hv.x;
          const int xic = (n + kk < kN) ? n + kk : n + kk - kN;
          const float2 xc = __half22float2(h_x[size_t(bd) * kN + xic]);
          refc_re += xc.x * hv.x - xc.y * hv.y;
          refc_im += xc.x * hv.y + xc.y * hv.x;
        }
        const float2 ov = __half22float2(h_out[size_t(bd) * kN + n]);
        auto acc = [](float dx, float dy, double& mx, double& num) {
          const double a2 = double(dx) * dx + double(dy) * dy;
          mx = std::max(mx, std::sqrt(a2));
          num += a2;
        };
        acc(ov.x - ref_re, ov.y - ref_im, max_raw, num_raw);
        acc(ov.x / float(kN) - ref_re, ov.y / float(kN) - ref_im, max_scaled, num_scaled);
        acc(ov.x / float(kN) - refc_re, ov.y / float(kN) - refc_im, max_corr, num_corr);
        den += double(ref_re) * ref_re + double(ref_im) * ref_im;
        den_corr += double(refc_re) * refc_re + double(refc_im) * refc_im;
      }
    }
    auto rel = [](double num, double d) { return d > 0.0 ? std::sqrt(num / d) : std::sqrt(num); };
    std::cout << std::setprecision(6) << "Validation vs direct circular conv (raw): L2_rel=" << rel(num_raw, den)
              << " max_abs=" << max_raw << "\n"
              << "Validation vs direct circular conv (raw/N): L2_rel=" << rel(num_scaled, den)
              << " max_abs=" << max_scaled << "\n"
              << "Validation vs direct circular corr (raw/N): L2_rel=" << rel(num_corr, den_corr)
              << " max_abs=" << max_corr << "\n";
  }

  thunderfft::ThunderFFTFinalize<__half>();
  CHECK_HIPFFT(hipfftDestroy(plan_h));
  CHECK_HIP(hipFree(d_x));
  CHECK_HIP(hipFree(d_out));
  CHECK_HIP(hipFree(d_H));
  CHECK_HIP(hipFree(d_H_perm));
  CHECK_HIP(hipFree(d_h_half));
  CHECK_HIP(hipFree(d_h_time));
  CHECK_HIP(hipFree(d_h_freq));
  if (d_x_real) CHECK_HIP(hipFree(d_x_real));
  if (d_out_real) CHECK_HIP(hipFree(d_out_real));
  return 0;
}
