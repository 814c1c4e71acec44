// thff_conv1d_fullfft4096 for MI355X: times the fused circular convolution of length 4096 on fp16 storage,
//   y[b][d] = IDFT(DFT(x[b][d]) * DFT(h[d])),
// through the C ABI (tfft_exec_c2c, tfft_conv_filter_layout, tfft_conv_exec).
//
// Contract kept from the reference's sample of the same name, because its Figure-5 driver starts the binary
// and parses its output (Figure5_1dFFT/benchmark_figure5_1dfft.py:27-53):
//   * the eleven positional arguments, in this order
//     (thunderfft/sample/conv1d/conv1d_fullfft_l4096_thunder.cu:305-327):
//       K device B D warmup runs include_filter_fft validate filter_fft_mode real_input real_filter
//   * the "ThunderFFT full FFT-Conv (...) time: X ms" line (:452-455) and the three "Validation vs direct
//     circular ..." lines (:516-522).
// Everything else is this build's own: inputs come from a counter-based generator, the expected values are
// summed in fp64 from the definition of the circular convolution, and large runs validate a spread sample
// of the signals instead of all of them. filter_fft_mode 0 transforms the taps in fp32 with hipFFT (the
// reference: cuFFT), mode 1 with the library's own fp16-storage transform.
// Environment: TFFT_MODE=3 selects the fp32-VALU kernels, TFFT_CONV_RESIDENT_PER_CU=k a resident grid.
#include <hipfft/hipfft.h>

#include <algorithm>
#include <iomanip>
#include <iostream>

#include "harness.hpp"

namespace {

using harness::cplx;
using harness::DeviceArray;
constexpr int kLen = 4096;

struct Config {
  int taps = 9, device = 0, B = 1, D = 1, warmup = 5, runs = 30;
  int filter_in_loop = 1, validate = 1, filter_fft_mode = 1, real_input = 0, real_filter = 0;
  int conv_kind() const {
    if (!real_input) return TFFT_CONV_COMPLEX;
    return real_filter ? TFFT_CONV_REAL_IN_OUT : TFFT_CONV_REAL_IN;
  }
  size_t signals() const { return (size_t)B * D; }
};

Config parse(int argc, char** argv) {
  Config c;
  harness::positional(argc, argv, {&c.taps, &c.device, &c.B, &c.D, &c.warmup, &c.runs, &c.filter_in_loop, &c.validate,
                                   &c.filter_fft_mode, &c.real_input, &c.real_filter});
  if (c.taps < 1 || c.taps > kLen) harness::die("K must be in [1, " + std::to_string(kLen) + "]");
  if (c.B < 1 || c.D < 1 || c.runs < 1 || c.warmup < 0) harness::die("B, D and runs must be positive");
  return c;
}

// The synthetic problem as the device sees it: fp16 signals, taps already divided by N (so that the
// unnormalised inverse returns the convolution times N and the spectra product stays inside fp16 range).
struct Problem {
  std::vector<__half2> x;      // [B*D][4096], imaginary parts zero for real input
  std::vector<float2> h_f32;   // [D][4096], zero beyond the taps
  std::vector<__half2> h_f16;  // the same rounded to fp16
  explicit Problem(const Config& c) : x(c.signals() * kLen), h_f32((size_t)c.D * kLen, make_float2(0.f, 0.f)),
                                      h_f16((size_t)c.D * kLen, __floats2half2_rn(0.f, 0.f)) {
    for (size_t i = 0; i < x.size(); ++i)
      x[i] = __floats2half2_rn(harness::uniform01(2 * i, 1), c.real_input ? 0.f : harness::uniform01(2 * i + 1, 1));
    for (int d = 0; d < c.D; ++d)
      for (int k = 0; k < c.taps; ++k) {
        const uint64_t id = 2 * ((uint64_t)d * kLen + k);
        const float2 v = make_float2(harness::uniform01(id, 2) / kLen,
                                     c.real_filter ? 0.f : harness::uniform01(id + 1, 2) / kLen);
        h_f32[(size_t)d * kLen + k] = v;
        h_f16[(size_t)d * kLen + k] = __floats2half2_rn(v.x, v.y);
      }
  }
};

// Filter spectra in the order the fused kernel reads, remade on every call of prepare().
class FilterSpectra {
 public:
  FilterSpectra(const Config& c, const Problem& p, const tfft_exec_options& opt)
      : D_(c.D), mode_(c.filter_fft_mode), opt_(opt), spectra_((size_t)c.D * kLen), permuted_((size_t)c.D * kLen) {
    if (mode_ == 0) {
      taps32_.upload(p.h_f32);
      spectra32_.resize(p.h_f32.size());
      if (hipfftPlan1d(&plan_, kLen, HIPFFT_C2C, D_) != HIPFFT_SUCCESS) harness::die("hipfftPlan1d failed");
    } else {
      taps16_.upload(p.h_f16);
    }
  }
  ~FilterSpectra() {
    if (plan_) hipfftDestroy(plan_);
  }
  void prepare() {
    if (mode_ == 0) {
      if (hipfftExecC2C(plan_, reinterpret_cast<hipfftComplex*>(taps32_.get()),
                        reinterpret_cast<hipfftComplex*>(spectra32_.get()), HIPFFT_FORWARD) != HIPFFT_SUCCESS)
        harness::die("hipfftExecC2C failed");
      harness::narrow(spectra32_.get(), spectra_.get(), spectra_.size());
    } else if (opt_.plan == 0) {
      // the library's own transform, default convolution kernel: transform and re-ordering in one kernel
      harness::tfft_ok(tfft_conv_filter_spectra(TFFT_F16, kLen, taps16_.get(), permuted_.get(), D_, nullptr),
                       "filter transform + layout");
      return;
    } else {
      harness::tfft_ok(tfft_exec_c2c(TFFT_F16, kLen, 1, taps16_.get(), spectra_.get(), D_, nullptr), "filter transform");
    }
    harness::tfft_ok(tfft_conv_filter_layout_opt(TFFT_F16, kLen, spectra_.get(), permuted_.get(), D_, nullptr, &opt_),
                     "filter layout");
  }
  const __half2* permuted() const { return permuted_.get(); }

 private:
  int D_, mode_;
  tfft_exec_options opt_;
  hipfftHandle plan_ = 0;
  DeviceArray<float2> taps32_, spectra32_;
  DeviceArray<__half2> taps16_, spectra_, permuted_;
};

// Expected results by the definitions, in fp64, for a sample of the signals:
//   convolution  c[n] = sum_k h[k] x[(n - k) mod N],  correlation  r[n] = sum_k h[k] x[(n + k) mod N].
// The kernel returns N * c (unnormalised inverse), so "raw" is expected to be off by that factor, "raw/N"
// to match c, and the correlation line shows that the result is not r.
struct Validation {
  harness::ErrorSums raw, scaled, corr;
  size_t checked = 0;
};

Validation validate(const Config& c, const Problem& p, const std::vector<cplx>& y_of_signal_sample,
                    const std::vector<size_t>& sample) {
  Validation v;
  v.checked = sample.size();
  const bool taps16 = c.filter_fft_mode != 0;  // the taps as the device read them
  for (size_t s = 0; s < sample.size(); ++s) {
    const size_t sig = sample[s];
    const size_t d = sig % (size_t)c.D;  // signal (b, d) uses filter d
    std::vector<cplx> x(kLen), h(c.taps);
    for (int n = 0; n < kLen; ++n) x[n] = harness::to_cplx(p.x[sig * kLen + n]);
    for (int k = 0; k < c.taps; ++k)
      h[k] = taps16 ? harness::to_cplx(p.h_f16[d * kLen + k]) : harness::to_cplx(p.h_f32[d * kLen + k]);
    for (int n = 0; n < kLen; ++n) {
      cplx conv = 0, corr = 0;
      for (int k = 0; k < c.taps; ++k) {
        conv += h[k] * x[(n - k) & (kLen - 1)];
        corr += h[k] * x[(n + k) & (kLen - 1)];
      }
      const cplx got = y_of_signal_sample[s * kLen + n];
      v.raw.add(got, conv);
      v.scaled.add(got / (double)kLen, conv);
      v.corr.add(got / (double)kLen, corr);
    }
  }
  return v;
}

}  // namespace

int main(int argc, char** argv) {
  const Config cfg = parse(argc, argv);
  harness::hip_ok(hipSetDevice(cfg.device), "hipSetDevice");
  tfft_exec_options opt;
  tfft_exec_options_init(&opt);
  if (const char* e = std::getenv("TFFT_MODE")) opt.plan = std::atoi(e);
  if (const char* e = std::getenv("TFFT_CONV_RESIDENT_PER_CU")) opt.grid_blocks_per_cu = std::atoi(e);

  const Problem problem(cfg);
  const int kind = cfg.conv_kind();
  const size_t points = cfg.signals() * kLen;

  // device images of the signals: interleaved complex, or the real parts alone for the real-input kinds
  DeviceArray<__half2> x_cplx, y_cplx;
  DeviceArray<__half> x_real, y_real;
  if (cfg.real_input) {
    std::vector<__half> re(points);
    for (size_t i = 0; i < points; ++i) re[i] = __low2half(problem.x[i]);
    x_real.upload(re);
  } else {
    x_cplx.upload(problem.x);
  }
  if (kind == TFFT_CONV_REAL_IN_OUT) y_real.resize(points); else y_cplx.resize(points);
  const void* x_dev = cfg.real_input ? static_cast<const void*>(x_real.get()) : static_cast<const void*>(x_cplx.get());
  void* y_dev = kind == TFFT_CONV_REAL_IN_OUT ? static_cast<void*>(y_real.get()) : static_cast<void*>(y_cplx.get());

  thunderfft::ThunderFFTInitialize<__half>(kLen);
  FilterSpectra filters(cfg, problem, opt);
  if (!cfg.filter_in_loop) filters.prepare();

  const float ms = harness::mean_ms(
      [&] {
        if (cfg.filter_in_loop) filters.prepare();
        harness::tfft_ok(tfft_conv_exec_opt(TFFT_F16, kLen, kind, x_dev, filters.permuted(), y_dev, cfg.B, cfg.D, nullptr, &opt),
                         "fused convolution");
      },
      cfg.warmup, cfg.runs);

  std::cout << std::fixed << std::setprecision(6) << "ThunderFFT full FFT-Conv (N=4096, include_filter_fft="
            << cfg.filter_in_loop << ", filter_fft_mode=" << cfg.filter_fft_mode << ") time: " << ms << " ms\n";
  {
    // beyond the contract: compulsory HBM bytes of the signal pass (signals in + out, filter spectra once)
    const double in_bytes = cfg.real_input ? 2.0 : 4.0, out_bytes = kind == TFFT_CONV_REAL_IN_OUT ? 2.0 : 4.0;
    const double bytes = (double)points * (in_bytes + out_bytes) + 4.0 * cfg.D * kLen;
    const double flops = 2.0 * 5.0 * kLen * 12.0 * (double)cfg.signals();
    const double sec = ms * 1e-3;
    std::cout << std::setprecision(3) << "  signals=" << cfg.signals() << " bytes=" << bytes / 1e6 << " MB  "
              << bytes / sec / 1e9 << " GB/s (" << 100.0 * bytes / sec / 8e12 << "% of 8 TB/s)  " << flops / sec / 1e12
              << " TFLOP/s (2 x 5NlogN per signal)\n";
  }

  if (cfg.validate) {
    // every signal when there are few, else 64 of them spread over the batch (the direct sums cost N*K each)
    std::vector<size_t> sample;
    const size_t total = cfg.signals(), want = std::min<size_t>(total, 64);
    for (size_t i = 0; i < want; ++i) sample.push_back(i * total / want);
    std::vector<cplx> y(sample.size() * kLen);
    if (kind == TFFT_CONV_REAL_IN_OUT) {
      const std::vector<__half> host = y_real.download();
      for (size_t s = 0; s < sample.size(); ++s)
        for (int n = 0; n < kLen; ++n) y[s * kLen + n] = cplx(__half2float(host[sample[s] * kLen + n]), 0.0);
    } else {
      const std::vector<__half2> host = y_cplx.download();
      for (size_t s = 0; s < sample.size(); ++s)
        for (int n = 0; n < kLen; ++n) y[s * kLen + n] = harness::to_cplx(host[sample[s] * kLen + n]);
    }
    const Validation v = validate(cfg, problem, y, sample);
    std::cout << std::setprecision(6) << "Validation vs direct circular conv (raw): L2_rel=" << v.raw.l2_rel()
              << " max_abs=" << v.raw.max_diff << "\n"
              << "Validation vs direct circular conv (raw/N): L2_rel=" << v.scaled.l2_rel()
              << " max_abs=" << v.scaled.max_diff << "\n"
              << "Validation vs direct circular corr (raw/N): L2_rel=" << v.corr.l2_rel()
              << " max_abs=" << v.corr.max_diff << "\n"
              << "  (" << v.checked << " of " << total << " signals checked, fp64 sums)\n";
  }
  thunderfft::ThunderFFTFinalize<__half>();
  return 0;
}
