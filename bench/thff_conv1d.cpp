// thff_conv1d for MI355X: overlap-save tiled 1D convolution with 64-point transforms on fp16 storage,
// through the C ABI (tfft_exec_c2c for the filter spectrum, tfft_conv_tiles_exec for the tiles).
//
// Contract kept from the reference's sample of the same name: its six positional arguments
//   input_len filter_len device validate warmup runs      (thunderfft/sample/conv1d/main.cu:216-221)
// and its report lines: the geometry block (:241-250), "Kernel pipeline time: X ms" (:303-304) and
// "Validation vs direct conv: L2_rel=... Linf_rel=..." (:317-319). The result is the valid cross-correlation
// out[i] = sum_k in[i + k] * filter[k], obtained by storing tap k at (64 - k) mod 64 scaled by 1/64 (:145-174).
// Data generation, buffers, the filter transform (the library's own fp32 transform; the reference used
// cuFFT) and the fp64 validation are this build's own.
#include <algorithm>
#include <iomanip>
#include <iostream>

#include "harness.hpp"

namespace {

using harness::cplx;
using harness::DeviceArray;
constexpr int kTileLen = 64;

struct Config {
  int input_len = 1 << 16, filter_len = 9, device = 0, validate = 1, warmup = 5, runs = 50;
};

// Spectrum of the time-reversed, 1/N-scaled taps, rounded to fp16: what every tile is multiplied by.
DeviceArray<__half2> filter_spectrum(const std::vector<cplx>& taps) {
  std::vector<float2> wrapped(kTileLen, make_float2(0.f, 0.f));
  for (size_t k = 0; k < taps.size(); ++k)
    wrapped[(kTileLen - k) % kTileLen] = make_float2((float)(taps[k].real() / kTileLen), (float)(taps[k].imag() / kTileLen));
  DeviceArray<float2> time32, freq32(kTileLen);
  time32.upload(wrapped);
  thunderfft::ThunderFFTInitialize<float>(kTileLen);
  harness::tfft_ok(tfft_exec_c2c(TFFT_F32, kTileLen, 1, time32.get(), freq32.get(), 1, nullptr), "filter transform");
  DeviceArray<__half2> freq16(kTileLen);
  harness::narrow(freq32.get(), freq16.get(), kTileLen);
  harness::hip_ok(hipDeviceSynchronize(), "hipDeviceSynchronize");
  thunderfft::ThunderFFTFinalize<float>();
  return freq16;
}

}  // namespace

int main(int argc, char** argv) {
  Config cfg;
  harness::positional(argc, argv, {&cfg.input_len, &cfg.filter_len, &cfg.device, &cfg.validate, &cfg.warmup, &cfg.runs});
  if (cfg.filter_len < 1 || cfg.filter_len > kTileLen)
    harness::die("filter_len must be in [1, " + std::to_string(kTileLen) + "]");
  if (cfg.input_len < cfg.filter_len) harness::die("input_len must be >= filter_len");
  if (cfg.runs < 1 || cfg.warmup < 0) harness::die("runs must be positive");
  harness::hip_ok(hipSetDevice(cfg.device), "hipSetDevice");

  const int valid = kTileLen - cfg.filter_len + 1;  // outputs kept per tile
  const int output_len = cfg.input_len - cfg.filter_len + 1;
  const int tiles = (output_len + valid - 1) / valid;
  const int tiles_per_block = 16;  // tiles one 64-thread workgroup of the tiled kernel holds
  const int blocks = (tiles + tiles_per_block - 1) / tiles_per_block;
  std::cout << "ThunderFFT tiled 1D conv (FFT size fixed to 64)\n"
            << "  input_len  : " << cfg.input_len << "\n"
            << "  filter_len : " << cfg.filter_len << "\n"
            << "  output_len : " << output_len << "\n"
            << "  valid_tile : " << valid << "\n"
            << "  num_tiles  : " << tiles << "\n"
            << "  total_blocks: " << blocks << "\n"
            << "  launch_blocks: " << blocks << "\n"
            << "  tiles_per_block(static): " << tiles_per_block << "\n"
            << "  warmup/runs: " << cfg.warmup << "/" << cfg.runs << "\n";

  // signal and taps uniform in [0, 1) + i [0, 1); the signal is kept as the device sees it (fp16)
  std::vector<__half2> signal(cfg.input_len);
  for (int i = 0; i < cfg.input_len; ++i)
    signal[i] = __floats2half2_rn(harness::uniform01(2 * (uint64_t)i, 11), harness::uniform01(2 * (uint64_t)i + 1, 11));
  std::vector<cplx> taps(cfg.filter_len);
  for (int k = 0; k < cfg.filter_len; ++k)
    taps[k] = cplx(harness::uniform01(2 * (uint64_t)k, 12), harness::uniform01(2 * (uint64_t)k + 1, 12));

  DeviceArray<__half2> d_signal, d_result(output_len);
  d_signal.upload(signal);
  const DeviceArray<__half2> d_spectrum = filter_spectrum(taps);

  thunderfft::ThunderFFTInitialize<__half>(kTileLen);
  const float ms = harness::mean_ms(
      [&] {
        harness::tfft_ok(tfft_conv_tiles_exec(TFFT_F16, kTileLen, d_signal.get(), d_spectrum.get(), d_result.get(),
                                              cfg.input_len, output_len, valid, nullptr),
                         "tiled convolution");
      },
      cfg.warmup, cfg.runs);
  std::cout << std::fixed << std::setprecision(6) << "Kernel pipeline time: " << ms << " ms\n";

  if (cfg.validate) {
    const std::vector<__half2> result = d_result.download();
    harness::ErrorSums err;
    std::vector<cplx> expect_head;
    for (int i = 0; i < output_len; ++i) {
      cplx want = 0;
      for (int k = 0; k < cfg.filter_len; ++k) want += harness::to_cplx(signal[i + k]) * taps[k];
      err.add(harness::to_cplx(result[i]), want);
      if (i < 8) expect_head.push_back(want);
    }
    std::cout << std::setprecision(4) << "Validation vs direct conv: L2_rel=" << err.l2_rel()
              << " Linf_rel=" << err.linf_rel() << "\n";
    for (size_t i = 0; i < expect_head.size(); ++i) {
      const cplx got = harness::to_cplx(result[i]);
      std::cout << "  y[" << i << "] th=(" << got.real() << ", " << got.imag() << ") ref=(" << expect_head[i].real() << ", "
                << expect_head[i].imag() << ")\n";
    }
  }
  thunderfft::ThunderFFTFinalize<__half>();
  return 0;
}
