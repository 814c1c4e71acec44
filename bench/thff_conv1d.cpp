// Tiled 1D convolution benchmark (overlap-save, transform length fixed to 64, fp16 storage): the
// MI355X build of the reference's thff_conv1d (thunderfft/sample/conv1d/main.cu:208-336).
//
// Same positional arguments (:216-221): input_len filter_len device validate warmup runs; same
// report block (:241-250), "Kernel pipeline time" line (:303-304) and validation lines (:317-325).
// Inputs come from srand(42) / rand() in the reference's order (:131-136, 258-260). The filter
// spectrum is made in fp32 by hipFFT where the reference uses cuFFT (:280-287). The device work goes
// through the C ABI (tfft_conv_tiles_exec).
#include <hip/hip_fp16.h>
#include <hip/hip_runtime.h>
#include <hipfft/hipfft.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <iomanip>
#include <iostream>
#include <vector>

#include <thunderfft/thunderfft.cuh>

namespace {

constexpr int kFFTSize = 64;

__global__ void f32_to_half2_kernel(const float2* __restrict__ in, __half2* __restrict__ out, int n) {
  const int idx = int(blockIdx.x) * int(blockDim.x) + int(threadIdx.x);
  if (idx < n) out[idx] = __floats2half2_rn(in[idx].x, in[idx].y);
}

void make_random_complex(std::vector<float2>& x) {
  for (auto& v : x) {
    v.x = rand() / float(RAND_MAX);
    v.y = rand() / float(RAND_MAX);
  }
}

double l2_rel(const std::vector<float2>& a, const std::vector<float2>& b) {
  long double num = 0.0L, den = 0.0L;
  for (size_t i = 0; i < a.size(); ++i) {
    const long double dx = (long double)a[i].x - b[i].x, dy = (long double)a[i].y - b[i].y;
    num += dx * dx + dy * dy;
    den += (long double)b[i].x * b[i].x + (long double)b[i].y * b[i].y;
  }
  return den > 0.0L ? (double)std::sqrt(num / den) : (double)std::sqrt(num);
}

double linf_rel(const std::vector<float2>& a, const std::vector<float2>& b) {
  long double max_diff = 0.0L, max_ref = 0.0L;
  for (size_t i = 0; i < a.size(); ++i) {
    const long double dx = (long double)a[i].x - b[i].x, dy = (long double)a[i].y - b[i].y;
    max_diff = std::max(max_diff, std::sqrt(dx * dx + dy * dy));
    max_ref = std::max(max_ref, std::sqrt((long double)b[i].x * b[i].x + (long double)b[i].y * b[i].y));
  }
  return max_ref > 0.0L ? (double)(max_diff / max_ref) : (double)max_diff;
}

}  // namespace

int main(int argc, char** argv) {
  int input_len = 1 << 16, filter_len = 9, device = 0, validate = 1, warmup = 5, runs = 50;
  if (argc >= 2) input_len = std::atoi(argv[1]);
  if (argc >= 3) filter_len = std::atoi(argv[2]);
  if (argc >= 4) device = std::atoi(argv[3]);
  if (argc >= 5) validate = std::atoi(argv[4]);
  if (argc >= 6) warmup = std::atoi(argv[5]);
  if (argc >= 7) runs = std::atoi(argv[6]);
  if (filter_len < 1 || filter_len > kFFTSize) {
    std::cerr << "filter_len must be in [1, " << kFFTSize << "]\n";
    return EXIT_FAILURE;
  }
  if (input_len < filter_len) {
    std::cerr << "input_len must be >= filter_len\n";
    return EXIT_FAILURE;
  }
  CHECK_HIP(hipSetDevice(device));
  srand(42);

  tfft_plan_info info;
  tfft_get_plan_info(TFFT_F16, kFFTSize, &info);
  const int tiles_per_block = 16;  // tiles one 64-thread workgroup of the tiled kernel holds
  const int valid_tile = kFFTSize - filter_len + 1;
  const int output_len = input_len - filter_len + 1;
  const int num_tiles = (output_len + valid_tile - 1) / valid_tile;
  const int launch_blocks = (num_tiles + tiles_per_block - 1) / tiles_per_block;
  std::cout << "ThunderFFT tiled 1D conv (FFT size fixed to 64)\n"
            << "  input_len  : " << input_len << "\n"
            << "  filter_len : " << filter_len << "\n"
            << "  output_len : " << output_len << "\n"
            << "  valid_tile : " << valid_tile << "\n"
            << "  num_tiles  : " << num_tiles << "\n"
            << "  total_blocks: " << launch_blocks << "\n"
            << "  launch_blocks: " << launch_blocks << "\n"
            << "  tiles_per_block(static): " << tiles_per_block << "\n"
            << "  warmup/runs: " << warmup << "/" << runs << "\n";

  std::vector<float2> h_input_f32(input_len), h_filter(filter_len);
  make_random_complex(h_input_f32);
  std::vector<__half2> h_input(input_len);
  for (int i = 0; i < input_len; ++i) h_input[i] = __floats2half2_rn(h_input_f32[i].x, h_input_f32[i].y);
  make_random_complex(h_filter);

  __half2 *d_input = nullptr, *d_output = nullptr, *d_filter_fft = nullptr;
  float2 *d_filter_time = nullptr, *d_filter_fft_f32 = nullptr;
  CHECK_HIP(hipMalloc(&d_input, sizeof(__half2) * input_len));
  CHECK_HIP(hipMalloc(&d_output, sizeof(__half2) * output_len));
  CHECK_HIP(hipMalloc(&d_filter_fft, sizeof(__half2) * kFFTSize));
  CHECK_HIP(hipMalloc(&d_filter_time, sizeof(float2) * kFFTSize));
  CHECK_HIP(hipMalloc(&d_filter_fft_f32, sizeof(float2) * kFFTSize));
  CHECK_HIP(hipMemcpy(d_input, h_input.data(), sizeof(__half2) * input_len, hipMemcpyHostToDevice));

  // taps at (N - i) % N, scaled by 1/N: the circular convolution then is the valid cross-correlation
  std::vector<float2> padded(kFFTSize, make_float2(0.f, 0.f));
  for (int i = 0; i < filter_len; ++i) {
    const int idx = (kFFTSize - i) % kFFTSize;
    padded[idx].x = h_filter[i].x / float(kFFTSize);
    padded[idx].y = h_filter[i].y / float(kFFTSize);
  }
  CHECK_HIP(hipMemcpy(d_filter_time, padded.data(), sizeof(float2) * kFFTSize, hipMemcpyHostToDevice));
  hipfftHandle plan;
  if (hipfftPlan1d(&plan, kFFTSize, HIPFFT_C2C, 1) != HIPFFT_SUCCESS ||
      hipfftExecC2C(plan, reinterpret_cast<hipfftComplex*>(d_filter_time),
                    reinterpret_cast<hipfftComplex*>(d_filter_fft_f32), HIPFFT_FORWARD) != HIPFFT_SUCCESS) {
    std::fprintf(stderr, "hipFFT failed on the filter transform\n");
    return EXIT_FAILURE;
  }
  f32_to_half2_kernel<<<1, 64>>>(d_filter_fft_f32, d_filter_fft, kFFTSize);
  CHECK_HIP(hipGetLastError());
  CHECK_HIP(hipDeviceSynchronize());

  thunderfft::ThunderFFTInitialize<__half>(kFFTSize);
  auto run_once = [&]() {
    thunderfft::check(tfft_conv_tiles_exec(TFFT_F16, kFFTSize, d_input, d_filter_fft, d_output, input_len, output_len,
                                           valid_tile, nullptr),
                      "tiled convolution");
  };
  hipEvent_t e0, e1;
  CHECK_HIP(hipEventCreate(&e0));
  CHECK_HIP(hipEventCreate(&e1));
  CHECK_HIP(hipDeviceSynchronize());
  for (int i = 0; i < warmup; ++i) run_once();
  CHECK_HIP(hipDeviceSynchronize());
  CHECK_HIP(hipEventRecord(e0));
  for (int i = 0; i < runs; ++i) run_once();
  CHECK_HIP(hipEventRecord(e1));
  CHECK_HIP(hipDeviceSynchronize());
  float ms = 0.f;
  CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
  ms /= float(std::max(runs, 1));
  std::cout << std::fixed << std::setprecision(6) << "Kernel pipeline time: " << ms << " ms\n";

  std::vector<__half2> h_output(output_len);
  CHECK_HIP(hipMemcpy(h_output.data(), d_output, sizeof(__half2) * output_len, hipMemcpyDeviceToHost));
  if (validate != 0) {
    std::vector<float2> got(output_len), ref(output_len, make_float2(0.f, 0.f));
    for (int i = 0; i < output_len; ++i) got[i] = __half22float2(h_output[i]);
    for (int i = 0; i < output_len; ++i) {
      float re = 0.f, im = 0.f;
      for (int k = 0; k < filter_len; ++k) {
        const float2 a = h_input_f32[i + k], b = h_filter[k];
        re += a.x * b.x - a.y * b.y;
        im += a.x * b.y + a.y * b.x;
      }
      ref[i] = make_float2(re, im);
    }
    std::cout << std::setprecision(4) << "Validation vs direct conv: L2_rel=" << l2_rel(got, ref)
              << " Linf_rel=" << linf_rel(got, ref) << "\n";
    for (int i = 0; i < std::min(8, output_len); ++i)
      std::cout << "  y[" << i << "] th=(" << got[i].x << ", " << got[i].y << ") ref=(" << ref[i].x << ", " << ref[i].y
                << ")\n";
  }

  thunderfft::ThunderFFTFinalize<__half>();
  hipfftDestroy(plan);
  CHECK_HIP(hipFree(d_input));
  CHECK_HIP(hipFree(d_output));
  CHECK_HIP(hipFree(d_filter_fft));
  CHECK_HIP(hipFree(d_filter_fft_f32));
  CHECK_HIP(hipFree(d_filter_time));
  return 0;
}
