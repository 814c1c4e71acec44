/* CPU oracle for the batched 1D complex FFT path. TEST INFRASTRUCTURE ONLY: nothing under
 * fft_amd/ links, loads or calls this file; tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py are its only users.
 *
 * What it restates. The reference's implementation of this path is CUDA tensor-core code
 * (inline PTX mma.sync, thunderfft/include/thunderfft/detail/kernel/unit_kernel_fp32.cuh:113-136)
 * and cannot be compiled or run without nvcc and an NVIDIA GPU, and its only comparator is
 * cuFFT (thunderfft/sample/bench/baseline/baseline_cufft.cu:41-52). The oracle therefore
 * restates the transform's DEFINITION as the reference fixes it, in fp64:
 *   - forward: X[k] = sum_n x[n] * exp(-2 pi i n k / N), unnormalised, natural order in and
 *     out -- sign from rotate() (detail/util_arith.h:85-91), forward "-i" butterflies
 *     (detail/kernel/shared_kernel_n256_b16.cuh:69-85), CUFFT_FORWARD comparator
 *     (baseline_cufft.cu:46-47);
 *   - inverse: the "+" sign, also unnormalised (shared_kernel_n256_b16.cuh:135-151; the conv
 *     sample divides its filter by N itself, sample/conv1d/conv1d_fullfft_l4096_thunder.cu:354-356);
 *   - layout: `batch` contiguous transforms of N interleaved (re, im) points
 *     (detail/kernel/global_kernel.cuh:37,49).
 *
 * Pinning. The reference ships no golden vectors, fixtures or stored outputs for this path
 * (no tests directory, no committed results), so parity is pinned by this definition plus the
 * closed-form answers of the reference's own test inputs (sample/bench/main.cu:35-41,
 * sample/fft1d/main.cu:16-30, conv1d_fullfft_l4096_thunder.cu:472-490), checked in
 * tests/test_oracle.py, and cross-checked against numpy.fft on complex128. In the strict sense
 * of "pinned by reference-held vectors" parity is UNPINNED.
 *
 * Two independent evaluations are provided so they can check each other:
 *   oracle_dft_direct : O(N^2) sum with the angle index reduced mod N in integers
 *   oracle_fft_radix2 : iterative decimation-in-time radix-2 FFT
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* in/out: interleaved (re, im) doubles, batch * n points. sign = -1 forward, +1 inverse. */
int oracle_dft_direct(const double* in, double* out, int n, long long batch, int sign) {
  if (n <= 0 || batch < 0 || (sign != 1 && sign != -1)) return 1;
  double* cs = (double*)malloc(sizeof(double) * 2 * (size_t)n);
  if (!cs) return 2;
  for (int j = 0; j < n; ++j) {
    const double a = 2.0 * M_PI * (double)j / (double)n;
    cs[2 * j] = cos(a);
    cs[2 * j + 1] = sign * sin(a);
  }
  for (long long b = 0; b < batch; ++b) {
    const double* x = in + 2 * (size_t)b * n;
    double* y = out + 2 * (size_t)b * n;
    for (int k = 0; k < n; ++k) {
      double re = 0.0, im = 0.0;
      long long idx = 0; /* (n_i * k) mod n, kept exact */
      for (int i = 0; i < n; ++i) {
        const double c = cs[2 * idx], s = cs[2 * idx + 1];
        re += x[2 * i] * c - x[2 * i + 1] * s;
        im += x[2 * i] * s + x[2 * i + 1] * c;
        idx += k;
        if (idx >= n) idx -= n;
      }
      y[2 * k] = re;
      y[2 * k + 1] = im;
    }
  }
  free(cs);
  return 0;
}

static int ilog2_exact(int n) {
  int l = 0;
  while ((1 << l) < n) ++l;
  return ((1 << l) == n) ? l : -1;
}

/* n must be a power of two. Out of place; in and out may alias. */
int oracle_fft_radix2(const double* in, double* out, int n, long long batch, int sign) {
  const int lg = ilog2_exact(n);
  if (lg < 0 || batch < 0 || (sign != 1 && sign != -1)) return 1;
  double* cs = (double*)malloc(sizeof(double) * (size_t)n); /* n/2 twiddles */
  int* rev = (int*)malloc(sizeof(int) * (size_t)n);
  double* tmp = (double*)malloc(sizeof(double) * 2 * (size_t)n);
  if (!cs || !rev || !tmp) {
    free(cs); free(rev); free(tmp);
    return 2;
  }
  for (int j = 0; j < n / 2; ++j) {
    const double a = 2.0 * M_PI * (double)j / (double)n;
    cs[2 * j] = cos(a);
    cs[2 * j + 1] = sign * sin(a);
  }
  for (int i = 0; i < n; ++i) {
    int r = 0;
    for (int b = 0; b < lg; ++b) r |= ((i >> b) & 1) << (lg - 1 - b);
    rev[i] = r;
  }
  for (long long b = 0; b < batch; ++b) {
    const double* x = in + 2 * (size_t)b * n;
    double* y = out + 2 * (size_t)b * n;
    for (int i = 0; i < n; ++i) {
      tmp[2 * i] = x[2 * rev[i]];
      tmp[2 * i + 1] = x[2 * rev[i] + 1];
    }
    for (int len = 2; len <= n; len <<= 1) {
      const int half = len / 2, step = n / len;
      for (int s0 = 0; s0 < n; s0 += len) {
        for (int j = 0; j < half; ++j) {
          const double c = cs[2 * (j * step)], s = cs[2 * (j * step) + 1];
          double* a = tmp + 2 * (s0 + j);
          double* bb = tmp + 2 * (s0 + j + half);
          const double tr = bb[0] * c - bb[1] * s;
          const double ti = bb[0] * s + bb[1] * c;
          bb[0] = a[0] - tr;
          bb[1] = a[1] - ti;
          a[0] += tr;
          a[1] += ti;
        }
      }
    }
    memcpy(y, tmp, sizeof(double) * 2 * (size_t)n);
  }
  free(cs); free(rev); free(tmp);
  return 0;
}

/* Convenience entry for fp32 inputs (the dtype the device path stores): widen, transform in
 * fp64 with the radix-2 evaluation, return fp64. */
int oracle_fft_f32(const float* in, double* out, int n, long long batch, int sign) {
  const size_t total = 2 * (size_t)n * (size_t)batch;
  double* wide = (double*)malloc(sizeof(double) * (total ? total : 1));
  if (!wide) return 2;
  for (size_t i = 0; i < total; ++i) wide[i] = (double)in[i];
  const int rc = oracle_fft_radix2(wide, out, n, batch, sign);
  free(wide);
  return rc;
}

/* Circular convolution by its definition, the reference's own validation loop
 * (sample/conv1d/conv1d_fullfft_l4096_thunder.cu:472-483) in fp64:
 *   y[s][n] = sum_{k < K} x[s][(n - k) mod N] * h[s % D][k]
 * x: `signals` x n interleaved complex; h: D x K interleaved complex; y like x. The fused device
 * kernel returns N * y when h was not pre-scaled by 1/N (unnormalised inverse, :354-356). */
int oracle_circular_conv(const double* x, const double* h, double* y, int n, int K, long long signals, int D) {
  if (n <= 0 || K <= 0 || K > n || signals < 0 || D <= 0) return 1;
  for (long long s = 0; s < signals; ++s) {
    const double* xs = x + 2 * (size_t)s * (size_t)n;
    const double* hs = h + 2 * (size_t)(s % D) * (size_t)K;
    double* ys = y + 2 * (size_t)s * (size_t)n;
    for (int i = 0; i < n; ++i) {
      double re = 0.0, im = 0.0;
      for (int k = 0; k < K; ++k) {
        int idx = i - k;
        if (idx < 0) idx += n;
        const double xr = xs[2 * idx], xi = xs[2 * idx + 1];
        re += xr * hs[2 * k] - xi * hs[2 * k + 1];
        im += xr * hs[2 * k + 1] + xi * hs[2 * k];
      }
      ys[2 * i] = re;
      ys[2 * i + 1] = im;
    }
  }
  return 0;
}
