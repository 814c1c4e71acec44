// Result table of the bench harness: same record and sinks as the reference's stat module
// (thunderfft/sample/bench/util/stat.h:7-30, stat.cu:33-77), extended with the throughput and
// roofline columns the reference never computes (it reports milliseconds only).
#pragma once
#include <cstdio>
#include <string>
#include <vector>

namespace stat {

struct RunStat {
  std::string type;    // e.g. "thunderfft(fp32)" or "hipfft(fp32)"
  unsigned N;
  std::string radix;   // e.g. "16x16x16"
  long long B;
  double max_abs_err;  // vs the hipFFT comparator, like the reference's vs-cuFFT figure
  double compute_ms;   // (t(2R) - t(R)) / R in-kernel repeats: one transform pass without HBM; NaN if not run
  double e2e_ms;       // mean of the timed single launches
  double gflops;       // 5 N log2 N per transform
  double gbps;         // one read + one write of every point
  double hbm_frac;     // gbps / 8000
};

inline std::vector<RunStat>& rows() {
  static std::vector<RunStat> r;
  return r;
}
inline std::string& title() {
  static std::string t = "FFT benchmark results";
  return t;
}
inline void set_title(const std::string& t) { title() = t; }
inline void push(const RunStat& s) { rows().push_back(s); }

inline void print_table() {
  std::printf("\n== %s ==\n", title().c_str());
  std::printf("%-20s %7s %10s %9s %12s %10s %10s %11s %9s %7s\n", "type", "N", "radix", "batch", "max_abs_err",
              "compute_ms", "e2e_ms", "GFLOP/s", "GB/s", "%HBM");
  for (const auto& s : rows())
    std::printf("%-20s %7u %10s %9lld %12.3e %10.4f %10.4f %11.1f %9.1f %6.1f%%\n", s.type.c_str(), s.N,
                s.radix.c_str(), s.B, s.max_abs_err, s.compute_ms, s.e2e_ms, s.gflops, s.gbps, 100.0 * s.hbm_frac);
}

inline void print_csv() {
  std::printf("type,N,radix,B,max_abs_err,compute_ms,e2e_ms,gflops,gbps,hbm_frac\n");
  for (const auto& s : rows())
    std::printf("%s,%u,%s,%lld,%.6e,%.6f,%.6f,%.3f,%.3f,%.5f\n", s.type.c_str(), s.N, s.radix.c_str(), s.B, s.max_abs_err,
                s.compute_ms, s.e2e_ms, s.gflops, s.gbps, s.hbm_frac);
}

inline void print_ndjson() {
  // JSON has no NaN: figures that were not measured (compute_ms of a comparator row, an error without a
  // comparator) are written as null
  auto num = [](double v, const char* fmt) {
    char buf[64];
    if (v != v) return std::string("null");
    std::snprintf(buf, sizeof buf, fmt, v);
    return std::string(buf);
  };
  for (const auto& s : rows())
    std::printf("{\"type\":\"%s\",\"N\":%u,\"radix\":\"%s\",\"B\":%lld,\"max_abs_err\":%s,\"compute_ms\":%s,\"e2e_ms\":%.6f,"
                "\"gflops\":%.3f,\"gbps\":%.3f,\"hbm_frac\":%.5f}\n",
                s.type.c_str(), s.N, s.radix.c_str(), s.B, num(s.max_abs_err, "%.6e").c_str(),
                num(s.compute_ms, "%.6f").c_str(), s.e2e_ms, s.gflops, s.gbps, s.hbm_frac);
}

}  // namespace stat
