#!/usr/bin/env python3
"""The "ours" column of the reference's Figure-5 sweep on MI355X.

The reference's driver (Figure5_1dFFT/benchmark_figure5_1dfft.py) times four implementations per filter
length; three of them (FlashFFTConv, its depthwise conv1d, cuDNN through torch) are external NVIDIA-side
code and are out of scope here. This script runs the fourth -- the fused FFT-convolution binary -- over the
same sweep (Figure5_1dFFT/run_figure5.sh:25-38: L = 4096, B = 64, D = 768, K = 3 .. 63 step 2, warmup 5,
runs 10, real input) with the same 11 positional arguments (:38-51 of the driver) and parses the same stdout
line (:27-34). The CSV keeps the driver's column name for this column (:162).

Usage: python tools/figure5_sweep.py [--B 64] [--D 768] [--k-start 3] [--k-end 63] [--k-step 2] [--csv out.csv]
"""
import argparse
import csv
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LINE = re.compile(
    r"ThunderFFT full FFT-Conv \(N=4096, include_filter_fft=[01], filter_fft_mode=[01]\) time:\s*([0-9.]+)\s*ms")


def run_ours(binary, k, device, b, d, warmup, runs, include_filter_fft, filter_fft_mode, real_input, real_filter):
    cmd = [binary, k, device, b, d, warmup, runs, include_filter_fft, 0, filter_fft_mode, real_input, real_filter]
    out = subprocess.check_output([str(c) for c in cmd], text=True)
    m = LINE.search(out)
    if not m:
        raise RuntimeError(f"no timing line in:\n{out}")
    return float(m.group(1))


def main():
    p = argparse.ArgumentParser()
    p.add_argument("--B", type=int, default=64)
    p.add_argument("--D", type=int, default=768)
    p.add_argument("--k-start", type=int, default=3)
    p.add_argument("--k-end", type=int, default=63)
    p.add_argument("--k-step", type=int, default=2)
    p.add_argument("--warmup", type=int, default=5)
    p.add_argument("--runs", type=int, default=10)
    p.add_argument("--device", type=int, default=0)
    p.add_argument("--ours-bin", default=os.path.join(ROOT, "bench", "thff_conv1d_fullfft4096"))
    p.add_argument("--ours-include-filter-fft", type=int, default=1)
    p.add_argument("--ours-filter-fft-mode", type=int, default=1)
    p.add_argument("--ours-real-filter", type=int, default=1)
    p.add_argument("--csv", default="")
    a = p.parse_args()
    if not os.path.exists(a.ours_bin):
        sys.exit(f"{a.ours_bin} is missing: run `make bench`")
    rows = []
    for k in range(a.k_start, a.k_end + 1, a.k_step):
        ms = run_ours(a.ours_bin, k, a.device, a.B, a.D, a.warmup, a.runs, a.ours_include_filter_fft,
                      a.ours_filter_fft_mode, 1, a.ours_real_filter)
        rows.append((k, ms))
        print(f"k={k:4d} ours_complex_realinput={ms:.6f}ms", flush=True)
    if a.csv:
        with open(a.csv, "w", newline="") as f:
            w = csv.writer(f)
            w.writerow(["k", "ours_complex_realinput"])
            w.writerows(rows)
        print(f"saved: {a.csv}")


if __name__ == "__main__":
    main()
