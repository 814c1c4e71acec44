#!/usr/bin/env python3
"""How a kernel's launch time settles from a process's first launch: per-launch HIP-event times of the fp16 N = 4096
transform (default) for 160 launches, (a) from idle, (b) after 60 ms of another kernel's work (the fp32 N = 4096
transform on other buffers), (c) after a 300 ms host sleep in the middle. Prints the mean of launches 1-5, 6-20, 21-50,
51-100, 101-160 of each case."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fft_amd  # noqa: E402


def timed(fn, count, stream):
    ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(count)]
    for a, b in ev:
        a.record(stream)
        fn()
        b.record(stream)
    torch.cuda.synchronize()
    return [a.elapsed_time(b) for a, b in ev]


def buckets(ts):
    edges = [(0, 5), (5, 20), (20, 50), (50, 100), (100, 160)]
    return "  ".join(f"{lo + 1}-{hi}: {sum(ts[lo:hi]) / (hi - lo):.4f}" for lo, hi in edges)


def main():
    dtype = sys.argv[1] if len(sys.argv) > 1 else "fp16"
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    dt = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}[dtype]
    batch = (1 << 28) // n
    stream = torch.cuda.current_stream()
    x = ((torch.rand(batch, n, 2, device="cuda") * 2 - 1) * n ** -0.5).to(dt)
    y = torch.empty_like(x)
    fft_amd.ThunderFFTInitialize(dtype, n)
    x32 = torch.rand(65536, 4096, 2, device="cuda")
    y32 = torch.empty_like(x32)
    fft_amd.ThunderFFTInitialize("fp32", 4096)
    torch.cuda.synchronize()
    time.sleep(0.5)
    run = lambda: fft_amd.ThunderFFT_global(x, y, batch, n, True, stream)  # noqa: E731
    other = lambda: fft_amd.ThunderFFT_global(x32, y32, 65536, 4096, True, stream)  # noqa: E731
    print(f"{dtype} N={n}, ms per launch, mean over launch numbers")
    print("(a) from idle (0.5 s sleep before):      ", buckets(timed(run, 160, stream)))
    time.sleep(0.5)
    timed(other, 90, stream)
    print("(b) right after 90 fp32 N=4096 launches: ", buckets(timed(run, 160, stream)))
    time.sleep(0.3)
    print("(c) after a 0.3 s host sleep:            ", buckets(timed(run, 160, stream)))
    t = timed(run, 60, stream)
    time.sleep(0.02)
    print("(d) 60 launches, 20 ms sleep, 160 more:  ", buckets(timed(run, 160, stream)))


if __name__ == "__main__":
    main()
