#!/usr/bin/env python3
"""Option sets of one (dtype, n) in BOTH regimes of the device: from idle (0.5 s host sleep, then 3 untimed + 20 timed
launches, the harness's default protocol) and settled (100 launches right after 60 more), alternating, three rounds.

    python tools/gpu_regime_ab.py --n 4096 d= p4=plan:4"""
import argparse
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fft_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--dtype", default="fp32")
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    n, batch = a.n, (1 << 28) // a.n
    dt = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}[a.dtype]
    x = ((torch.rand(batch, n, 2, device="cuda") * 2 - 1) * (1.0 if a.dtype == "fp32" else n ** -0.5)).to(dt)
    y = torch.empty_like(x)
    fft_amd.ThunderFFTInitialize(a.dtype, n)
    stream = torch.cuda.current_stream()
    variants = []
    for v in a.variants:
        name, _, body = v.partition("=")
        variants.append((name, {k: int(val) for k, val in (kv.split(":") for kv in body.split(",") if kv)} or None))

    def block(opts, warm, count):
        for _ in range(warm):
            fft_amd.ThunderFFT_global(x, y, batch, n, True, stream, options=opts)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record(stream)
        for _ in range(count):
            fft_amd.ThunderFFT_global(x, y, batch, n, True, stream, options=opts)
        e.record(stream)
        torch.cuda.synchronize()
        return s.elapsed_time(e) / count

    per_point = 16 if a.dtype == "fp32" else 8
    pct = lambda ms: per_point * n * batch / (ms * 1e-3) / 8e12 * 100  # noqa: E731
    for r in range(3):
        for name, opts in variants:
            time.sleep(0.5)
            cold = block(opts, 3, 20)
            settled = block(opts, 60, 100)
            print(f"{a.dtype} n={n} round {r} {name:10s} from idle 3+20: {cold:.4f} ms = {pct(cold):5.1f} %    settled 60+100: "
                  f"{settled:.4f} ms = {pct(settled):5.1f} %")


if __name__ == "__main__":
    main()
