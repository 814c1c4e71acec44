"""Interleaved A/B of the fused convolution's plans at the Figure-5 shape (B = 64, D = 768, N = 4096, fp16), on a
settled device: every plan is launched `burst` times back to back per round, event-timed as a group, after a check of
each plan's result against plan 0's.

    python tools/gpu_conv_ab.py [--kind 2] [--plans 0,5,6,7] [--rounds 10] [--burst 50]"""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fft_amd  # noqa: E402
from fft_amd import conv  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kind", type=int, default=2, help="0 complex, 1 real in, 2 real in and out")
    ap.add_argument("--plans", default="0,5,6,7")
    ap.add_argument("--B", type=int, default=64)
    ap.add_argument("--D", type=int, default=768)
    ap.add_argument("--rounds", type=int, default=10)
    ap.add_argument("--burst", type=int, default=50)
    a = ap.parse_args()
    N = 4096
    plans = [int(p) for p in a.plans.split(",")]
    g = torch.Generator(device="cuda").manual_seed(1)
    shape = (a.B, a.D, N) if a.kind else (a.B, a.D, N, 2)
    x = torch.rand(*shape, device="cuda", generator=g).to(torch.float16)
    h = torch.zeros(a.D, 33, 2, device="cuda")
    h[..., 0] = torch.rand(a.D, 33, device="cuda", generator=g) / N
    hp = conv.filter_spectra(h.to(torch.float16), N)
    outs = {}
    for p in plans:
        outs[p] = conv.fft_conv(x, hp, N, a.kind, options={"plan": p} if p else None)
    torch.cuda.synchronize()
    for p in plans[1:]:
        same = torch.equal(outs[p], outs[plans[0]])
        err = (outs[p].float() - outs[plans[0]].float()).abs().max().item()
        print(f"plan {p} vs plan {plans[0]}: {'bit-identical' if same else 'max abs diff %.3e' % err}")
    y = torch.empty_like(outs[plans[0]])
    bytes_moved = x.numel() * 2 + y.numel() * 2 + hp.numel() * 2
    stream = torch.cuda.current_stream()
    times = {p: [] for p in plans}
    for r in range(a.rounds + 2):
        for p in plans:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(stream)
            for _ in range(a.burst):
                conv.fft_conv(x, hp, N, a.kind, out=y, options={"plan": p} if p else None)
            e.record(stream)
            torch.cuda.synchronize()
            if r >= 2:
                times[p].append(s.elapsed_time(e) / a.burst)
    for p in plans:
        t = sum(times[p]) / len(times[p])
        print(f"kind {a.kind} B={a.B} D={a.D} plan {p}: {t:.4f} ms  {bytes_moved / (t * 1e-3) / 8e12 * 100:.1f} % of 8 TB/s (signals + filters once)")


if __name__ == "__main__":
    main()
