"""Interleaved A/B timing of tfft_exec_c2c_opt option sets for one (dtype, n): every variant is launched in
turn, ROUNDS times over, and the mean / min of its event-timed launches is printed as % of 8 TB/s.

    python tools/gpu_opt_sweep.py --n 16384 [--dtype fp32] [--in-place] name=key:val,key:val ...

A variant with no options (name=) is the default kernel."""
import argparse
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import fft_amd  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=16384)
    ap.add_argument("--dtype", default="fp32")
    ap.add_argument("--points", type=int, default=1 << 28)
    ap.add_argument("--rounds", type=int, default=30)
    ap.add_argument("--in-place", action="store_true")
    ap.add_argument("--burst", type=int, default=1,
                    help="launches per timed group, back to back with no host synchronisation in between (sustained "
                         "load); 1 = isolated launches")
    ap.add_argument("variants", nargs="+")
    a = ap.parse_args()
    n, batch = a.n, a.points // a.n
    dt = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}[a.dtype]
    x = ((torch.rand(batch, n, 2, device="cuda") * 2 - 1) * (1.0 if a.dtype == "fp32" else n ** -0.5)).to(dt)
    y = x if a.in_place else torch.empty_like(x)
    fft_amd.ThunderFFTInitialize(a.dtype, n)
    variants = []
    for v in a.variants:
        name, _, body = v.partition("=")
        opts = {k: int(val) for k, val in (kv.split(":") for kv in body.split(",") if kv)}
        variants.append((name, opts))
    stream = torch.cuda.current_stream()
    times = {name: [] for name, _ in variants}
    # in place: restore the input before every group (an unnormalised transform grows its data); bursts run on
    # whatever the previous launch left, pre-scaled so that the last pass stays finite
    keep = x.clone() if a.in_place else None
    if keep is not None and a.burst > 1:
        keep.mul_(float(n) ** (-a.burst / 2.0))
    for r in range(a.rounds + 2):
        for name, opts in variants:
            if keep is not None:
                x.copy_(keep)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(stream)
            for _ in range(a.burst):
                fft_amd.ThunderFFT_global(x, y, batch, n, True, stream, options=opts or None)
            e.record(stream)
            torch.cuda.synchronize()
            if r >= 2:
                times[name].append(s.elapsed_time(e) / a.burst)
    per_point = 16 if a.dtype == "fp32" else 8
    for name, _ in variants:
        t = times[name]
        mean, best = sum(t) / len(t), min(t)
        f = lambda ms: per_point * n * batch / (ms * 1e-3) / 8e12 * 100  # noqa: E731
        print(f"{a.dtype} n={n} {'in place' if a.in_place else 'out of place'} {name:28s} mean {mean:.4f} ms = {f(mean):5.1f} %   "
              f"best {best:.4f} ms = {f(best):5.1f} %")


if __name__ == "__main__":
    main()
