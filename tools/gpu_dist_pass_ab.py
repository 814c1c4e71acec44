#!/usr/bin/env python3
"""Per-pass times of the distributed transform's local kernels on one rank's shard (rank 0 of 8, N = 2^28 or
2^30), each pass timed on its own with HIP events, interleaved over the experiment settings in argv
("NAME=VALUE[,NAME=VALUE...]" sets those environment variables for a group of launches; "base" sets none)."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

from fft_amd.distributed import DistributedFFT, HipOps  # noqa: E402


def main():
    lg = int(os.environ.get("LOG2N", "28"))
    settings = sys.argv[1:] or ["base"]
    dev = torch.device("cuda", 0)
    ops = HipOps(dev)
    plan = DistributedFFT(1 << lg, ops=ops, rank=0, world=8)
    a, b = plan.buf
    a.uniform_(-1, 1)
    b.uniform_(-1, 1)
    x = torch.rand(plan.points, 2, device=dev) * 2 - 1
    P = plan.world
    passes = {
        "A-pack (transpose)": lambda: ops.permute(x, a, plan._spec_a_pack(), True),
        "1a radix-8 in place": lambda: ops.radix(b, b, plan._spec_1a(False), True),
        "1b rows R1": lambda: ops.fft_rows(b, P * plan.c2, plan.r1, True),
        "B-pack (twiddle + transpose)": lambda: ops.permute(b, a, plan._spec_b_pack(), True),
        "3a radix-8 in place": lambda: ops.radix(b, b, plan._spec_3a(), True),
        "C-pack (transpose)": lambda: ops.permute(b, a, plan._spec_c_pack(), True),
        "C-unpack (interleave)": lambda: ops.permute(b, a, plan._spec_c_unpack(), True),
    }
    touched = set()
    for s in settings:
        if s != "base":
            touched.update(kv.split("=")[0] for kv in s.split(","))
    bytes_per_pass = plan.points * 16
    reps, rounds = 10, 3
    times = {(s, k): [] for s in settings for k in passes}
    for _ in range(rounds):
        for s in settings:
            for k in touched:
                os.environ.pop(k, None)
            if s != "base":
                for kv in s.split(","):
                    k, v = kv.split("=")
                    os.environ[k] = v
            for name, fn in passes.items():
                fn()
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                for _ in range(reps):
                    fn()
                e1.record()
                torch.cuda.synchronize()
                times[(s, name)].append(e0.elapsed_time(e1) / reps)
                b.uniform_(-1, 1)  # in-place passes grow the data
    print(f"shard {plan.points} points ({bytes_per_pass / 2**20:.0f} MiB read + written per pass), us per pass (TB/s)")
    print(f"{'pass':32s}" + "".join(f"{s[:26]:>28s}" for s in settings))
    for name in passes:
        row = f"{name:32s}"
        for s in settings:
            t = min(times[(s, name)])
            row += f"{t * 1e3:14.1f} ({bytes_per_pass / (t * 1e-3) * 1e-12:5.2f})   "
        print(row)


if __name__ == "__main__":
    main()
