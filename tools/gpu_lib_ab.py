#!/usr/bin/env python3
"""A/B of two builds of libthunderfft_hip.so: times the default kernel of every length with each build in turn
(one child process per build and pass, THUNDERFFT_HIP_LIB selects the file), 2^28 points, HIP events around groups
of launches, and prints % of 8 TB/s per build side by side.

    python tools/gpu_lib_ab.py --old ab/libthunderfft_hip_old.so [--dtype fp32] [--passes 2] [--in-place]
"""
import argparse
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(a):
    sys.path.insert(0, ROOT)
    import torch

    import fft_amd

    dt = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}[a.dtype]
    out = {}
    stream = torch.cuda.current_stream()
    for n in a.lengths:
        batch = a.points // n
        x = ((torch.rand(batch, n, 2, device="cuda") * 2 - 1) * (1.0 if a.dtype == "fp32" else n ** -0.5)).to(dt)
        y = x if a.in_place else torch.empty_like(x)
        fft_amd.ThunderFFTInitialize(a.dtype, n)
        keep = x.clone() if a.in_place else None
        for _ in range(3):
            fft_amd.ThunderFFT_global(x, y, batch, n, True, stream)
        ts = []
        for _ in range(a.groups):
            if keep is not None:
                x.copy_(keep)
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(stream)
            for _ in range(a.burst):
                fft_amd.ThunderFFT_global(x, y, batch, n, True, stream)
            e.record(stream)
            torch.cuda.synchronize()
            ts.append(s.elapsed_time(e) / a.burst)
        per_point = 16 if a.dtype == "fp32" else 8
        ts.sort()
        med = ts[len(ts) // 2]
        out[n] = [round(per_point * a.points / (med * 1e-3) / 8e12 * 100, 2),
                  round(per_point * a.points / (ts[0] * 1e-3) / 8e12 * 100, 2)]
        del x, y, keep
        torch.cuda.empty_cache()
    print("RESULT " + json.dumps(out))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--old", default=os.path.join(ROOT, "ab", "libthunderfft_hip_old.so"))
    ap.add_argument("--more", default="", help="further builds, name=path[,name=path...]")
    ap.add_argument("--dtype", default="fp32")
    ap.add_argument("--points", type=int, default=1 << 28)
    ap.add_argument("--lengths", type=lambda s: [int(v) for v in s.split(",")],
                    default=[64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384])
    ap.add_argument("--groups", type=int, default=12)
    ap.add_argument("--burst", type=int, default=5, help="launches per timed group (in place: 1 is forced)")
    ap.add_argument("--passes", type=int, default=2)
    ap.add_argument("--in-place", action="store_true")
    ap.add_argument("--child", action="store_true")
    a = ap.parse_args()
    if a.in_place:
        a.burst = 1
    if a.child:
        return child(a)
    builds = [("old", a.old), ("new", os.path.join(ROOT, "fft_amd", "lib", "libthunderfft_hip.so"))]
    builds += [tuple(kv.split("=")) for kv in a.more.split(",") if kv]
    res = {name: [] for name, _ in builds}
    for _ in range(a.passes):
        for name, path in builds:
            env = dict(os.environ, THUNDERFFT_HIP_LIB=path)
            cmd = [sys.executable, os.path.abspath(__file__), "--child", "--dtype", a.dtype, "--points", str(a.points),
                   "--lengths", ",".join(map(str, a.lengths)), "--groups", str(a.groups), "--burst", str(a.burst)]
            if a.in_place:
                cmd.append("--in-place")
            p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=500)
            line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
            if p.returncode != 0 or not line:
                print(p.stdout[-2000:])
                raise SystemExit(1)
            res[name].append(json.loads(line[0][7:]))
    print(f"{a.dtype}, {'in place' if a.in_place else 'x -> y'}, 2^{a.points.bit_length() - 1} points, median (best) % of 8 TB/s "
          f"over {a.groups} groups of {a.burst} launches, per pass")
    print(f"{'N':>6} " + "".join(f"{name[:14] + ' #' + str(i + 1):>22}" for i in range(a.passes) for name, _ in builds))
    for n in a.lengths:
        row = f"{n:>6} "
        for i in range(a.passes):
            for name, _ in builds:
                med, best = res[name][i][str(n)]
                row += f"{med:>14.2f} ({best:5.2f})"
        print(row)


if __name__ == "__main__":
    main()
