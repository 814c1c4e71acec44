#!/usr/bin/env python3
"""Headline benchmark: batched 1D C2C FFT, N=4096 fp32, 2^28 points per GPU, on MI355X.

    python bench.py --gpus N --steps K --warmup W

One "step" is one pass of the hot path (tfft_exec_c2c through the C ABI) over the whole batch,
from one device-resident buffer to another, the call shape of the reference's public entry point
(thunderfft/sample/fft1d/main.cu:116; `--in-place` selects the reference bench's in-place form,
thunderfft/sample/bench/baseline/thunderfft_bench.h:124-131). For N > 1
the driver starts one process per GPU (torch.distributed.run); the batch dimension shards
trivially, every rank transforms its own 2^28 points (weak scaling) and no collective touches
the data path. Rank 0 prints ONE JSON line:

  value        whole-job GFLOP/s with the 5 N log2 N convention, inputs resident in HBM
  roofline     algorithmic HBM bytes (16 B per point: one read + one write) per launch divided
               by the mean launch duration measured with events on the launch stream, against
               8 TB/s; `traffic` is the PMC-measured HBM bytes per launch recorded under
               profiles/ for this same kernel (null if no such record matches)
  cpu_baseline the CPU FFT (scipy/pocketfft on complex64, all host cores of the box's share)
               timed on a bounded sample of the same workload; a reported baseline, not a target
"""
import argparse
import json
import math
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

N_FFT = 4096
BATCH = 65536  # 2^28 points: 2 GiB of fp32 complex per GPU
HBM_PEAK_GBPS = 8000.0
BYTES_PER_POINT = 16  # fp32 complex read once + written once


def cpu_baseline(n, seconds_budget=12.0):
    """CPU FFT of the same shape on a bounded sample, best of 3."""
    import numpy as np
    import scipy.fft

    import oracle  # noqa: F401  (the CPU side lives with the oracle; this leg is its only bench user)
    from fft_amd import synth

    cores = max(1, min(16, os.cpu_count() or 1))
    batch = 4096  # 2^24 points = 128 MiB of complex64
    x = synth.uniform_complex(batch, n, seed=0)
    flops = 5.0 * n * math.log2(n) * batch

    def best(fn, reps=3):
        fn()
        t = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            t.append(time.perf_counter() - t0)
        return min(t)

    t_multi = best(lambda: scipy.fft.fft(x, axis=-1, workers=cores))
    t_single = best(lambda: np.fft.fft(x, axis=-1))
    return {
        "value": round(flops / t_multi * 1e-9, 2),
        "unit": "GFLOP/s",
        "cores": cores,
        "kind": "port",
        "sample": f"scipy.fft.fft complex64 N={n} batch={batch} (2^24 points), workers={cores}, best of 3",
        "single_thread_numpy_gflops": round(flops / t_single * 1e-9, 2),
    }


def recorded_traffic(kernel_tag):
    """HBM bytes per launch from the committed PMC summary, if it describes this kernel."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return None
    entry = rec.get(kernel_tag)
    return entry.get("hbm_bytes_per_launch") if entry else None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--n", type=int, default=N_FFT)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--in-place", action="store_true", help="transform the buffer in place (the reference bench protocol) instead of x -> y")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", choices=("fp32", "fp16", "bf16"), default="fp32",
                    help="storage type; fp32 is the headline configuration, fp16 / bf16 the 16-bit one (8 B per point)")
    ap.add_argument("--out-of-place", action="store_true", help="force x -> y for 16-bit storage (default there: in place)")
    args = ap.parse_args()
    # fp32: x -> y (the call shape of the reference's fft1d sample; measured equal or faster than in place).
    # 16-bit storage: in place like the reference's bench (sample/bench/baseline/thunderfft_bench.h:124-125);
    # writing the rows just read measured 6 points faster than a separate output stream there.
    args.out_of_place = args.out_of_place or (args.dtype == "fp32" and not args.in_place)

    import torch

    import fft_amd
    from fft_amd import sharding

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the FFT path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        dist.init_process_group("nccl", device_id=dev)

    n, batch = args.n, args.batch
    # synthetic input, uniform(-1, 1), generated on the device (values do not affect timing)
    g = torch.Generator(device=dev).manual_seed(1234 + rank)
    x = torch.rand(batch, n, 2, device=dev, generator=g) * 2 - 1
    if args.dtype != "fp32":
        # 16-bit storage: inputs scaled by 1/sqrt(N) as the reference does (sample/bench/main.cu:36)
        x = (x * float(n) ** -0.5).to(torch.float16 if args.dtype == "fp16" else torch.bfloat16)
    # An unnormalised forward FFT grows random data by sqrt(N) per pass. In place, start low enough
    # (an exact power of two, mantissas untouched) that up to RESCALE_EVERY passes stay finite.
    RESCALE_EVERY = 32
    if not args.out_of_place:
        x.mul_(float(n) ** (-min(args.steps + args.warmup, RESCALE_EVERY) / 4.0))
    y = torch.empty_like(x) if args.out_of_place else x
    fft_amd.ThunderFFTInitialize(args.dtype, n)
    stream = torch.cuda.current_stream(dev)

    def step():
        fft_amd.ThunderFFT_global(x, y, batch, n, True, stream)

    done = 0

    def keep_finite():
        # only reached on runs longer than RESCALE_EVERY passes; outside the event-timed launches
        nonlocal done
        done += 1
        if not args.out_of_place and done % RESCALE_EVERY == 0:
            x.mul_(float(n) ** (-RESCALE_EVERY / 2.0))

    def sync():
        torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
        keep_finite()
    sharding.barrier(dist, sync)

    starts = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    ends = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps)]
    t0 = time.perf_counter()
    for i in range(args.steps):
        starts[i].record(stream)
        step()
        ends[i].record(stream)
        keep_finite()
    sharding.barrier(dist, sync)
    wall = time.perf_counter() - t0

    kernel_ms = sum(s.elapsed_time(e) for s, e in zip(starts, ends)) / args.steps
    wall, kernel_ms = sharding.max_over_ranks([wall, kernel_ms], dist, dev)

    # The same kernel with its butterflies and LDS exchanges compiled out (library option mode = 1: the rows are
    # loaded and stored in the transform's access pattern, results are meaningless): what this workgroup shape can
    # stream at all. Outside the timed region; fp32 lengths that have the ablation build only.
    shape_ms = None
    if args.dtype == "fp32":
        try:
            fft_amd.set_option("mode", 1)
            step()
            sync()
            ev = [(torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)) for _ in range(5)]
            for a, b in ev:
                a.record(stream)
                step()
                b.record(stream)
            sync()
            shape_ms = sum(a.elapsed_time(b) for a, b in ev) / len(ev)
        except fft_amd.ThunderFFTError:
            shape_ms = None
        finally:
            fft_amd.set_option("mode", 0)

    ms_per_step = wall / args.steps * 1e3
    bytes_per_launch = (BYTES_PER_POINT if args.dtype == "fp32" else BYTES_PER_POINT // 2) * n * batch
    achieved = bytes_per_launch / (kernel_ms * 1e-3) * 1e-9

    if rank == 0:
        info = fft_amd.plan_info(args.dtype, n)
        line = {
            "metric": f"batched 1D C2C FFT GFLOP/s (5NlogN) + % HBM roofline, N={n} {args.dtype}",
            "value": round(sharding.aggregate_gflops(n, batch, world, ms_per_step * 1e-3), 1),
            "unit": "GFLOP/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            # fp32 storage: fp32 VALU arithmetic; 16-bit storage at N >= 256: 16-bit matrix-core operands, fp32 accumulation
            "dtype": {"fp32": "f32", "fp16": "f16", "bf16": "bf16"}[args.dtype],
            "data": "synthetic",
            "config": {
                "workload": f"batched 1D C2C FFT N={n} batch={batch} per GPU {args.dtype}, "
                            f"{'out of place' if args.out_of_place else 'in place'}, forward",
                "points_per_gpu": n * batch,
                "radix": "x".join(str(r) for r in info["radix"]),
                "parallelism": f"batch-sharded x{world}, no data-path collective",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": recorded_traffic(f"fft_lds_kernel_f32_n{n}_b{batch}") if args.dtype == "fp32"
                else recorded_traffic(f"fft_kernel_{args.dtype}_n{n}_b{batch}"),
                "kernel": f"tfft::fft_lds_kernel<Plan{n}, StoreF32, forward>" if args.dtype == "fp32"
                else (f"tfft::fft_mfma4096_kernel<{args.dtype}, forward, N={n}>" if 256 <= n <= 4096
                      else f"tfft::fft_mfma_big_kernel<{args.dtype}, forward, N={n}>" if n > 4096
                      else f"tfft::fft_lds_kernel<Plan{n}, {args.dtype} storage, forward>"),
                "kernel_ms": round(kernel_ms, 4),
                "algorithmic_bytes_per_launch": bytes_per_launch,
                # fraction of peak the kernel's own load/store pattern reaches with the arithmetic compiled out
                "load_store_only_frac": (round(bytes_per_launch / (shape_ms * 1e-3) * 1e-9 / HBM_PEAK_GBPS, 4)
                                         if shape_ms else None),
            },
        }
        if not args.no_cpu_baseline and world == 1:
            line["cpu_baseline"] = cpu_baseline(n)
        print(json.dumps(line), flush=True)

    fft_amd.ThunderFFTFinalize(args.dtype)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
