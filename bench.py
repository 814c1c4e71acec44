#!/usr/bin/env python3
"""Headline benchmark: batched 1D C2C FFT, N=4096 fp32, 2^28 points per GPU, on MI355X.

    python bench.py --gpus N --steps K --warmup W [--scaling weak|strong] [--n LENGTH]

One "step" is one pass of the hot path (tfft_exec_c2c through the C ABI) over the whole batch,
from one device-resident buffer to another, the call shape of the reference's public entry point
(thunderfft/sample/fft1d/main.cu:116; `--in-place` selects the reference bench's in-place form,
thunderfft/sample/bench/baseline/thunderfft_bench.h:124-131).

N > 1: one process per GPU. Started under torch.distributed.run (WORLD_SIZE set) the script is one
rank; started plainly with --gpus N > 1 it first starts torch.distributed.run with N ranks of itself
as a CHILD process (decided from argv alone, before torch or the HIP library are imported: nothing
that has touched a GPU is ever replaced or re-launched) and relays rank 0's line. The batch
dimension shards trivially and no collective touches the data path:
  --scaling weak   every rank transforms its own 2^28 points (default)
  --scaling strong the 2^28 points (--batch transforms) are split over the ranks in contiguous shards
`n_gpus` is the all-reduced count of ranks that took part. Rank 0 prints ONE JSON line:

  value        whole-job GFLOP/s with the 5 N log2 N convention, inputs resident in HBM
  roofline     algorithmic HBM bytes (16 B per point: one read + one write) per launch divided
               by the mean launch duration -- one event pair on the launch stream around the K timed
               launches, elapsed / K -- against 8 TB/s; `traffic` is the PMC-measured HBM bytes per launch recorded under
               profiles/ for this same kernel (null if no such record matches). Multi-pass lengths
               (n > 16384) also report the per-sweep model: every sweep reads and writes all points.
  cpu_baseline the CPU FFT (scipy/pocketfft on complex64, one worker per host core of the box's share, the
               core count stated) timed on the whole workload; a reported baseline, not a target
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
POINTS = 1 << 28  # per launch: 2 GiB of fp32 complex (batch 65536 at N = 4096)
HBM_PEAK_GBPS = 8000.0
BYTES_PER_POINT = 16  # fp32 complex read once + written once


def host_cores():
    """Host threads this process may use: its CPU affinity, capped by the job's CPU share where the launcher
    exports one (OMP_NUM_THREADS on the GPU boxes, whose os.cpu_count() shows the whole machine)."""
    try:
        cores = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        cores = os.cpu_count() or 1
    share = os.environ.get("OMP_NUM_THREADS", "")
    if share.isdigit() and int(share) > 0:
        cores = min(cores, int(share))
    return max(1, cores)


def cpu_baseline(n, batch):
    """CPU FFT of the whole workload (same length, same batch, complex64) on the box's own host cores, best of 3:
    pocketfft through scipy.fft with one worker per core, and single-threaded numpy.fft beside it."""
    import numpy as np
    import scipy.fft

    from fft_amd import synth

    cores = host_cores()
    x = np.empty((batch, n), dtype=np.complex64)
    rows = max(1, (1 << 24) // n)  # generated 2^24 points at a time: the generator's temporaries stay small
    for first in range(0, batch, rows):
        x[first:first + rows] = synth.uniform_complex(min(rows, batch - first), n, seed=first)
    flops = 5.0 * n * math.log2(n) * batch

    def best(fn, reps):
        fn()
        t = []
        for _ in range(reps):
            t0 = time.perf_counter()
            fn()
            t.append(time.perf_counter() - t0)
        return min(t)

    t_multi = best(lambda: scipy.fft.fft(x, axis=-1, workers=cores), 3)
    t_single = best(lambda: np.fft.fft(x, axis=-1), 2)
    return {
        "value": round(flops / t_multi * 1e-9, 2),
        "unit": "GFLOP/s",
        "cores": cores,
        "kind": "port",
        "sample": f"the whole workload: scipy.fft.fft (pocketfft) complex64 N={n} batch={batch} "
                  f"({n * batch} points), workers={cores}, best of 3; {t_multi * 1e3:.1f} ms per pass",
        "host_cpus_visible": os.cpu_count(),
        "single_thread_numpy_gflops": round(flops / t_single * 1e-9, 2),
    }


def plan_signature(label, info):
    """What identifies the kernel a traffic record was taken on: its name and launch geometry."""
    return (f"{label} | threads={info.get('threads_per_block')} transforms={info.get('transforms_per_block')} "
            f"points/thread={info.get('points_per_thread')} lds={info.get('lds_bytes')}")


def recorded_traffic(kernel_tag, signature):
    """(HBM bytes per launch, where the figure comes from). The bytes are the PMC measurement committed under
    profiles/ (2 FETCH_SIZE + WRITE_SIZE of a rocprofv3 run of this script) and are reported only while that record
    describes the kernel this run launched; otherwise null, with the reason."""
    path = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    try:
        rec = json.load(open(path))
    except (OSError, ValueError):
        return None, "no record: profiles/pmc_traffic.json missing"
    entry = rec.get(kernel_tag)
    if not entry:
        return None, f"no PMC record for {kernel_tag}"
    if entry.get("signature") != signature:
        return None, (f"stale: {entry.get('source')} was taken on [{entry.get('signature')}], "
                      f"this run launched [{signature}]")
    return entry.get("hbm_bytes_per_launch"), entry.get("source")


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None,
                    help="timed launches (default 20; 100 for 16-bit storage and for fp32 N = 16384)")
    ap.add_argument("--warmup", type=int, default=None,
                    help="untimed launches before them (default 3; 50 for 16-bit storage and fp32 N = 16384, whose kernels "
                         "take tens of milliseconds of sustained work to reach their rate)")
    ap.add_argument("--n", "--fft-len", dest="n", type=int, default=N_FFT,
                    help="transform length (use --fft-len behind torch.distributed.run, whose own parser takes "
                         "--n for an abbreviation)")
    ap.add_argument("--batch", type=int, default=0,
                    help="transforms per rank (weak) or in total (strong); default 2^28 / n")
    ap.add_argument("--scaling", choices=("weak", "strong"), default="weak")
    ap.add_argument("--in-place", action="store_true", help="transform the buffer in place (the reference bench protocol) instead of x -> y")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--dtype", choices=("fp32", "fp16", "bf16"), default="fp32",
                    help="storage type; fp32 is the headline configuration, fp16 / bf16 the 16-bit one (8 B per point)")
    ap.add_argument("--out-of-place", action="store_true", help="x -> y: the default (kept so that older command lines still parse)")
    ap.add_argument("--event-per-step", action="store_true",
                    help="bracket every launch with its own event pair instead of one pair around the timed region")
    ap.add_argument("--stub-step", action="store_true",
                    help="launcher self-test without a GPU: gloo ranks, a sleep as the step; the line says data=stub")
    args = ap.parse_args(argv)
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")
    if args.batch <= 0:
        args.batch = max(1, POINTS // args.n)
    # 3 + 20 launches of a 16-bit configuration read 67-70 % of 8 TB/s where 50 + 100 read 78-80 %: the device is
    # still settling. The fp32 kernels up to N = 8192 show no such ramp (77.6 % at 3 + 20, 76.4 % at 50 + 100); the
    # N = 16384 kernel (two workgroups per CU, the one length that is latency- rather than bandwidth-limited) does:
    # 72 % over the first 3 + 20 launches of a process, 75.2-75.5 % at 50 + 100, three processes in a row each way.
    settled = args.dtype != "fp32" or args.n == 16384
    if args.steps is None:
        args.steps = 100 if settled else 20
    if args.warmup is None:
        args.warmup = 50 if settled else 3
    # x -> y for every storage type (the call shape of the reference's fft1d sample, sample/fft1d/main.cu:116);
    # --in-place is the reference bench's form (sample/bench/baseline/thunderfft_bench.h:124-125). Once the device
    # has settled the two forms run at the same rate for all three storage types.
    args.out_of_place = not args.in_place
    return args


def launch_ranks(args, argv):
    """Start `--gpus` ranks of this script under torch.distributed.run as a child process, pass rank 0's JSON line
    through, return the child's exit code. Called before anything GPU-related is imported."""
    import socket
    import subprocess

    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           "--",  # everything after it is the script and its arguments (its own --n would read as an abbreviation)
           os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True)
    for line in proc.stdout:
        # only the result line goes to stdout; anything else a rank printed there goes to stderr
        (sys.stdout if line.startswith('{"metric"') else sys.stderr).write(line)
    return proc.wait()


def kernel_label(dtype, n, radix):
    if dtype == "fp32":
        if n > 16384:
            return (f"tfft::fft_tile_kernel x {len(radix)} sweeps ({'x'.join(map(str, radix))}), forward: "
                    "strided column sweeps with four-step twiddles, last sweep writes natural order")
        if n == 8192:
            return "tfft::fft_dit_kernel<Plan4096, forward> (two interleaved 4096-point transforms per workgroup)"
        if n == 16384:
            return f"tfft::fft_lds_kernel<Plan{n}, StoreF32, forward, straight-line>"
        return f"tfft::fft_lds_kernel<Plan{n}, StoreF32, forward>"
    if 64 <= n <= 4096:
        return f"tfft::fft_mfma4096_kernel<{dtype}, forward, N={n}>"
    if n > 4096:
        return f"tfft::fft_mfma_big_kernel<{dtype}, forward, N={n}>"
    return f"tfft::fft_lds_kernel<Plan{n}, {dtype} storage, forward>"


def main(argv=None):
    argv = sys.argv[1:] if argv is None else argv
    args = parse_args(argv)
    env_world = os.environ.get("WORLD_SIZE")
    if env_world is None and args.gpus > 1:
        raise SystemExit(launch_ranks(args, argv))
    world = int(env_world or "1")
    if world != args.gpus:
        raise SystemExit(f"bench.py: --gpus {args.gpus} but WORLD_SIZE={world}; start it plainly (it launches its own "
                         f"ranks) or with --nproc-per-node {args.gpus}")
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))

    import torch

    from fft_amd import sharding

    stub = args.stub_step
    if stub:
        dev = torch.device("cpu")
    else:
        if not torch.cuda.is_available():
            raise SystemExit("bench.py needs a GPU: the FFT path has no CPU fallback")
        torch.cuda.set_device(local_rank)
        dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        if stub:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)
    # ranks that really take part (not the number asked for)
    ones = torch.ones(1, dtype=torch.int64, device=dev)
    if dist:
        dist.all_reduce(ones)
    n_ranks = int(ones.item())

    n = args.n
    if args.scaling == "strong":
        total_batch = args.batch
        _, batch = sharding.shard_range(total_batch, world, rank)
    else:
        batch = args.batch
        total_batch = batch * world

    if stub:
        info = {"radix": [n]}

        def step():
            time.sleep(0.002)

        def rescale_due():
            return False

        def rescale():
            pass

        def sync():
            pass

        stream = None
    else:
        import fft_amd

        # synthetic input, uniform(-1, 1), generated on the device (values do not affect timing)
        g = torch.Generator(device=dev).manual_seed(1234 + rank)
        x = torch.rand(max(batch, 1), n, 2, device=dev, generator=g) * 2 - 1
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
        info = fft_amd.plan_info(args.dtype, n)
        stream = torch.cuda.current_stream(dev)

        def step():
            fft_amd.ThunderFFT_global(x, y, batch, n, True, stream)

        done = 0

        def rescale_due():
            # in place only, on runs longer than RESCALE_EVERY passes
            nonlocal done
            done += 1
            return not args.out_of_place and done % RESCALE_EVERY == 0

        def rescale():
            x.mul_(float(n) ** (-RESCALE_EVERY / 2.0))

        def sync():
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
        if rescale_due():
            rescale()
    sharding.barrier(dist, sync)

    # Events on the launch stream bracket the timed region: the K launches run back to back, as a caller's would,
    # and the mean launch duration is the region's elapsed time / K (launch gaps included). An in-place run longer
    # than RESCALE_EVERY passes closes the pair around each rescaling of its data, so that the event time holds
    # transform launches only (the wall clock, and with it `value`, includes those extra kernels).
    # --event-per-step brackets every launch with its own pair instead; the extra packets between launches cost
    # each launch 10-20 us.
    per_step = args.event_per_step and not stub
    starts, ends = [], []

    def open_pair():
        if not stub:
            starts.append(torch.cuda.Event(enable_timing=True))
            starts[-1].record(stream)

    def close_pair():
        if not stub:
            ends.append(torch.cuda.Event(enable_timing=True))
            ends[-1].record(stream)

    t0 = time.perf_counter()
    if not per_step:
        open_pair()
    for i in range(args.steps):
        if per_step:
            open_pair()
        step()
        if per_step:
            close_pair()
        if rescale_due():
            if not per_step:
                close_pair()
            rescale()
            if not per_step:
                open_pair()
    if not per_step:
        close_pair()
    sharding.barrier(dist, sync)
    wall = time.perf_counter() - t0

    kernel_ms = (wall / args.steps * 1e3 if stub
                 else sum(s.elapsed_time(e) for s, e in zip(starts, ends)) / args.steps)
    wall, kernel_ms = sharding.max_over_ranks([wall, kernel_ms], dist, dev)

    # The same kernel with its butterflies and LDS exchanges compiled out (tfft_bench_access_pattern, pattern 1: the
    # rows are loaded and stored in the transform's access pattern, the output is meaningless): what this workgroup
    # shape can stream at all. Its own entry point, outside and behind the timed region -- so on a device that has
    # settled, where the N = 4096 transform itself reads 76-77 %; fp32 lengths that have the build only.
    shape_ms = None
    # (N = 8192 has no such build of its default kernel, fft_dit_kernel: the figure would be another kernel's.)
    if not stub and args.dtype == "fp32" and n <= 16384 and n != 8192 and batch > 0:
        try:
            scratch = y if args.out_of_place else torch.empty_like(x)  # never the benchmark's live data in place
            fft_amd.bench_access_pattern(x, scratch, batch, n, 1, True, stream)
            sync()
            # timed like the transform itself: launches back to back inside one event pair
            reps = 10
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record(stream)
            for _ in range(reps):
                fft_amd.bench_access_pattern(x, scratch, batch, n, 1, True, stream)
            b.record(stream)
            sync()
            shape_ms = a.elapsed_time(b) / reps
        except fft_amd.ThunderFFTError:
            shape_ms = None

    ms_per_step = wall / args.steps * 1e3
    per_point = BYTES_PER_POINT if args.dtype == "fp32" else BYTES_PER_POINT // 2
    # the slowest rank's launch moved that rank's shard; in both scaling modes rank 0 holds a largest shard
    bytes_per_launch = per_point * n * batch
    achieved = bytes_per_launch / (kernel_ms * 1e-3) * 1e-9
    sweeps = len(info["radix"]) if n > 16384 else 1

    if rank == 0:
        label = kernel_label(args.dtype, n, info["radix"])
        signature = plan_signature(label, info)
        tag = (f"fft_lds_kernel_f32_n{n}_b{batch}" if args.dtype == "fp32" else f"fft_kernel_{args.dtype}_n{n}_b{batch}")
        traffic, traffic_source = (None, "stub") if stub else recorded_traffic(tag, signature)
        line = {
            "metric": f"batched 1D C2C FFT GFLOP/s (5NlogN) + % HBM roofline, N={n} {args.dtype}",
            "value": round(5.0 * n * math.log2(n) * total_batch / (ms_per_step * 1e-3) * 1e-9, 1),
            "unit": "GFLOP/s",
            "n_gpus": n_ranks,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4),
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            # fp32 storage: fp32 VALU arithmetic; 16-bit storage: 16-bit matrix-core operands, fp32 accumulation
            "dtype": {"fp32": "f32", "fp16": "f16", "bf16": "bf16"}[args.dtype],
            "data": "stub (no GPU work: launcher self-test)" if stub else "synthetic",
            "config": {
                "workload": f"batched 1D C2C FFT N={n} batch={batch} per GPU {args.dtype}, "
                            f"{'out of place' if args.out_of_place else 'in place'}, forward",
                "points_per_gpu": n * batch,
                "total_batch": total_batch,
                "radix": "x".join(str(r) for r in info["radix"]),
                "parallelism": f"batch-sharded x{n_ranks} ({args.scaling} scaling), no data-path collective",
            },
            "roofline": {
                "bound": "hbm",
                "achieved": round(achieved, 1),
                "peak": HBM_PEAK_GBPS,
                "unit": "GB/s",
                "frac": round(achieved / HBM_PEAK_GBPS, 4),
                "traffic": traffic,
                "traffic_source": traffic_source,
                "kernel": label,
                "kernel_signature": signature,
                "kernel_ms": round(kernel_ms, 4),
                "kernel_ms_from": ("one HIP event pair per launch" if args.event_per_step
                                   else f"one HIP event pair around the {args.steps} timed launches / {args.steps}"),
                "algorithmic_bytes_per_launch": bytes_per_launch,
                # fraction of peak the kernel's own load/store pattern reaches with the arithmetic compiled out
                "load_store_only_frac": (round(bytes_per_launch / (shape_ms * 1e-3) * 1e-9 / HBM_PEAK_GBPS, 4)
                                         if shape_ms else None),
            },
        }
        if sweeps > 1:
            # multi-pass transform: the compulsory-bytes fraction above cannot exceed 1 / sweeps; this is the
            # fraction of peak at which the sweeps themselves stream (every sweep reads and writes every point)
            line["roofline"]["sweeps"] = sweeps
            line["roofline"]["sweep_model_bytes_per_launch"] = sweeps * bytes_per_launch
            line["roofline"]["frac_sweep_model"] = round(sweeps * achieved / HBM_PEAK_GBPS, 4)
        if not args.no_cpu_baseline and world == 1 and not stub:
            line["cpu_baseline"] = cpu_baseline(n, batch)
        print(json.dumps(line), flush=True)

    if not stub:
        fft_amd.ThunderFFTFinalize(args.dtype)
    if dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
