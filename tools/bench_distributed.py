#!/usr/bin/env python3
"""Times one large distributed 1D FFT (four-step, RCCL all-to-all over xGMI).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        tools/bench_distributed.py --log2n 30 [--layout natural|transposed]

Every rank holds N/P points (2^30 on 8 GPUs: 1 GiB of fp32 complex per GPU). Prints one JSON line
on rank 0: time per transform (max over ranks), GFLOP/s (5 N log2 N) and the bytes each rank moved
through all-to-all. Runs on a single GPU as well (P = 1, no exchange) for a smoke check.
"""
import argparse
import json
import math
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--log2n", type=int, default=30)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--layout", default="natural", choices=("natural", "transposed"))
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from fft_amd import sharding
    from fft_amd.distributed import DistributedFFT

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    n = 1 << args.log2n
    plan = DistributedFFT(n)
    g = torch.Generator(device=dev).manual_seed(rank)
    x = torch.rand(n // world, 2, device=dev, generator=g) * 2 - 1
    lay = dict(input_layout=args.layout, output_layout=args.layout)

    def step():
        plan.transform(x, True, **lay)

    wall = sharding.timed_steps(step, args.steps, args.warmup, dist if world > 1 else None,
                                lambda: torch.cuda.synchronize(dev), dev)
    if rank == 0:
        sec = wall / args.steps
        exchanges = 3 if args.layout == "natural" else 1
        print(json.dumps({
            "workload": f"distributed four-step 1D C2C FFT, N=2^{args.log2n} fp32, {args.layout} layout",
            "n_gpus": world,
            "ms_per_transform": round(sec * 1e3, 3),
            "gflops": round(5.0 * n * math.log2(n) / sec * 1e-9, 1),
            "all_to_all_exchanges": exchanges if world > 1 else 0,
            "bytes_sent_per_rank_per_exchange": (n // world) * 8 * (world - 1) // world,
        }))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
