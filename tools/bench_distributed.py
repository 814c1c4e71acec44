#!/usr/bin/env python3
"""Times one large distributed 1D FFT (four-step, RCCL all-to-all over xGMI).

    python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 \
        tools/bench_distributed.py --log2n 30 [--layout natural|transposed]
    python tools/bench_distributed.py --log2n 28                      # one GPU, P = 1: the local passes alone
    python tools/bench_distributed.py --log2n 28 --virtual-ranks 8    # one GPU runs all 8 ranks' passes in turn

Every rank holds N/P points (2^30 on 8 GPUs: 1 GiB of fp32 complex per GPU). Prints one JSON line on rank 0:
time per transform (max over ranks), GFLOP/s (5 N log2 N), the passes each rank made over its shard (one read +
one write each) and the bytes each rank moved through all-to-all. With --virtual-ranks the exchanges are
device-to-device chunk copies and the time is the sum over the ranks: what it prices is the local passes.
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
    ap.add_argument("--virtual-ranks", type=int, default=0)
    args = ap.parse_args()

    import torch
    import torch.distributed as dist

    from fft_amd import sharding
    from fft_amd.distributed import DistributedFFT, HipOps, run_virtual

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        dist.init_process_group("nccl", device_id=dev)
    n = 1 << args.log2n
    lay = dict(input_layout=args.layout, output_layout=args.layout)
    virt = args.virtual_ranks if world == 1 else 0
    ranks = virt or world

    if virt:
        ops = HipOps(dev)
        plans = [DistributedFFT(n, ops=ops, rank=r, world=virt) for r in range(virt)]
        g = torch.Generator(device=dev).manual_seed(0)
        xs = [torch.rand(n // virt, 2, device=dev, generator=g) * 2 - 1 for _ in range(virt)]

        def step():
            run_virtual(plans, xs, True, **lay)

        plan = plans[0]
    else:
        plan = DistributedFFT(n)
        g = torch.Generator(device=dev).manual_seed(rank)
        x = torch.rand(n // world, 2, device=dev, generator=g) * 2 - 1

        def step():
            plan.transform(x, True, **lay)

    wall = sharding.timed_steps(step, args.steps, args.warmup, dist if world > 1 else None,
                                lambda: torch.cuda.synchronize(dev), dev)
    if rank == 0:
        sec = wall / args.steps
        shard = n // ranks
        exchanges = (3 if args.layout == "natural" else 1) if ranks > 1 else 0
        line = {
            "workload": f"distributed four-step 1D C2C FFT, N=2^{args.log2n} fp32, {args.layout} layout",
            "n_gpus": world,
            "virtual_ranks": virt,
            "ms_per_transform": round(sec * 1e3, 3),
            "gflops": round(5.0 * n * math.log2(n) / sec * 1e-9, 1),
            "passes_over_shard_per_rank": plan.passes,
            "all_to_all_exchanges": exchanges,
            "shard_bytes": shard * 8,
            "bytes_sent_per_rank_per_exchange": shard * 8 * (ranks - 1) // ranks if ranks > 1 else 0,
        }
        if virt or world == 1:
            # one device did every rank's passes (and, virtual, the exchange copies): bytes it moved per second
            moved = 16.0 * n * (plan.passes + exchanges)
            line["device_GBps_over_all_passes"] = round(moved / sec * 1e-9, 1)
        print(json.dumps(line))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
