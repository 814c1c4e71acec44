"""One rank of the distributed transform on a real process group with DEVICE tensors: started by
tests/test_zy_distributed_ranks_gpu.py under torch.distributed.run (RANK / WORLD_SIZE / MASTER_* from the environment).

    python dist_rank_worker.py <backend> <log2 n> <out_dir>

Every rank uses cuda:0 (a box has one GPU): `gloo` ranks exchange their device chunks between processes through host
memory, `nccl` (= RCCL) accepts one rank per GPU only, so it runs with a world of one and sends its chunks to itself. Each rank draws its
shard of the same seeded signal, runs DistributedFFT.transform() -- HipOps passes, torch.distributed
all_to_all_single on the device buffers -- in the three layouts and both directions, and saves the results for the
parent to compare with the fp64 oracle."""
import faulthandler
import os
import sys
import time

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from fft_amd import sharding, synth  # noqa: E402
from fft_amd.distributed import DistributedFFT  # noqa: E402


def to_device(z, dev):
    return torch.from_numpy(np.stack([z.real, z.imag], axis=-1).astype(np.float32)).to(dev)


def say(rank, what):
    if os.environ.get("TFFT_TEST_VERBOSE"):
        print(f"[rank {rank}] {time.time() % 1000:8.2f} {what}", file=sys.stderr, flush=True)


def main():
    backend, lg, out_dir = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    # a rank that is still here after this long is stuck: say where, and end (the parent sees a failure, not a hang)
    faulthandler.dump_traceback_later(float(os.environ.get("TFFT_TEST_RANK_LIMIT", "80")), exit=True)
    n = 1 << lg
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    if backend == "nccl":
        dist.init_process_group("nccl", device_id=dev)
    else:
        dist.init_process_group(backend)
    try:
        # the collectives bench.py relies on, on device tensors of this backend
        if backend == "gloo":
            # gloo moves host memory: the chunks go device -> host -> the other processes -> device. (Handing gloo
            # the device tensors themselves works with two ranks; with four ranks on one GPU its device-tensor
            # all-to-all was seen to stall inside gloo with every rank's device idle, so the test does not rely on it.)
            real = dist.all_to_all_single

            def staged(recv, send, group=None):
                host_send = send.cpu()
                host_recv = torch.empty_like(host_send)
                real(host_recv, host_send, group=group)
                recv.copy_(host_recv)

            dist.all_to_all_single = staged
        say(rank, "group up")
        cdev = dev if backend == "nccl" else torch.device("cpu")  # where this backend reduces
        ones = torch.ones(1, dtype=torch.int64, device=cdev)
        dist.all_reduce(ones)
        assert int(ones.item()) == world
        sharding.barrier(dist, torch.cuda.synchronize)
        assert sharding.max_over_ranks([float(rank), 1.5], dist, cdev) == [float(world - 1), 1.5]
        # a plain exchange of device chunks: chunk d of rank s must arrive as chunk s of rank d
        send = (torch.arange(world * 4, device=dev, dtype=torch.float32) + 100 * rank).reshape(world * 4, 1).repeat(1, 2)
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send)
        want = torch.cat([torch.arange(rank * 4, rank * 4 + 4, device=dev, dtype=torch.float32) + 100 * s
                          for s in range(world)]).reshape(world * 4, 1).repeat(1, 2)
        assert torch.equal(recv, want)
        say(rank, "collectives ok")

        x = synth.uniform_complex(1, n, seed=21)[0]
        plan = DistributedFFT(n)
        say(rank, "plan built")
        assert (plan.rank, plan.world) == (rank, world)
        per = n // world
        mine = to_device(x[rank * per:(rank + 1) * per], dev)
        keep = mine.clone()
        cols = x.reshape(plan.n1, plan.n2)[:, rank * plan.c2:(rank + 1) * plan.c2].T.copy()  # [C2][N1]
        tin = to_device(cols, dev)
        for forward, tag in ((True, "f"), (False, "i")):
            nat = plan.transform(mine, forward).clone()  # results live in the plan's buffers until the next call
            assert plan.passes == (8 if world > 1 else 5)
            say(rank, f"natural {tag}")
            np.save(os.path.join(out_dir, f"nat_{tag}_{rank}.npy"), nat.cpu().numpy())
            tr = plan.transform(mine, forward, output_layout="transposed")
            np.save(os.path.join(out_dir, f"tr_{tag}_{rank}.npy"), tr.cpu().numpy())
            say(rank, f"transposed out {tag}")
            one = plan.transform(tin, forward, input_layout="transposed", output_layout="transposed")
            assert plan.passes == (5 if world > 1 else 4)
            np.save(os.path.join(out_dir, f"one_{tag}_{rank}.npy"), one.cpu().numpy())
            say(rank, f"transposed both {tag}")
            # the caller's own output buffer
            out = torch.full((per, 2), float("nan"), device=dev)
            got = plan.transform(mine, forward, out=out)
            assert got.data_ptr() == out.data_ptr() and torch.equal(got.reshape(per, 2).cpu(), nat.reshape(per, 2).cpu())
        assert torch.equal(mine, keep)  # the input is never modified
        np.save(os.path.join(out_dir, f"kmap_{rank}.npy"), plan.transposed_output_index())
        torch.cuda.synchronize()
        sharding.barrier(dist, torch.cuda.synchronize)
    finally:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
