"""Multi-GPU plumbing for the batched path: one process per GPU, the batch dimension sharded.

Batched transforms are independent units, so ranks take contiguous slices of the batch and no
collective touches the data path; torch.distributed is used only for the barriers and the
max-over-ranks reduction of the timings (backend "nccl" = RCCL on the GPU box, "gloo" in the
CPU tests). The reference has no multi-GPU code at all (single cudaSetDevice per sample,
thunderfft/sample/fft1d/main.cu:80).
"""
import time


def shard_range(total, world, rank):
    """Contiguous, balanced slice [start, start + count) of `total` batch entries for `rank`."""
    if world < 1 or not (0 <= rank < world) or total < 0:
        raise ValueError(f"bad shard request total={total} world={world} rank={rank}")
    base, extra = divmod(total, world)
    start = rank * base + min(rank, extra)
    return start, base + (1 if rank < extra else 0)


def barrier(dist, sync=None):
    """Device sync (if any) + process-group barrier + device sync: both sides of a timed region."""
    if sync is not None:
        sync()
    if dist is not None and dist.is_initialized() and dist.get_world_size() > 1:
        dist.barrier()
    if sync is not None:
        sync()


def max_over_ranks(values, dist, device=None):
    """Element-wise maximum of a list of floats over all ranks (identity for one rank)."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return list(values)
    import torch

    t = torch.tensor(list(values), dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(v) for v in t.cpu()]


def timed_steps(step, steps, warmup, dist=None, sync=None, device=None):
    """`warmup` untimed calls, then exactly `steps` calls bracketed by barrier(); returns the
    wall time of the timed region in seconds, maximised over ranks."""
    for _ in range(warmup):
        step()
    barrier(dist, sync)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier(dist, sync)
    wall = time.perf_counter() - t0
    return max_over_ranks([wall], dist, device)[0]


def aggregate_gflops(n, batch_per_rank, world, seconds_per_step):
    """Whole-job throughput with the 5 N log2 N convention: every rank did `batch_per_rank` transforms."""
    import math

    return 5.0 * n * math.log2(n) * batch_per_rank * world / seconds_per_step * 1e-9
