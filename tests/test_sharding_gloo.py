"""CPU tests of the N > 1 path with two gloo ranks: batch sharding reproduces the single-process
result, and the timing helper reduces with MAX over ranks. The local transform is the CPU oracle
here (no GPU in this container); on the GPU box the same plumbing drives the HIP path."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

from fft_amd import sharding, synth  # noqa: E402


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, total, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import oracle

        x = synth.uniform_complex(total, n, seed=17)
        start, count = sharding.shard_range(total, world, rank)
        calls = []

        def step():
            calls.append(1)
            return oracle.fft(x[start:start + count], True)

        wall = sharding.timed_steps(step, steps=2, warmup=1, dist=dist)
        assert len(calls) == 3
        y = step()
        np.save(os.path.join(out_dir, f"part{rank}.npy"), y)
        # MAX reduce: rank r contributes r + 1
        got = sharding.max_over_ranks([float(rank + 1), wall], dist)
        assert got[0] == float(world)
        assert got[1] >= wall - 1e-12
    finally:
        dist.destroy_process_group()


def test_shard_range_partitions_exactly():
    for total in (0, 1, 7, 64, 65536, 65537):
        for world in (1, 2, 3, 8):
            covered = []
            for r in range(world):
                s, c = sharding.shard_range(total, world, r)
                covered += list(range(s, s + c))
                assert c in (total // world, total // world + 1)
            assert covered == list(range(total))
    with pytest.raises(ValueError):
        sharding.shard_range(8, 2, 2)


def test_aggregate_gflops():
    # 8 ranks x 65536 transforms of 4096 points in 0.75 ms
    v = sharding.aggregate_gflops(4096, 65536, 8, 0.75e-3)
    assert abs(v - 8 * 5 * 4096 * 12 * 65536 / 0.75e-3 * 1e-9) < 1e-6 * v


def test_two_rank_batch_sharding_matches_single_process(tmp_path, oracle_mod):
    n, total, world = 256, 37, 2  # ragged: 19 + 18
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, total, str(tmp_path)), nprocs=world, join=True)
    parts = [np.load(tmp_path / f"part{r}.npy") for r in range(world)]
    assert [p.shape[0] for p in parts] == [19, 18]
    whole = oracle_mod.fft(synth.uniform_complex(total, n, seed=17), True)
    assert np.array_equal(np.concatenate(parts, axis=0), whole)
