"""CPU tests of the distributed four-step transform with gloo ranks: the pass descriptions (permutes with their
twiddles, radix-P passes over the exchanged chunks), the exchanges and the three layouts, with the fp64 mirror
of the passes (NumpyOps, the oracle as its local transform) standing in for the HIP kernels. The GPU tests run
the same descriptions through the kernels on virtual ranks."""
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

from fft_amd import synth  # noqa: E402
from fft_amd.distributed import DistributedFFT, run_virtual, split_n  # noqa: E402
from dist_reference import NumpyOps  # noqa: E402


def oracle_rows(x, forward):
    import oracle

    return oracle.fft(x, forward)


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def to_tensor(z):
    return torch.from_numpy(np.stack([z.real, z.imag], axis=-1).astype(np.float64))


def _c(a):
    a = np.asarray(a)
    return a[..., 0] + 1j * a[..., 1]


def _worker(rank, world, port, n, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x = synth.uniform_complex(1, n, seed=5)[0].astype(np.complex128)
        plan = DistributedFFT(n, ops=NumpyOps(oracle_rows))
        per = n // world
        t = to_tensor(x[rank * per:(rank + 1) * per])
        n1, n2 = plan.n1, plan.n2
        cols = x.reshape(n1, n2)[:, rank * plan.c2:(rank + 1) * plan.c2].T.copy()  # [C2][N1]: my columns, all n1
        tin = to_tensor(cols)
        for forward in (True, False):
            tag = "f" if forward else "i"
            nat = plan.transform(t, forward)  # natural in, natural out: three exchanges, eight passes
            assert plan.passes == 8
            np.save(os.path.join(out_dir, f"nat_{tag}_{rank}.npy"), nat.numpy().copy())
            tr = plan.transform(t, forward, output_layout="transposed")  # two exchanges
            np.save(os.path.join(out_dir, f"tr_{tag}_{rank}.npy"), tr.numpy().copy())
            one = plan.transform(tin, forward, input_layout="transposed", output_layout="transposed")  # one exchange
            assert plan.passes == 5
            np.save(os.path.join(out_dir, f"one_{tag}_{rank}.npy"), one.numpy().copy())
        assert torch.equal(t, to_tensor(x[rank * per:(rank + 1) * per]))  # the input is never modified
        np.save(os.path.join(out_dir, f"kmap_{rank}.npy"), plan.transposed_output_index())
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,n", ((2, 1 << 10), (4, 1 << 11), (4, 1 << 12)))
def test_distributed_fourstep_matches_oracle(tmp_path, oracle_mod, world, n):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    x = synth.uniform_complex(1, n, seed=5)[0].astype(np.complex128)
    for forward, tag in ((True, "f"), (False, "i")):
        ref = oracle_mod.fft(x, forward)
        nat = np.concatenate([_c(np.load(tmp_path / f"nat_{tag}_{r}.npy")).reshape(-1) for r in range(world)])
        assert oracle_mod.rel_l2(nat, ref) < 1e-13
        # transposed output: element e of rank r's result is X[kmap_r[e]]
        for name in ("tr", "one"):
            for r in range(world):
                z = _c(np.load(tmp_path / f"{name}_{tag}_{r}.npy"))
                kmap = np.load(tmp_path / f"kmap_{r}.npy")
                assert z.shape == kmap.shape
                assert oracle_mod.rel_l2(z, ref[kmap]) < 1e-13, (name, r)
    # every output index appears exactly once over the ranks
    allk = np.concatenate([np.load(tmp_path / f"kmap_{r}.npy").reshape(-1) for r in range(world)])
    assert np.array_equal(np.sort(allk), np.arange(n))


@pytest.mark.parametrize("world,n", ((1, 1 << 8), (2, 1 << 10), (8, 1 << 14)))
def test_virtual_ranks_match_oracle(oracle_mod, world, n):
    """The same algorithm with all ranks stepped in one process (run_virtual), including eight ranks and the
    single-rank degenerate case (no exchange, no radix pass)."""
    x = synth.uniform_complex(1, n, seed=9)[0].astype(np.complex128)
    plans = [DistributedFFT(n, ops=NumpyOps(), rank=r, world=world) for r in range(world)]
    per = n // world
    shards = [to_tensor(x[r * per:(r + 1) * per]) for r in range(world)]
    for forward in (True, False):
        ref = oracle_mod.fft(x, forward)
        res = run_virtual(plans, shards, forward)
        nat = np.concatenate([_c(t.numpy()).reshape(-1) for t in res])
        assert oracle_mod.rel_l2(nat, ref) < 1e-12
        assert plans[0].passes == (8 if world > 1 else 5)
        res = run_virtual(plans, shards, forward, output_layout="transposed")
        for r in range(world):
            assert oracle_mod.rel_l2(_c(res[r].numpy()), ref[plans[r].transposed_output_index()]) < 1e-12


def test_split_n():
    assert split_n(1 << 30, 8) == (1 << 15, 1 << 15)
    assert split_n(1 << 11, 4) == (32, 64)
    with pytest.raises(ValueError):
        split_n(3 << 10, 2)
    with pytest.raises(ValueError):
        split_n(64, 16)
    with pytest.raises(ValueError):
        split_n(1 << 8, 4)  # 16 x 16: no even chunk per peer
