"""CPU tests of the distributed four-step orchestration with gloo ranks: the all-to-all
transposes, twiddle indexing and layouts, with the fp64 oracle standing in for the local HIP
transforms (there is no GPU in this container; the GPU test runs the same code single-rank)."""
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
from fft_amd.distributed import DistributedFFT, split_n  # noqa: E402


class OracleOps:
    """fp64 local numerics for the CPU tests."""

    def fft_rows(self, t, forward):
        import oracle

        a = t.numpy().astype(np.float64)
        y = oracle.fft(a[..., 0] + 1j * a[..., 1], forward)
        return torch.from_numpy(np.stack([y.real, y.imag], axis=-1))

    def twiddle(self, t, row0, n_total, forward):
        rows, cols = t.shape[0], t.shape[1]
        q = (row0 + np.arange(rows, dtype=np.int64))[:, None] * np.arange(cols, dtype=np.int64)[None, :]
        w = np.exp((-2j if forward else 2j) * np.pi * (q % n_total) / n_total)
        a = t.numpy()
        y = (a[..., 0] + 1j * a[..., 1]) * w
        return torch.from_numpy(np.stack([y.real, y.imag], axis=-1))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n, out_dir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        x = synth.uniform_complex(1, n, seed=5)[0].astype(np.complex128)
        plan = DistributedFFT(n, ops=OracleOps())
        per = n // world
        shard = x[rank * per:(rank + 1) * per]
        t = torch.from_numpy(np.stack([shard.real, shard.imag], axis=-1))
        for forward in (True, False):
            nat = plan.transform(t, forward)  # natural in, natural out: three exchanges
            tr = plan.transform(t, forward, output_layout="transposed")  # two exchanges
            # transposed input: columns n2 of my block, all n1
            n1, n2 = plan.n1, plan.n2
            cols = x.reshape(n1, n2)[:, rank * (n2 // world):(rank + 1) * (n2 // world)].T.copy()
            tin = torch.from_numpy(np.stack([cols.real, cols.imag], axis=-1))
            one = plan.transform(tin, forward, input_layout="transposed", output_layout="transposed")  # one exchange
            tag = "f" if forward else "i"
            np.save(os.path.join(out_dir, f"nat_{tag}_{rank}.npy"), nat.numpy())
            np.save(os.path.join(out_dir, f"tr_{tag}_{rank}.npy"), tr.numpy())
            np.save(os.path.join(out_dir, f"one_{tag}_{rank}.npy"), one.numpy())
    finally:
        dist.destroy_process_group()


def _c(a):
    return a[..., 0] + 1j * a[..., 1]


@pytest.mark.parametrize("world,n", ((2, 1 << 10), (4, 1 << 11)))
def test_distributed_fourstep_matches_oracle(tmp_path, oracle_mod, world, n):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, n, str(tmp_path)), nprocs=world, join=True)
    x = synth.uniform_complex(1, n, seed=5)[0].astype(np.complex128)
    n1, n2 = split_n(n, world)
    for forward, tag in ((True, "f"), (False, "i")):
        ref = oracle_mod.fft(x, forward)
        nat = np.concatenate([_c(np.load(tmp_path / f"nat_{tag}_{r}.npy")).reshape(-1) for r in range(world)])
        assert oracle_mod.rel_l2(nat, ref) < 1e-13
        # transposed output: rank r holds Z[k1][k2] = X[k1 + n1*k2] for k1 in its block
        refz = ref.reshape(n2, n1).T  # [k1][k2]
        for name in ("tr", "one"):
            z = np.concatenate([_c(np.load(tmp_path / f"{name}_{tag}_{r}.npy")) for r in range(world)], axis=0)
            assert z.shape == (n1, n2)
            assert oracle_mod.rel_l2(z, refz) < 1e-13, name


def test_split_n():
    assert split_n(1 << 30, 8) == (1 << 15, 1 << 15)
    assert split_n(1 << 11, 4) == (32, 64)
    with pytest.raises(ValueError):
        split_n(3 << 10, 2)
    with pytest.raises(ValueError):
        split_n(64, 16)
