"""The distributed transform with REAL ranks on device tensors: separate processes under torch.distributed.run, HIP
kernels for the local passes, torch.distributed.all_to_all_single on the device buffers for the exchanges
(tests/dist_rank_worker.py). A box has one GPU, so every rank uses cuda:0: gloo carries the device chunks between two
and four processes (through host memory); RCCL accepts one rank per GPU and runs with a world of one (communicator set-up, the
all-reduce / barrier calls bench.py makes, and all_to_all_single on device memory through RCCL itself). What stays
unexercised is RCCL's transport between different GPUs."""
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from fft_amd import synth  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _c(a):
    a = np.asarray(a, dtype=np.float64)
    return a[..., 0] + 1j * a[..., 1]


@pytest.mark.parametrize("backend,world,lg", (("gloo", 2, 18), ("gloo", 4, 20), ("nccl", 1, 20)))
def test_transform_over_a_process_group(tmp_path, oracle_mod, backend, world, lg):
    n = 1 << lg
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()),
           os.path.join(HERE, "dist_rank_worker.py"), backend, str(lg), str(tmp_path)]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    run = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=110)
    assert run.returncode == 0, run.stdout[-4000:]
    x = synth.uniform_complex(1, n, seed=21)[0].astype(np.complex128)
    for forward, tag in ((True, "f"), (False, "i")):
        ref = oracle_mod.fft(x, forward)
        nat = np.concatenate([_c(np.load(tmp_path / f"nat_{tag}_{r}.npy")).reshape(-1) for r in range(world)])
        err = oracle_mod.rel_l2(nat, ref)
        assert err <= 2e-6, (backend, world, tag, err)
        for name in ("tr", "one"):
            for r in range(world):
                z = _c(np.load(tmp_path / f"{name}_{tag}_{r}.npy"))
                kmap = np.load(tmp_path / f"kmap_{r}.npy")
                assert z.shape == kmap.shape
                err = oracle_mod.rel_l2(z, ref[kmap])
                assert err <= 2e-6, (backend, world, name, tag, r, err)
    allk = np.concatenate([np.load(tmp_path / f"kmap_{r}.npy").reshape(-1) for r in range(world)])
    assert np.array_equal(np.sort(allk), np.arange(n))
