"""Randomised parity sweep: lengths, batches (whole and ragged chunks), storage types, directions, in place or not,
and per-call launch options (alternative plan, resident grid, XCD run length) drawn from a seeded generator, every
case against the fp64 oracle. It exists for the corners the hand-written cases miss (a partial last chunk met by a
workgroup on its second trip, a one-transform batch with a resident grid, ...)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402

import fft_amd  # noqa: E402
from fft_amd import synth  # noqa: E402

TOL = {"fp32": 2e-6, "fp16": 1e-2, "bf16": 1e-2}
DT = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}


def one_case(oracle_mod, rng, case):
    kind = rng.choice(["fp32", "fp32", "fp16", "bf16"])
    n = int(rng.choice(fft_amd.SUPPORTED_N))
    per_chunk = max(1, 4096 // n)
    shape = rng.integers(0, 4)
    if shape == 0:
        batch = int(rng.integers(1, 2 * per_chunk + 2))
    elif shape == 1:
        batch = int(per_chunk * rng.integers(1, 40))
    else:
        batch = int(per_chunk * rng.integers(1, 600) + rng.integers(0, per_chunk + 1))
    batch = max(1, min(batch, (1 << 22) // n))
    forward = bool(rng.integers(0, 2))
    inplace = bool(rng.integers(0, 2))
    options = {}
    if rng.integers(0, 2):
        options["plan"] = 3
    grid = int(rng.choice([0, 0, 1, 2]))
    if grid:
        options["grid_blocks_per_cu"] = grid
    elif rng.integers(0, 4) == 0:
        options["grid_blocks"] = int(rng.integers(1, 9))
    xcd = int(rng.choice([-1, -1, 0, 2, 5]))
    if xcd >= 0:
        options["xcd_run_log2"] = xcd
    scale = 1.0 / np.sqrt(n) if kind == "fp16" else 1.0
    x = (synth.uniform_complex(batch, n, seed=1000 + case) * scale).astype(np.complex64)
    d_in = torch.from_numpy(x.view(np.float32).reshape(batch, n, 2)).to("cuda").to(DT[kind]).contiguous()
    stored = d_in.float().cpu().numpy().astype(np.float64)
    d_out = d_in if inplace else torch.full_like(d_in, float("nan"))
    fft_amd.ThunderFFTInitialize(kind, n)
    tag = f"case {case}: {kind} n={n} batch={batch} forward={forward} inplace={inplace} options={options}"
    try:
        fft_amd.ThunderFFT_global(d_in, d_out, batch, n, forward, options=options or None)
    except fft_amd.ThunderFFTError as e:
        # a length without that alternative plan, or a resident grid asked of a kernel that runs one workgroup per
        # transform (fp32 n = 8192, 16-bit n >= 8192 on their default plans)
        grid_asked = "grid_blocks_per_cu" in options or "grid_blocks" in options
        assert e.code == 1 and ("plan" in options or (grid_asked and n >= 8192)), tag
        return
    torch.cuda.synchronize()
    got = d_out.float().cpu().numpy().astype(np.float64)
    got = got[..., 0] + 1j * got[..., 1]
    ref = oracle_mod.fft(stored[..., 0] + 1j * stored[..., 1], forward)
    assert np.all(np.isfinite(got.view(np.float64))), tag
    assert oracle_mod.rel_l2(got, ref) <= TOL[kind], tag
    worst = max(oracle_mod.rel_l2(got[b], ref[b]) for b in {0, batch // 2, batch - 1, max(0, batch - 2)})
    assert worst <= 2 * TOL[kind], tag


@pytest.mark.parametrize("seed", [1, 2, 3, 4])
def test_randomised_parity(oracle_mod, seed):
    rng = np.random.default_rng(seed)
    for k in range(60):
        one_case(oracle_mod, rng, 100 * seed + k)
