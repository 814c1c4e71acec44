"""GPU parity tests: the HIP path, called through the C ABI, against the fp64 CPU oracle.

Tolerances come from the build's accuracy gate: relative L2 error vs the fp64 DFT of the
stored inputs <= 1e-5 for fp32, <= 1e-2 for fp16 / bf16 storage (the output rounding of a
16-bit format alone is ~3e-4 / ~2e-3). The reference's own fp32 kernels multiply in TF32
(thunderfft/include/thunderfft/detail/kernel/unit_kernel_fp32.cuh:123) and are not the target.
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402

import fft_amd  # noqa: E402
from fft_amd import synth  # noqa: E402

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fft_golden.npz")
ALL_N = fft_amd.SUPPORTED_N
TOL = {"fp32": 1e-5, "fp16": 1e-2, "bf16": 1e-2}
TORCH_DT = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}


def to_device(x, kind):
    """complex64 numpy [batch, n] -> device tensor [batch, n, 2] of the storage type."""
    t = torch.from_numpy(np.ascontiguousarray(x).view(np.float32).reshape(x.shape + (2,)))
    return t.to("cuda").to(TORCH_DT[kind]).contiguous()


def to_host(t):
    """device tensor [batch, n, 2] -> complex128 numpy [batch, n]."""
    a = t.float().cpu().numpy().astype(np.float64)
    return a[..., 0] + 1j * a[..., 1]


def run(x, kind, forward=True, inplace=False, options=None):
    n = x.shape[-1]
    d_in = to_device(x, kind)
    d_out = d_in if inplace else torch.full_like(d_in, float("nan"))
    fft_amd.ThunderFFTInitialize(kind, n)
    fft_amd.ThunderFFT_global(d_in, d_out, x.shape[0], n, forward, options=options)
    torch.cuda.synchronize()
    return to_host(d_out), to_host(to_device(x, kind))


@pytest.mark.parametrize("n", ALL_N)
@pytest.mark.parametrize("forward", (True, False))
def test_fp32_vs_oracle(oracle_mod, n, forward):
    batch = max(3, 3 * 4096 // n + 1)  # not a multiple of the per-workgroup chunk: exercises the tail
    x = synth.uniform_complex(batch, n, seed=n + int(forward))
    got, stored = run(x, "fp32", forward)
    ref = oracle_mod.fft(stored, forward)
    err = oracle_mod.rel_l2(got, ref)
    assert err <= TOL["fp32"], f"n={n} forward={forward} rel_l2={err:.3e}"
    assert err <= 2e-6, f"fp32 accuracy regressed: n={n} rel_l2={err:.3e}"


@pytest.mark.parametrize("n", ALL_N)
@pytest.mark.parametrize("kind", ("fp16", "bf16"))
def test_half_storage_vs_oracle(oracle_mod, n, kind):
    batch = 2 * 4096 // n + 3
    # pre-scaled by 1/sqrt(N) like the reference's half-precision bench input (sample/bench/main.cu:36)
    x = synth.uniform_complex(batch, n, seed=7 * n, scale=1.0 / np.sqrt(n))
    for forward in (True, False):
        got, stored = run(x, kind, forward)
        ref = oracle_mod.fft(stored, forward)
        err = oracle_mod.rel_l2(got, ref)
        print(f"{kind} n={n} forward={forward} rel_l2={err:.3e}")
        assert err <= TOL[kind], f"{kind} n={n} forward={forward} rel_l2={err:.3e}"


@pytest.mark.parametrize("n", ALL_N)
def test_golden_vectors(oracle_mod, n):
    g = np.load(GOLDEN)
    x = g[f"x_{n}"]
    got, _ = run(x, "fp32", True)
    assert oracle_mod.rel_l2(got, g[f"fwd_{n}"]) <= TOL["fp32"]
    got, _ = run(x, "fp32", False)
    assert oracle_mod.rel_l2(got, g[f"inv_{n}"]) <= TOL["fp32"]


@pytest.mark.parametrize("kind", ("fp16", "bf16"))
@pytest.mark.parametrize("n", ALL_N)
def test_golden_vectors_half_storage(oracle_mod, n, kind):
    """The committed fixtures through the 16-bit kernels: inputs rounded to the storage type and scaled by
    1/sqrt(N) (the reference's half-precision input scale, sample/bench/main.cu:36); the expected output is the
    fixture's fp64 transform scaled alike, and the gate covers the input rounding too."""
    g = np.load(GOLDEN)
    scale = 1.0 / np.sqrt(n)
    x = (g[f"x_{n}"] * scale).astype(np.complex64)
    got, _ = run(x, kind, True)
    assert oracle_mod.rel_l2(got, g[f"fwd_{n}"] * scale) <= TOL[kind]
    got, _ = run(x, kind, False)
    assert oracle_mod.rel_l2(got, g[f"inv_{n}"] * scale) <= TOL[kind]


@pytest.mark.parametrize("n", (64, 256, 1024, 4096, 16384))
def test_in_place_equals_out_of_place(n):
    batch = 40000 // n + 5
    x = synth.uniform_complex(batch, n, seed=11)
    a, _ = run(x, "fp32", True, inplace=False)
    b, _ = run(x, "fp32", True, inplace=True)
    assert np.array_equal(a, b)


@pytest.mark.parametrize("batch", (1, 2, 15, 16, 17, 63, 64, 65, 257))
def test_ragged_batches_small_n(oracle_mod, batch):
    # N=64 packs 64 transforms into one workgroup chunk: every tail shape must be exact
    for n in (64, 256):
        x = synth.uniform_complex(batch, n, seed=batch)
        got, stored = run(x, "fp32", True)
        assert oracle_mod.rel_l2(got, oracle_mod.fft(stored)) <= TOL["fp32"]


def test_tail_does_not_write_past_the_batch():
    n, batch, extra = 128, 37, 64
    x = synth.uniform_complex(batch + extra, n, seed=5)
    d = to_device(x, "fp32")
    out = torch.full_like(d, 123.0)
    fft_amd.ThunderFFTInitialize("fp32", n)
    fft_amd.ThunderFFT_global(d, out, batch, n, True)
    torch.cuda.synchronize()
    assert torch.all(out[batch:] == 123.0)
    assert not torch.any(out[:batch] == 123.0)


def test_empty_batch_is_a_noop():
    d = torch.zeros(4, 64, 2, device="cuda")
    fft_amd.ThunderFFTInitialize("fp32", 64)
    fft_amd.ThunderFFT_global(d, d, 0, 64, True)
    torch.cuda.synchronize()


@pytest.mark.parametrize("n", ALL_N)
def test_kat_fp32_every_length(n):
    """Known answers from the reference's own inputs, every fp32 length, both directions, a batch that leaves a
    partial workgroup chunk: the bench sine sin(2 pi n / N) / sqrt(N) (sample/bench/main.cu:35-41) has two
    non-zero bins, -+ i sqrt(N) / 2 forward and +- i sqrt(N) / 2 inverse; a unit impulse at p gives
    exp(-+ 2 pi i k p / N); a constant gives the DC bin alone."""
    batch = 2 * max(1, 4096 // n) + 1
    for forward in (True, False):
        sign = -1.0 if forward else 1.0
        got, _ = run(synth.bench_sine(batch, n), "fp32", forward)
        want = np.zeros((batch, n), dtype=np.complex128)
        want[:, 1] = sign * 0.5j * np.sqrt(n)
        want[:, n - 1] = -sign * 0.5j * np.sqrt(n)
        assert np.max(np.abs(got - want)) < 2e-5, (n, forward)
        x = np.zeros((batch, n), dtype=np.complex64)
        pos = [(37 * b + 5) % n for b in range(batch)]
        for b in range(batch):
            x[b, pos[b]] = 1.0
        got, _ = run(x, "fp32", forward)
        k = np.arange(n)
        for b in range(batch):
            assert np.max(np.abs(got[b] - np.exp(sign * 2j * np.pi * k * pos[b] / n))) < 2e-6, (n, forward, b)
        got, _ = run(np.full((batch, n), 0.25 / n, dtype=np.complex64), "fp32", forward)
        want = np.zeros((batch, n), dtype=np.complex128)
        want[:, 0] = 0.25
        assert np.max(np.abs(got - want)) < 1e-6, (n, forward)


@pytest.mark.parametrize("kind", ("fp16", "bf16"))
@pytest.mark.parametrize("n", (128, 256, 1024, 4096, 8192, 16384))
def test_kat_half_storage(kind, n):
    """Known answers for the 16-bit kernels (matrix-core kernels from N = 256): the reference's half-precision
    bench input sin(2 pi n / N) / sqrt(N) (sample/bench/main.cu:35-41) has two non-zero bins, -+ i sqrt(N) / 2; a
    unit impulse at point p gives exp(-2 pi i k p / N); a constant gives the DC bin alone. Every transform of a
    batch that leaves a partial workgroup tile is checked on its own."""
    batch = 2 * max(1, 4096 // n) + 1
    eps = 2.0 ** -10 if kind == "fp16" else 2.0 ** -7
    got, _ = run(synth.bench_sine(batch, n), kind, True)
    want = np.zeros((batch, n), dtype=np.complex128)
    want[:, 1] = -0.5j * np.sqrt(n)
    want[:, n - 1] = 0.5j * np.sqrt(n)
    assert np.max(np.abs(got - want)) <= 4 * eps * np.sqrt(n), (kind, n, np.max(np.abs(got - want)))
    x = np.zeros((batch, n), dtype=np.complex64)
    pos = [(37 * b + 5) % n for b in range(batch)]
    for b in range(batch):
        x[b, pos[b]] = 1.0
    got, _ = run(x, kind, True)
    k = np.arange(n)
    for b in range(batch):
        want_b = np.exp(-2j * np.pi * k * pos[b] / n)
        assert np.max(np.abs(got[b] - want_b)) <= 4 * eps, (kind, n, b)
    x = np.full((batch, n), 0.25 / n, dtype=np.complex64)  # exactly representable; DC bin = 0.25
    got, _ = run(x, kind, True)
    want = np.zeros((batch, n), dtype=np.complex128)
    want[:, 0] = 0.25
    assert np.max(np.abs(got - want)) <= 2 * eps * 0.25, (kind, n)


def test_kat_fft1d_tones():
    # reference fft1d sample: N=64, tones at bins 1+b and 7 (sample/fft1d/main.cu:16-30)
    n, batch = 64, 40
    got, _ = run(synth.fft1d_tones(batch, n), "fp32", True)
    want = np.zeros((batch, n), dtype=np.complex128)
    for b in range(batch):
        want[b, (1 + b) % n] += n * (0.5 + 0.1 * b)
        want[b, 7] += 0.1 * n
    assert np.max(np.abs(got - want)) < 1e-3


def test_kat_convolution_identity():
    # IFFT(FFT(x) . FFT(h/N)) = circular convolution (conv1d_fullfft_l4096_thunder.cu:354-356, 472-490)
    n, k = 4096, 31
    rng = np.random.default_rng(1)
    x = rng.standard_normal(n).astype(np.float32)
    h = np.zeros(n, dtype=np.float32)
    h[:k] = rng.standard_normal(k)
    X, _ = run(x.astype(np.complex64)[None], "fp32", True)
    H, _ = run((h / n).astype(np.complex64)[None], "fp32", True)
    y, _ = run((X * H).astype(np.complex64), "fp32", False)
    direct = np.array([np.dot(h[:k].astype(np.float64), x[(i - np.arange(k)) % n].astype(np.float64)) for i in range(n)])
    assert np.max(np.abs(y[0] - direct)) < 1e-4


def _spot_check_vs_oracle(oracle_mod, x, X, n, idx, tol):
    xs = x[idx].float().cpu().numpy().astype(np.float64)
    Xs = X[idx].float().cpu().numpy().astype(np.float64)
    ref = oracle_mod.fft(xs[..., 0] + 1j * xs[..., 1], True)
    err = max(oracle_mod.rel_l2(Xs[b, :, 0] + 1j * Xs[b, :, 1], ref[b]) for b in range(len(idx)))
    assert err <= tol, err


@pytest.mark.parametrize("n", ALL_N[2:])
def test_full_size_properties(oracle_mod, n):
    """Size-independent checks at the sweep configuration's full size (2^28 points per launch, N = 256 ... 16384):
    inverse(forward(x)) = N x, Parseval per transform, and 8 scattered transforms against the fp64 oracle."""
    batch = (1 << 28) // n
    g = torch.Generator(device="cuda").manual_seed(n)
    x = torch.rand(batch, n, 2, device="cuda", generator=g) * 2 - 1
    fft_amd.ThunderFFTInitialize("fp32", n)
    X = torch.empty_like(x)
    fft_amd.ThunderFFT_global(x, X, batch, n, True)
    y = torch.empty_like(x)
    fft_amd.ThunderFFT_global(X, y, batch, n, False)
    torch.cuda.synchronize()
    y.div_(n).sub_(x)
    rt = (torch.linalg.vector_norm(y) / torch.linalg.vector_norm(x)).item()
    assert rt < 2e-6, rt
    del y
    ex = (x * x).sum(dim=(1, 2), dtype=torch.float64)
    eX = (X * X).sum(dim=(1, 2), dtype=torch.float64) / n
    assert torch.max(torch.abs(eX / ex - 1)).item() < 1e-5
    idx = [0, 1, batch // 3, batch // 2, batch - 2, batch - 1, 12345 % batch, 777 % batch]
    _spot_check_vs_oracle(oracle_mod, x, X, n, idx, 2e-6)
    del x, X
    torch.cuda.empty_cache()


@pytest.mark.parametrize("kind", ("fp16", "bf16"))
def test_full_size_16bit_configuration(oracle_mod, kind):
    """The 16-bit configuration at full size, N = 4096 x 65536 in place: round trip, Parseval per transform and
    8 scattered transforms against the fp64 oracle of the stored inputs."""
    n, batch = 4096, 65536
    dt = TORCH_DT[kind]
    g = torch.Generator(device="cuda").manual_seed(7)
    x = ((torch.rand(batch, n, 2, device="cuda", generator=g) * 2 - 1) / n ** 0.5).to(dt)
    fft_amd.ThunderFFTInitialize(kind, n)
    X = x.clone()
    fft_amd.ThunderFFT_global(X, X, batch, n, True)
    y = X.clone()
    fft_amd.ThunderFFT_global(y, y, batch, n, False)
    torch.cuda.synchronize()
    tol = TOL[kind]
    xf = x.float()
    rt = (torch.linalg.vector_norm(y.float() / n - xf) / torch.linalg.vector_norm(xf)).item()
    assert rt < tol, rt
    del y
    ex = (xf * xf).sum(dim=(1, 2), dtype=torch.float64)
    Xf = X.float()
    eX = (Xf * Xf).sum(dim=(1, 2), dtype=torch.float64) / n
    assert torch.max(torch.abs(eX / ex - 1)).item() < 2 * tol
    del xf, Xf
    idx = [0, 1, batch // 3, batch // 2, batch - 2, batch - 1, 12345, 777]
    _spot_check_vs_oracle(oracle_mod, x, X, n, idx, tol)
    del x, X
    torch.cuda.empty_cache()


@pytest.mark.parametrize("kind,n", (("fp32", 4096), ("fp32", 8192), ("fp32", 64), ("fp16", 1024), ("bf16", 16384)))
def test_more_than_2_to_the_32_points(oracle_mod, kind, n):
    """Offsets beyond 2^31 (and 2^32) points: one in-place launch over a little more than 2^32 points (32 GiB of
    fp32, 16 GiB of 16-bit data). Transforms at the start, around the 2^31- and 2^32-point marks and at the very
    end are compared with the oracle; everything in between must at least have been transformed (Parseval on a
    strided sample)."""
    batch = (1 << 32) // n + 3
    dt = TORCH_DT[kind]
    x = torch.empty(batch, n, 2, device="cuda", dtype=dt)
    g = torch.Generator(device="cuda").manual_seed(n)
    step = 1 << 16
    for b0 in range(0, batch, step):  # filled in slices: no second 32 GiB temporary
        b1 = min(batch, b0 + step)
        x[b0:b1] = ((torch.rand(b1 - b0, n, 2, device="cuda", generator=g) * 2 - 1) / n ** 0.5).to(dt)
    half, full = (1 << 31) // n, (1 << 32) // n
    idx = sorted({0, 1, half - 1, half, half + 1, full - 1, full, full + 1, batch - 1})
    sample = list(range(7, batch, max(1, batch // 257)))
    keep_in = x[idx].clone()
    e_in = (x[sample].float() ** 2).sum(dim=(1, 2), dtype=torch.float64)
    fft_amd.ThunderFFTInitialize(kind, n)
    fft_amd.ThunderFFT_global(x, x, batch, n, True)
    torch.cuda.synchronize()
    tol = TOL[kind]
    xs = keep_in.float().cpu().numpy().astype(np.float64)
    Xs = x[idx].float().cpu().numpy().astype(np.float64)
    ref = oracle_mod.fft(xs[..., 0] + 1j * xs[..., 1], True)
    for k in range(len(idx)):
        assert oracle_mod.rel_l2(Xs[k, :, 0] + 1j * Xs[k, :, 1], ref[k]) <= tol, (kind, n, idx[k])
    e_out = (x[sample].float() ** 2).sum(dim=(1, 2), dtype=torch.float64) / n
    assert torch.max(torch.abs(e_out / e_in - 1)).item() < (1e-5 if kind == "fp32" else 2 * tol)
    del x
    torch.cuda.empty_cache()


@pytest.mark.parametrize("mode", [3, 4])
@pytest.mark.parametrize("n", ALL_N)
def test_alternative_plans_vs_oracle(oracle_mod, n, mode):
    """Plans 3 and 4 select the A/B kernels of a length (another factorisation, workgroup size or LDS
    exchange form); a small resident grid makes workgroups loop over chunks."""
    batch = 3 * 256 * max(1, 4096 // n) + 5
    x = synth.uniform_complex(batch, n, seed=n + 99)
    for forward in (True, False):
        try:
            got, stored = run(x, "fp32", forward, options={"plan": mode, "grid_blocks_per_cu": 1})
        except fft_amd.ThunderFFTError:
            assert mode == 4, "every length has a first alternative plan"
            pytest.skip(f"no second alternative plan for n={n}")
        err = oracle_mod.rel_l2(got, oracle_mod.fft(stored, forward))
        assert err <= 2e-6, f"n={n} forward={forward} rel_l2={err:.3e}"


@pytest.mark.parametrize("mode", [0, 3, 5, 6, 7, 9, 11, 12])
@pytest.mark.parametrize("forward", (True, False))
def test_fp16_4096_plans_vs_oracle(oracle_mod, mode, forward):
    """fp16 storage, N = 4096. Mode 0 (default), 5, 6, 7, 11, 12: all three passes on the matrix cores (fp16
    operands, fp32 accumulation, data rounded to fp16 between passes), one workgroup per transform (0 and 12
    with the pass-1 twiddles built incrementally, 5 and 6 from a table) or as a resident prefetching grid
    (7, 11); mode 3: the fp32 VALU kernel; mode 9: that kernel as a resident prefetching grid. Same gate for all,
    and a tighter regression bound."""
    n = 4096
    batch = 2500 if mode in (7, 9, 11) else 37  # resident grids: several transforms per workgroup, uneven tail
    x = synth.uniform_complex(batch, n, seed=mode) / np.sqrt(n)  # the reference's input scale, sample/bench/main.cu:36
    got, stored = run(x.astype(np.complex64), "fp16", forward,
                      options={"plan": mode, "grid_blocks_per_cu": 2 if mode == 7 else 0})
    ref = oracle_mod.fft(stored, forward)
    err = oracle_mod.rel_l2(got, ref)
    print(f"mode={mode} forward={forward} rel_l2={err:.3e}")
    assert err <= TOL["fp16"], f"mode={mode} forward={forward} rel_l2={err:.3e}"
    assert err <= 3e-3, f"accuracy regressed: mode={mode} rel_l2={err:.3e}"
    # every transform on its own: a misplaced row or column would hide in the aggregate
    worst = max(oracle_mod.rel_l2(got[b], ref[b]) for b in range(batch))
    assert worst <= TOL["fp16"], worst


@pytest.mark.parametrize("mode", [0, 3, 6, 8, 11])
@pytest.mark.parametrize("forward", (True, False))
def test_bf16_4096_plans_vs_oracle(oracle_mod, mode, forward):
    """bf16 storage, N = 4096. Mode 0 (default, one workgroup per transform) and 11 (resident grid): matrix
    cores with bf16 operands; modes 6 and 8: the DFT matrix split into bf16 high and low parts; mode 3: the
    fp32 VALU kernel. Unscaled inputs: bf16 intermediates keep the fp32 range."""
    n = 4096
    batch = 2500 if mode in (11, 8) else 37
    x = synth.uniform_complex(batch, n, seed=40 + mode)
    got, stored = run(x, "bf16", forward, options={"plan": mode})
    ref = oracle_mod.fft(stored, forward)
    err = oracle_mod.rel_l2(got, ref)
    print(f"bf16 mode={mode} forward={forward} rel_l2={err:.3e}")
    assert err <= TOL["bf16"], f"mode={mode} forward={forward} rel_l2={err:.3e}"
    worst = max(oracle_mod.rel_l2(got[b], ref[b]) for b in range(batch))
    assert worst <= TOL["bf16"], worst


@pytest.mark.parametrize("kind", ("fp16", "bf16"))
@pytest.mark.parametrize("n", (256, 512, 1024, 2048))
@pytest.mark.parametrize("mode", (0, 3, 4, 6, 10, 11, 12))
def test_half_storage_matrix_core_lengths(oracle_mod, n, kind, mode):
    """16-bit storage, N = 256 ... 2048: modes 0 (default), 6, 10, 11, 12 run on the matrix cores (16 / (N / 256)
    transforms per workgroup; 11 a resident grid, the others one workgroup per tile with the pass-1 twiddles
    from a table or built incrementally and one or two LDS images), modes 3 and 4 are the fp32 VALU
    kernels. Batches that leave a partial tile and batches of several tiles per workgroup, per-transform
    errors (a misplaced row would hide in the aggregate), both directions, out of place and in place."""
    per_tile = 4096 // n
    for batch in (1, per_tile + 1, 3 * 768 * per_tile + per_tile // 2 + 1):
        scale = 1.0 / np.sqrt(n) if kind == "fp16" else 1.0
        x = (synth.uniform_complex(batch, n, seed=n + batch) * scale).astype(np.complex64)
        for forward in (True, False):
            for inplace in (False, True):
                got, stored = run(x, kind, forward, inplace=inplace, options={"plan": mode})
                ref = oracle_mod.fft(stored, forward)
                err = oracle_mod.rel_l2(got, ref)
                assert err <= TOL[kind], f"{kind} n={n} mode={mode} batch={batch} forward={forward} rel_l2={err:.3e}"
                step = max(1, batch // 97)
                worst = max(oracle_mod.rel_l2(got[b], ref[b]) for b in list(range(0, batch, step)) + [batch - 1])
                assert worst <= TOL[kind], (n, mode, batch, forward, worst)


@pytest.mark.parametrize("kind", ("fp16", "bf16"))
@pytest.mark.parametrize("n,mode", ((64, 0), (64, 12), (64, 3), (64, 4), (128, 0), (128, 12), (128, 3), (128, 4)))
def test_half_storage_matrix_core_short_lengths(oracle_mod, n, kind, mode):
    """16-bit storage, N = 64 and 128 on the matrix cores (modes 0 and 12): the N = 256
    kernel with a block-diagonal second pass, 4096 / N transforms per workgroup and 256 / N per 256-point row. The
    other modes are the VALU kernels. Batches that leave the last row and the last tile partly filled, buffers that
    end right behind the batch (a read past it would fault or show), per-transform errors, both directions, out of
    place and in place, and nothing written behind the batch."""
    per_row, per_tile = 256 // n, 4096 // n
    for batch in (1, per_row + 1, per_tile - 1, per_tile + per_row + 1, 5 * 768 * per_tile + per_tile // 2 + 1):
        scale = 1.0 / np.sqrt(n) if kind == "fp16" else 1.0
        x = (synth.uniform_complex(batch, n, seed=n + batch) * scale).astype(np.complex64)
        for forward in (True, False):
            for inplace in (False, True):
                got, stored = run(x, kind, forward, inplace=inplace, options={"plan": mode} if mode else None)
                ref = oracle_mod.fft(stored, forward)
                err = oracle_mod.rel_l2(got, ref)
                assert err <= TOL[kind], f"{kind} n={n} mode={mode} batch={batch} forward={forward} rel_l2={err:.3e}"
                step = max(1, batch // 97)
                worst = max(oracle_mod.rel_l2(got[b], ref[b]) for b in list(range(0, batch, step)) + [batch - 1])
                assert worst <= TOL[kind], (n, mode, batch, forward, worst)
    batch, extra = per_tile + per_row + 1, 5
    d = to_device((synth.uniform_complex(batch + extra, n, seed=3) / np.sqrt(n)).astype(np.complex64), kind)
    out = torch.full_like(d, 123.0)
    fft_amd.ThunderFFTInitialize(kind, n)
    fft_amd.ThunderFFT_global(d, out, batch, n, True, options={"plan": mode} if mode else None)
    torch.cuda.synchronize()
    assert torch.all(out[batch:] == 123.0)
    assert not torch.any(out[:batch] == 123.0)


@pytest.mark.parametrize("kind", ("fp16", "bf16"))
@pytest.mark.parametrize("n", (8192, 16384))
@pytest.mark.parametrize("mode", (0, 3, 4))
def test_half_storage_matrix_core_long_lengths(oracle_mod, n, kind, mode):
    """16-bit storage, N = 8192 and 16384: mode 0 (default) splits the transform into 2 or 4 interleaved
    4096-point transforms on the matrix cores and combine them in registers; modes 3 and 4 are the fp32 VALU
    kernels. Per-transform errors, both directions, out of place and in place."""
    for batch in (1, 5, 531):
        scale = 1.0 / np.sqrt(n) if kind == "fp16" else 1.0
        x = (synth.uniform_complex(batch, n, seed=n + batch) * scale).astype(np.complex64)
        for forward in (True, False):
            for inplace in (False, True):
                got, stored = run(x, kind, forward, inplace=inplace, options={"plan": mode})
                ref = oracle_mod.fft(stored, forward)
                err = oracle_mod.rel_l2(got, ref)
                assert err <= TOL[kind], f"{kind} n={n} mode={mode} batch={batch} forward={forward} rel_l2={err:.3e}"
                step = max(1, batch // 61)
                worst = max(oracle_mod.rel_l2(got[b], ref[b]) for b in list(range(0, batch, step)) + [batch - 1])
                assert worst <= TOL[kind], (n, mode, batch, forward, worst)


def test_matrix_core_tail_does_not_write_past_the_batch():
    n, batch, extra = 512, 13, 11  # 8 transforms per tile: the last tile is partial
    x = synth.uniform_complex(batch + extra, n, seed=5) / np.sqrt(n)
    d = to_device(x.astype(np.complex64), "fp16")
    out = torch.full_like(d, 123.0)
    fft_amd.ThunderFFTInitialize("fp16", n)
    fft_amd.ThunderFFT_global(d, out, batch, n, True)
    torch.cuda.synchronize()
    assert torch.all(out[batch:] == 123.0)
    assert not torch.any(out[:batch] == 123.0)


def test_fp16_4096_in_place_and_small_batches(oracle_mod):
    # the default fp16 kernel is a resident grid: in place, and batches smaller than the grid
    n = 4096
    for batch in (1, 3, 700, 1000):
        x = (synth.uniform_complex(batch, n, seed=batch) / np.sqrt(n)).astype(np.complex64)
        got, stored = run(x, "fp16", True, inplace=True)
        assert oracle_mod.rel_l2(got, oracle_mod.fft(stored, True)) <= 3e-3, batch


def test_resident_grid_matches_one_shot():
    # the same kernel walked by a resident grid (each workgroup loops over chunks) is bit-identical
    n, batch = 1024, 5000
    x = synth.uniform_complex(batch, n, seed=21)
    a, _ = run(x, "fp32", True)
    b, _ = run(x, "fp32", True, options={"grid_blocks_per_cu": 1})
    assert np.array_equal(a, b)


@pytest.mark.parametrize("kind", ["fp32", "fp16", "bf16"])
@pytest.mark.parametrize("n", [64, 512, 4096, 16384, 1 << 16])
def test_canary_and_determinism(n, kind):
    """Out-of-bounds and race check without a device sanitizer: guard bands on both sides of the input
    and output stay untouched, and two runs give the same bits."""
    if kind != "fp32" and n > 16384:
        pytest.skip("multi-pass lengths are fp32 only")
    batch = 2 * max(1, 8192 // n) + 1
    guard = 4096
    dt = TORCH_DT[kind]
    x = synth.uniform_complex(batch, n, seed=n) / np.sqrt(n)
    d_x = to_device(x.astype(np.complex64), kind)
    buf_in = torch.full((guard + batch * n * 2 + guard,), 7.0, device="cuda", dtype=dt)
    buf_out = torch.full((guard + batch * n * 2 + guard,), -3.0, device="cuda", dtype=dt)
    buf_in[guard:guard + batch * n * 2] = d_x.reshape(-1)
    v_in = buf_in[guard:guard + batch * n * 2].view(batch, n, 2)
    v_out = buf_out[guard:guard + batch * n * 2].view(batch, n, 2)
    fft_amd.ThunderFFTInitialize(kind, n)
    fft_amd.ThunderFFT_global(v_in, v_out, batch, n, True)
    torch.cuda.synchronize()
    first = v_out.clone()
    fft_amd.ThunderFFT_global(v_in, v_out, batch, n, True)
    torch.cuda.synchronize()
    assert torch.equal(first.view(torch.int16 if kind != "fp32" else torch.int32),
                       v_out.view(torch.int16 if kind != "fp32" else torch.int32))
    assert torch.all(buf_out[:guard] == -3.0) and torch.all(buf_out[-guard:] == -3.0)
    assert torch.all(buf_in[:guard] == 7.0) and torch.all(buf_in[-guard:] == 7.0)
    assert torch.equal(v_in, d_x)
    assert torch.all(torch.isfinite(v_out.float()))


@pytest.mark.parametrize("kind,n", [("fp32", 4096), ("fp32", 1024), ("fp16", 4096), ("bf16", 4096)])
def test_hip_graph_capture(kind, n):
    """tfft_exec_c2c for one-kernel lengths makes no allocation and no synchronisation, so it can be
    captured into a HIP graph and replayed."""
    batch = 300
    x = (synth.uniform_complex(batch, n, seed=3) / np.sqrt(n)).astype(np.complex64)
    d_in = to_device(x, kind)
    eager = torch.empty_like(d_in)
    fft_amd.ThunderFFTInitialize(kind, n)
    fft_amd.ThunderFFT_global(d_in, eager, batch, n, True)
    torch.cuda.synchronize()
    out = torch.zeros_like(d_in)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        fft_amd.ThunderFFT_global(d_in, out, batch, n, True)  # on the capture stream (torch's current stream)
    for _ in range(2):
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eager)


@pytest.mark.parametrize("kind,n", [("fp32", 64), ("fp32", 1024), ("fp32", 4096), ("fp16", 256), ("fp16", 2048),
                                    ("bf16", 4096), ("fp16", 16384), ("fp32", 1 << 16)])
def test_xcd_chunk_order_does_not_change_results(kind, n):
    """The order in which workgroups take chunks (runs of 2^L consecutive chunks per XCD) is a permutation of the
    batch: every L gives bit-identical output, for batches that do not fill the last group of runs, for one-shot
    and resident grids."""
    per_chunk = max(1, 4096 // n)
    batch = (1 if n > 16384 else 777) * per_chunk + per_chunk // 2 + 1
    scale = 1.0 / np.sqrt(n) if kind == "fp16" else 1.0
    x = (synth.uniform_complex(batch, n, seed=3) * scale).astype(np.complex64)
    want, _ = run(x, kind, True, options={"xcd_run_log2": 0})
    # one workgroup per transform (fp32 n = 8192, 16-bit n >= 8192): there is no resident grid to ask for
    one_per_transform = (kind == "fp32" and n == 8192) or (kind != "fp32" and n in (8192, 16384))
    for log2, grid in ((1, 0), (3, 0), (5, 0), (7, 0), (12, 0), (4, 1)):
        if grid and one_per_transform:
            continue
        got, _ = run(x, kind, True, options={"xcd_run_log2": log2, "grid_blocks_per_cu": grid})
        assert np.array_equal(got, want), (kind, n, log2, grid)


def test_linearity():
    n, batch = 2048, 33
    a = synth.uniform_complex(batch, n, seed=1)
    b = synth.uniform_complex(batch, n, seed=2)
    A, _ = run(a, "fp32")
    B, _ = run(b, "fp32")
    C, _ = run((0.5 * a - 0.25j * b).astype(np.complex64), "fp32")
    ref = 0.5 * A - 0.25j * B
    assert np.linalg.norm(C - ref) / np.linalg.norm(ref) < 2e-6


def test_exec_without_initialize_fails_loudly():
    fft_amd.ThunderFFTFinalize("bf16")
    d = torch.zeros(1, 64, 2, device="cuda", dtype=torch.bfloat16)
    with pytest.raises(fft_amd.ThunderFFTError) as e:
        fft_amd.ThunderFFT_global(d, d, 1, 64, True)
    assert e.value.code == 2


def test_native_library_is_the_path():
    # the product path must be the in-tree HIP library: it is mapped into this process
    maps = open("/proc/self/maps").read()
    assert "libthunderfft_hip.so" in maps


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("n", [256, 512, 1024, 2048, 4096])
def test_non_finite_input_stays_in_its_tile(dtype, n):
    """What the public header states about NaN / infinity: a non-finite value turns its own transform non-finite
    and, for 16-bit storage at 512 <= n <= 2048 on the default (matrix-core) plan, the other transforms of its
    4096-point tile; no transform outside the tile is touched. Plan 3 (the fp32-VALU kernels) confines it to the
    one transform at every length."""
    per_tile = max(1, 4096 // n)
    batch = 4 * per_tile
    dt = TORCH_DT[dtype]
    g = torch.Generator(device="cuda").manual_seed(n)
    x = ((torch.rand(batch, n, 2, device="cuda", generator=g) * 2 - 1) / n ** 0.5).to(dt)
    bad = per_tile + (per_tile // 2)  # a transform in the middle of the second tile
    x[bad, n // 3, 0] = float("nan")
    fft_amd.ThunderFFTInitialize(dtype, n)
    for plan in (0, 3):
        y = torch.empty_like(x)
        fft_amd.ThunderFFT_global(x, y, batch, n, True, options={"plan": plan} if plan else None)
        torch.cuda.synchronize()
        finite = torch.isfinite(y.float()).all(dim=(1, 2)).cpu().numpy()
        shared_tile = plan == 0 and 512 <= n <= 2048
        for t in range(batch):
            same_tile = t // per_tile == bad // per_tile
            if t == bad:
                assert not finite[t]
            elif same_tile and shared_tile:
                assert not finite[t], f"{dtype} n={n}: the header says transform {t} shares a matrix product with {bad}"
            else:
                assert finite[t], f"{dtype} n={n} plan={plan}: transform {t} must not see transform {bad}'s NaN"


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
def test_non_finite_input_at_n64_stays_in_its_row(dtype):
    """n = 64 with 16-bit storage runs on the matrix cores with four transforms per 256-point row sharing a
    block-diagonal product: the header says a non-finite value reaches exactly the four consecutive transforms that
    start at a multiple of four, and nothing else; plan 3 (the VALU kernel) keeps it to the one transform. At
    n = 128 the product is shared between the two transforms of a pair."""
    n, batch, bad = 64, 203, 133  # the batch does not fill its last row or its last tile
    dt = TORCH_DT[dtype]
    g = torch.Generator(device="cuda").manual_seed(5)
    x = ((torch.rand(batch, n, 2, device="cuda", generator=g) * 2 - 1) / n ** 0.5).to(dt)
    x[bad, 17, 1] = float("inf")
    fft_amd.ThunderFFTInitialize(dtype, n)
    for plan in (0, 3):
        y = torch.empty_like(x)
        fft_amd.ThunderFFT_global(x, y, batch, n, True, options={"plan": plan} if plan else None)
        torch.cuda.synchronize()
        finite = torch.isfinite(y.float()).all(dim=(1, 2)).cpu().numpy()
        for t in range(batch):
            hit = t == bad or (plan == 0 and t // 4 == bad // 4)
            assert finite[t] != hit, (dtype, plan, t)
    fft_amd.ThunderFFTInitialize(dtype, 128)
    x = ((torch.rand(9, 128, 2, device="cuda", generator=g) * 2 - 1) / 128 ** 0.5).to(dt)
    x[4, 3, 0] = float("nan")
    for plan, hits in ((3, {4}), (0, {4, 5})):
        y = torch.empty_like(x)
        fft_amd.ThunderFFT_global(x, y, 9, 128, True, options={"plan": plan} if plan else None)
        torch.cuda.synchronize()
        finite = torch.isfinite(y.float()).all(dim=(1, 2)).cpu().numpy()
        assert {t for t in range(9) if not finite[t]} == hits, (dtype, plan)


def test_grid_options_rejected_where_the_grid_is_the_batch():
    """fp32 n = 8192 and 16-bit n = 8192 / 16384 run one workgroup per transform: a resident grid does not exist
    and asking for one is an error, not a silent no-op."""
    for dtype, n in (("fp32", 8192), ("fp16", 8192), ("bf16", 16384)):
        x = torch.zeros(4, n, 2, device="cuda", dtype=TORCH_DT[dtype])
        fft_amd.ThunderFFTInitialize(dtype, n)
        for opts in ({"grid_blocks_per_cu": 2}, {"grid_blocks": 3}):
            with pytest.raises(fft_amd.ThunderFFTError) as e:
                fft_amd.ThunderFFT_global(x, x, 4, n, True, options=opts)
            assert e.value.code == 1
        fft_amd.ThunderFFT_global(x, x, 4, n, True, options={"xcd_run_log2": 2})  # other options still apply
    torch.cuda.synchronize()
