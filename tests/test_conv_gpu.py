"""GPU parity tests of the fused FFT convolution (tfft_conv_exec / tfft_conv_filter_layout), called
through the C ABI, against the fp64 circular-convolution oracle.

The oracle restates the reference's own validation loop
(thunderfft/sample/conv1d/conv1d_fullfft_l4096_thunder.cu:472-483); inputs follow its main():
signals uniform in [0, 1), taps uniform in [0, 1) scaled by 1/N (:346-361). Tolerances: relative L2
<= 1e-5 for fp32 storage, <= 1e-2 for fp16 storage.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402

import fft_amd  # noqa: E402
from fft_amd import conv  # noqa: E402

N = 4096
TORCH_DT = {"fp32": torch.float32, "fp16": torch.float16}


def interleave(a, dt):
    a = np.ascontiguousarray(a.astype(np.complex64))
    return torch.from_numpy(a.view(np.float32).reshape(a.shape + (2,))).to("cuda").to(dt).contiguous()


def to_complex(t):
    a = t.float().cpu().numpy().astype(np.float64)
    return a[..., 0] + 1j * a[..., 1]


def make_case(B, D, K, seed, real_input=False, real_filter=False):
    rng = np.random.default_rng(seed)
    x = rng.random((B, D, N)) + (0 if real_input else 1j * rng.random((B, D, N)))
    h = (rng.random((D, K)) + (0 if real_filter else 1j * rng.random((D, K)))) / N
    return x.astype(np.complex128), h.astype(np.complex128)


def run_conv(x, h, kind, dtype):
    dt = TORCH_DT[dtype]
    B, D, _ = x.shape
    d_h = interleave(h, dt)
    hp = conv.filter_spectra(d_h, N)
    if kind == conv.CONV_COMPLEX:
        d_x = interleave(x, dt)
        x_stored = to_complex(d_x)
    else:
        d_x = torch.from_numpy(np.ascontiguousarray(x.real.astype(np.float32))).to("cuda").to(dt).contiguous()
        x_stored = d_x.float().cpu().numpy().astype(np.float64) + 0j
    y = conv.fft_conv(d_x, hp, N, kind)
    torch.cuda.synchronize()
    h_stored = to_complex(d_h)
    if kind == conv.CONV_REAL_IN_OUT:
        got = y.float().cpu().numpy().astype(np.float64) + 0j
    else:
        got = to_complex(y)
    return got.reshape(B * D, N), x_stored.reshape(B * D, N), h_stored


@pytest.mark.parametrize("K", [1, 9, 63, 257])
def test_fp32_complex_vs_oracle(oracle_mod, K):
    x, h = make_case(3, 5, K, seed=K)
    got, xs, hs = run_conv(x, h, conv.CONV_COMPLEX, "fp32")
    ref = oracle_mod.circular_conv(xs, hs) * N  # unnormalised inverse
    err = oracle_mod.rel_l2(got, ref)
    assert err <= 1e-5, f"K={K} rel_l2={err:.3e}"
    assert err <= 3e-6, f"fp32 conv accuracy regressed: K={K} rel_l2={err:.3e}"


@pytest.mark.parametrize("kind", [conv.CONV_COMPLEX, conv.CONV_REAL_IN, conv.CONV_REAL_IN_OUT])
@pytest.mark.parametrize("K", [3, 33, 63])
def test_fp16_vs_oracle(oracle_mod, kind, K):
    real_in = kind != conv.CONV_COMPLEX
    real_f = kind == conv.CONV_REAL_IN_OUT
    x, h = make_case(4, 6, K, seed=100 + K + kind, real_input=real_in, real_filter=real_f)
    got, xs, hs = run_conv(x, h, kind, "fp16")
    ref = oracle_mod.circular_conv(xs, hs) * N
    if kind == conv.CONV_REAL_IN_OUT:
        ref = ref.real + 0j
    err = oracle_mod.rel_l2(got, ref)
    assert err <= 1e-2, f"kind={kind} K={K} rel_l2={err:.3e}"


@pytest.mark.parametrize("B", [1, 5])
def test_real_pairs_with_odd_batch(oracle_mod, B):
    # real in / real out packs signals (b, b+1) into one complex transform; an odd B leaves one alone
    x, h = make_case(B, 3, 17, seed=B, real_input=True, real_filter=True)
    got, xs, hs = run_conv(x, h, conv.CONV_REAL_IN_OUT, "fp16")
    ref = oracle_mod.circular_conv(xs, hs).real * N + 0j
    err = oracle_mod.rel_l2(got, ref)
    assert err <= 1e-2, f"B={B} rel_l2={err:.3e}"
    # per signal too: a swapped or dropped pair member would hide in the aggregate
    for s in range(got.shape[0]):
        assert oracle_mod.rel_l2(got[s], ref[s]) <= 1e-2, s


def test_signal_uses_filter_index_mod_D(oracle_mod):
    # filter d is a delta at tap d scaled by (d + 1) / N: signal (b, d) comes back shifted by d and scaled
    B, D = 3, 7
    rng = np.random.default_rng(5)
    x = (rng.random((B, D, N)) + 1j * rng.random((B, D, N))).astype(np.complex64)
    h = np.zeros((D, D), dtype=np.complex64)
    for d in range(D):
        h[d, d] = (d + 1) / N
    d_x = interleave(x, torch.float32)
    y = to_complex(conv.fft_conv(d_x, conv.filter_spectra(interleave(h, torch.float32), N), N))
    torch.cuda.synchronize()
    for d in range(D):
        want = np.roll(x[:, d].astype(np.complex128), d, axis=-1) * (d + 1)
        assert np.linalg.norm(y[:, d] - want) / np.linalg.norm(want) < 3e-6, d


def test_filter_spectra_in_one_kernel_equals_the_two_calls():
    """tfft_conv_filter_spectra (forward transform of the zero-padded taps straight into the order the fused kernel
    reads) against tfft_exec_c2c followed by tfft_conv_filter_layout: bit for bit, for one filter, an odd count and
    many; the convolution accepts the buffer as one the library laid out; fp32 has no one-kernel form."""
    g = torch.Generator(device="cuda").manual_seed(77)
    fft_amd.ThunderFFTInitialize("fp16", N)
    for D in (1, 7, 768):
        taps = torch.zeros(D, N, 2, device="cuda", dtype=torch.float16)
        taps[:, :63] = ((torch.rand(D, 63, 2, device="cuda", generator=g) * 2 - 1) / N).to(torch.float16)
        spec = torch.empty_like(taps)
        fft_amd.ThunderFFT_global(taps, spec, D, N, True)
        want = conv.filter_layout(spec, N)
        got = conv.filter_spectra_fused(taps, N)
        torch.cuda.synchronize()
        assert torch.equal(got, want), D
        x = torch.rand(2, D, N, device="cuda", generator=g).to(torch.float16)
        y = conv.fft_conv(x, got, N, conv.CONV_REAL_IN)  # the layout record of `got` is the matrix-core kernels'
        assert torch.equal(y, conv.fft_conv(x, want, N, conv.CONV_REAL_IN))
    with pytest.raises(fft_amd.ThunderFFTError) as e:
        conv.filter_spectra_fused(torch.zeros(2, N, 2, device="cuda"), N)
    assert e.value.code == 1  # TFFT_ERR_UNSUPPORTED
    with pytest.raises(fft_amd.ThunderFFTError) as e:  # a buffer laid out for the matrix-core kernels, read by the VALU one
        conv.fft_conv(x, got, N, conv.CONV_REAL_IN, options={"plan": 3})
    assert e.value.code == 3


def test_filter_layout_is_a_permutation():
    D = 3
    h = torch.arange(D * N * 2, dtype=torch.float32, device="cuda").reshape(D, N, 2).contiguous()
    hp = conv.filter_layout(h, N)
    torch.cuda.synchronize()
    a = h.reshape(D, N, 2).cpu().numpy()
    b = hp.reshape(D, N, 2).cpu().numpy()
    for d in range(D):
        # every (re, im) pair survives intact, and stays inside its own filter
        assert np.array_equal(np.sort(a[d, :, 0]), np.sort(b[d, :, 0]))
        assert np.array_equal(b[d, :, 1], b[d, :, 0] + 1)


def test_large_batch_linearity_and_shift():
    # full Figure-5 shape (B = 64, D = 768, real fp16 input, real filter): conv with a delta of weight
    # 1/N returns the input; checked on every signal
    B, D = 64, 768
    x = torch.rand(B, D, N, device="cuda").to(torch.float16)
    h = torch.zeros(D, 1, 2, device="cuda", dtype=torch.float16)
    h[:, 0, 0] = 1.0 / N
    hp = conv.filter_spectra(h, N)
    y = conv.fft_conv(x, hp, N, conv.CONV_REAL_IN_OUT)
    torch.cuda.synchronize()
    err = (torch.linalg.vector_norm((y.float() - x.float())) / torch.linalg.vector_norm(x.float())).item()
    assert err < 2e-3, err
    yc = conv.fft_conv(x, hp, N, conv.CONV_REAL_IN)
    torch.cuda.synchronize()
    # two different kernels (one signal per transform / two): results differ by a few fp16 ulps of 1.0
    assert torch.max(torch.abs(yc[..., 0].float() - y.float())).item() <= 4e-3
    assert torch.max(torch.abs(yc[..., 1].float())).item() < 4e-3


def test_resident_form_matches_one_shot():
    # the fp16 kernels as a resident grid (workgroups loop over signals, rows prefetched): same bits
    B, D = 40, 60
    x = torch.rand(B, D, N, device="cuda").to(torch.float16)
    h = (torch.rand(D, 9, 2, device="cuda") / N).to(torch.float16)
    h[..., 1] = 0
    hp = conv.filter_spectra(h, N)
    for kind in (conv.CONV_REAL_IN_OUT, conv.CONV_REAL_IN):
        a = conv.fft_conv(x, hp, N, kind)
        b = conv.fft_conv(x, hp, N, kind, options={"grid_blocks_per_cu": 2})
        torch.cuda.synchronize()
        assert torch.equal(a, b), kind


def test_valu_plan_and_filter_layout_mismatch(oracle_mod):
    """fp16 storage: plan 3 runs the fp32-VALU kernels, which read the permuted filters in another order than
    the matrix-core kernels. Same plan for layout and convolution: both kernels agree with the oracle; a buffer
    permuted for one kernel and handed to the other is rejected instead of convolving wrong."""
    B, D, K = 2, 3, 7
    rng = np.random.default_rng(5)
    x = torch.from_numpy(rng.random((B, D, N)).astype(np.float32)).to("cuda").to(torch.float16)
    taps = np.zeros((D, K, 2), dtype=np.float32)
    taps[..., 0] = rng.random((D, K)) / N
    h = torch.from_numpy(taps).to("cuda").to(torch.float16)
    ref = oracle_mod.circular_conv(x.float().cpu().numpy().astype(np.float64).reshape(B * D, N) + 0j,
                                   h.float().cpu().numpy().astype(np.float64)[..., 0] + 0j).real * N
    hp_mfma = conv.filter_spectra(h, N)
    hp_valu = conv.filter_spectra(h, N, options={"plan": 3})
    for hp, opts in ((hp_mfma, None), (hp_valu, {"plan": 3})):
        y = conv.fft_conv(x, hp, N, conv.CONV_REAL_IN_OUT, options=opts)
        torch.cuda.synchronize()
        got = y.float().cpu().numpy().astype(np.float64).reshape(B * D, N)
        assert oracle_mod.rel_l2(got + 0j, ref + 0j) <= 1e-2
    for hp, opts in ((hp_mfma, {"plan": 3}), (hp_valu, None)):
        with pytest.raises(fft_amd.ThunderFFTError) as e:
            conv.fft_conv(x, hp, N, conv.CONV_REAL_IN_OUT, options=opts)
        assert e.value.code == 3


@pytest.mark.parametrize("kind", [conv.CONV_COMPLEX, conv.CONV_REAL_IN, conv.CONV_REAL_IN_OUT])
def test_staged_kernel_matches_one_workgroup_per_signal(oracle_mod, kind):
    """fp16 storage: the default kernel (runs of signals per workgroup, the next signal landing in LDS by LDS-DMA
    under the current one's passes) does the same arithmetic as the one-workgroup-per-signal kernel (plan 4), so
    the results are bit-identical -- for runs of 2 and 8 forced on small shapes (plans 5, 6), for runs of 4 with
    every wave staging its own columns and the two pass images trading places from item to item (plan 7), for
    batches that do not fill the last run (odd run lengths included), for an odd number of real signals (a pair
    with one member), and at a shape large enough for the default to pick the staged form itself. One shape is
    also checked against the oracle."""
    rng = np.random.default_rng(kind)
    for B, D in ((1, 3), (5, 2), (17, 3), (19, 5), (64, 96)):
        shape = (B, D, N, 2) if kind == conv.CONV_COMPLEX else (B, D, N)
        x = torch.from_numpy(rng.random(shape).astype(np.float32)).to("cuda").to(torch.float16)
        taps = np.zeros((D, 9, 2), dtype=np.float32)
        taps[..., 0] = rng.random((D, 9)) / N
        h = torch.from_numpy(taps).to("cuda").to(torch.float16)
        hp = conv.filter_spectra(h, N)
        want = conv.fft_conv(x, hp, N, kind, options={"plan": 4})
        for plan in (0, 5, 6, 7):
            got = torch.full_like(want, float("nan"))
            conv.fft_conv(x, hp, N, kind, out=got, options={"plan": plan} if plan else None)
            torch.cuda.synchronize()
            assert torch.equal(got, want), (kind, B, D, plan)
        if (B, D) == (17, 3):
            xs = x.float().cpu().numpy().astype(np.float64)
            xs = (xs[..., 0] + 1j * xs[..., 1]) if kind == conv.CONV_COMPLEX else xs + 0j
            ref = oracle_mod.circular_conv(xs.reshape(B * D, N), h.float().cpu().numpy().astype(np.float64)[..., 0] + 0j) * N
            got = conv.fft_conv(x, hp, N, kind, options={"plan": 6}).float().cpu().numpy().astype(np.float64)
            got = (got[..., 0] + 1j * got[..., 1]) if kind != conv.CONV_REAL_IN_OUT else got + 0j
            assert oracle_mod.rel_l2(got.reshape(B * D, N), ref) <= 1e-2
    # in place (x and y alias): the next signal is fetched before the current one is stored, never the same bytes
    if kind == conv.CONV_COMPLEX:
        for plan in (6, 7):
            y = x.clone()
            conv.fft_conv(y, hp, N, kind, out=y, options={"plan": plan})
            torch.cuda.synchronize()
            assert torch.equal(y, conv.fft_conv(x, hp, N, kind, options={"plan": 4})), plan


def test_filter_layout_records_are_ranges_and_can_be_dropped():
    """The order check matches byte ranges (a pointer into a recorded buffer is still checked), a range refilled
    by the caller is released with tfft_conv_forget_layout, and tfft_finalize drops the records of its dtype."""
    B, D, K = 1, 4, 5
    x = torch.rand(B, D, N, device="cuda").to(torch.float16)
    h = (torch.rand(D, K, 2, device="cuda") / N).to(torch.float16)
    hp_mfma = conv.filter_spectra(h, N)
    hp_valu = conv.filter_spectra(h, N, options={"plan": 3})
    # filters 2 and 3 of the matrix-core buffer, through a pointer inside it
    with pytest.raises(fft_amd.ThunderFFTError) as e:
        conv.fft_conv(x[:, 2:].contiguous(), hp_mfma[2:], N, conv.CONV_REAL_IN_OUT, options={"plan": 3})
    assert e.value.code == 3
    # the caller overwrites that buffer with VALU-order spectra: stale record until it is dropped
    saved = hp_valu.clone()
    hp_mfma.copy_(saved)
    with pytest.raises(fft_amd.ThunderFFTError):
        conv.fft_conv(x, hp_mfma, N, conv.CONV_REAL_IN_OUT, options={"plan": 3})
    conv.forget_layout(hp_mfma)
    y = conv.fft_conv(x, hp_mfma, N, conv.CONV_REAL_IN_OUT, options={"plan": 3})
    want = conv.fft_conv(x, hp_valu, N, conv.CONV_REAL_IN_OUT, options={"plan": 3})
    torch.cuda.synchronize()
    assert torch.equal(y, want)
    # finalize forgets: the (now wrong) pairing is no longer rejected, the library only tracks what it wrote
    fft_amd.ThunderFFTFinalize("fp16")
    fft_amd.ThunderFFTInitialize("fp16", N)
    conv.fft_conv(x, hp_valu, N, conv.CONV_REAL_IN_OUT)
    torch.cuda.synchronize()


def test_errors():
    x = torch.zeros(1, 1, 1024, 2, device="cuda")
    with pytest.raises(fft_amd.ThunderFFTError) as e:
        conv.fft_conv(x, torch.zeros(1, 1024, 2, device="cuda"), 1024)
    assert e.value.code == 1  # unsupported length
    with pytest.raises(fft_amd.ThunderFFTError) as e:
        conv.fft_conv(torch.zeros(1, 2, N, device="cuda"), torch.zeros(2, N, 2, device="cuda"), N, conv.CONV_REAL_IN)
    assert e.value.code == 1  # real input exists for fp16 storage only
    with pytest.raises(fft_amd.ThunderFFTError):
        conv.fft_conv(torch.zeros(1, 2, N, 2, device="cuda"), torch.zeros(3, N, 2, device="cuda"), N)


# ---- overlap-save tiles of length 64 (tfft_conv_tiles_exec)
@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-5), ("fp16", 1e-2)])
@pytest.mark.parametrize("L,K", [(65536, 9), (5000, 1), (5000, 33), (777, 64), (64, 64), (100, 37)])
def test_tiled_conv_vs_oracle(oracle_mod, dtype, tol, L, K):
    rng = np.random.default_rng(L + K)
    x = (rng.random(L) + 1j * rng.random(L)).astype(np.complex64)
    taps = (rng.random(K) + 1j * rng.random(K)).astype(np.complex64)
    dt = TORCH_DT[dtype]
    d_x = interleave(x, dt)
    d_t = interleave(taps, dt)
    hf = conv.tiled_filter_spectrum(d_t, 64)
    y = conv.conv1d_tiled(d_x, hf, K, 64)
    torch.cuda.synchronize()
    assert y.shape == (L - K + 1, 2)
    ref = oracle_mod.valid_correlation(to_complex(d_x), to_complex(d_t))
    err = oracle_mod.rel_l2(to_complex(y), ref)
    assert err <= tol, f"L={L} K={K} rel_l2={err:.3e}"


def test_tiled_conv_does_not_write_past_output():
    L, K = 1000, 9
    x = torch.rand(L, 2, device="cuda")
    hf = conv.tiled_filter_spectrum(torch.rand(K, 2, device="cuda"), 64)
    buf = torch.full((L + 64, 2), -7.0, device="cuda")
    conv.conv1d_tiled(x, hf, K, 64, out=buf[: L - K + 1])
    torch.cuda.synchronize()
    assert torch.all(buf[L - K + 1:] == -7.0)
    assert torch.all(buf[: L - K + 1] != -7.0)


@pytest.mark.parametrize("dtype,tol", [("fp32", 1e-5), ("fp16", 1e-2)])
def test_convolution_golden_vectors(oracle_mod, dtype, tol):
    """Committed fixtures (tests/golden/conv_golden.npz, plain numpy sums of the definitions): complex and
    real-pair fused convolution of length 4096, and overlap-save tiles. Inputs are exact in fp16; taps go in scaled
    by 1/64 so that fp16 intermediates stay in range, outputs are N / 64 times the definition."""
    import os

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "conv_golden.npz"))
    dt = TORCH_DT[dtype]
    B, D, n = g["x"].shape
    hp = conv.filter_spectra(interleave(g["h"] / 64.0, dt), N)
    y = to_complex(conv.fft_conv(interleave(g["x"], dt), hp, N, conv.CONV_COMPLEX)) * (64.0 / N)
    assert oracle_mod.rel_l2(y.reshape(B * D, n), g["y"].reshape(B * D, n)) <= tol
    if dtype == "fp16":
        hp = conv.filter_spectra(interleave(g["h_real"] / 64.0 + 0j, dt), N)
        xr = torch.from_numpy(np.ascontiguousarray(g["x_real"].astype(np.float32))).to("cuda").to(dt).contiguous()
        yr = conv.fft_conv(xr, hp, N, conv.CONV_REAL_IN_OUT).float().cpu().numpy().astype(np.float64) * (64.0 / N)
        assert oracle_mod.rel_l2(yr.reshape(B * D, n) + 0j, g["y_real"].reshape(B * D, n)) <= tol
    taps = g["tiles_taps"]
    hf = conv.tiled_filter_spectrum(interleave(taps, dt), 64)
    yt = to_complex(conv.conv1d_tiled(interleave(g["tiles_x"], dt), hf, taps.shape[0], 64))
    assert oracle_mod.rel_l2(yt, g["tiles_y"]) <= tol
