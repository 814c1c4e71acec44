"""CPU tests of the oracle: the two C evaluations against each other, against numpy.fft on
complex128, against the committed golden vectors and against the closed-form answers of the
reference's own test inputs."""
import os

import numpy as np
import pytest

from fft_amd import synth

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "fft_golden.npz")
ALL_N = (64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384)


@pytest.mark.parametrize("n", (2, 4, 8, 64, 256, 1024))
def test_direct_vs_radix2_vs_numpy(oracle_mod, n):
    x = synth.uniform_complex(3, n, seed=n)
    for fwd in (True, False):
        a = oracle_mod.dft_direct(x, fwd)
        b = oracle_mod.fft_radix2(x, fwd)
        c = oracle_mod.fft_numpy(x, fwd)
        assert oracle_mod.rel_l2(a, c) < 1e-13
        assert oracle_mod.rel_l2(b, c) < 1e-13


@pytest.mark.parametrize("n", ALL_N)
def test_oracle_matches_golden(oracle_mod, n):
    g = np.load(GOLDEN)
    x = g[f"x_{n}"]
    assert oracle_mod.rel_l2(oracle_mod.fft(x, True), g[f"fwd_{n}"]) < 1e-13
    assert oracle_mod.rel_l2(oracle_mod.fft(x, False), g[f"inv_{n}"]) < 1e-13


@pytest.mark.parametrize("n", (64, 128, 1024, 4096))
def test_kat_bench_sine(oracle_mod, n):
    # reference bench input (sample/bench/main.cu:35-41): only bins 1 and N-1 are non-zero
    x = synth.bench_sine(2, n)
    X = oracle_mod.fft(x)
    want = np.zeros((2, n), dtype=np.complex128)
    want[:, 1] = -0.5j * np.sqrt(n)
    want[:, n - 1] = +0.5j * np.sqrt(n)
    assert np.max(np.abs(X - want)) < 1e-6  # input is rounded to fp32


@pytest.mark.parametrize("n", (64, 256))
def test_kat_fft1d_tones(oracle_mod, n):
    # reference fft1d input (sample/fft1d/main.cu:16-30)
    batch = 9
    x = synth.fft1d_tones(batch, n)
    X = oracle_mod.fft(x)
    want = np.zeros((batch, n), dtype=np.complex128)
    for b in range(batch):
        want[b, (1 + b) % n] += n * (0.5 + 0.1 * b)
        want[b, 7] += 0.1 * n
    assert np.max(np.abs(X - want)) < 1e-4 * n * 1e-2


def test_kat_convolution_identity(oracle_mod):
    # IFFT(FFT(x) * FFT(h / N)) is the circular convolution
    # (sample/conv1d/conv1d_fullfft_l4096_thunder.cu:354-356, 472-490)
    n, k = 256, 7
    rng = np.random.default_rng(5)
    x = rng.standard_normal(n)
    h = np.zeros(n)
    h[:k] = rng.standard_normal(k)
    y = oracle_mod.fft(oracle_mod.fft(x) * oracle_mod.fft(h / n), forward=False)
    direct = np.array([sum(x[(i - j) % n] * h[j] for j in range(k)) for i in range(n)])
    assert np.max(np.abs(y - direct)) < 1e-12


def test_empty_batch(oracle_mod):
    x = np.zeros((0, 64), dtype=np.complex64)
    assert oracle_mod.fft(x).shape == (0, 64)


def test_roundtrip_and_linearity(oracle_mod):
    n = 512
    x = synth.uniform_complex(2, n, seed=3).astype(np.complex128)
    y = synth.uniform_complex(2, n, seed=4).astype(np.complex128)
    X, Y = oracle_mod.fft(x), oracle_mod.fft(y)
    assert oracle_mod.rel_l2(oracle_mod.fft(X, False) / n, x) < 1e-14
    assert oracle_mod.rel_l2(oracle_mod.fft(2.0 * x + 1j * y), 2.0 * X + 1j * Y) < 1e-14


def test_circular_conv_definition_vs_transform(oracle_mod):
    # the O(N K) definition (the reference's validation loop, conv1d_fullfft_l4096_thunder.cu:472-483)
    # against the convolution theorem evaluated with the oracle's own transforms
    rng = np.random.default_rng(3)
    x = rng.random((7, 512)) + 1j * rng.random((7, 512))
    h = rng.random((3, 17)) + 1j * rng.random((3, 17))
    a = oracle_mod.circular_conv(x, h)
    b = oracle_mod.circular_conv_fft(x, h)
    assert oracle_mod.rel_l2(a, b) < 1e-13
    # signal s uses filter s % D
    assert np.allclose(a[4], oracle_mod.circular_conv(x[4:5], h[1:2])[0])
    # a delta at tap k is a circular shift
    d = np.zeros((1, 4), dtype=complex)
    d[0, 3] = 2.0
    assert np.allclose(oracle_mod.circular_conv(x[:1], d)[0], 2.0 * np.roll(x[0], 3))


def test_valid_correlation_is_a_slice_of_circular_conv(oracle_mod):
    # sum_k x[i + k] f[k] (sample/conv1d/main.cu:156-174) equals the circular convolution of x with the
    # taps stored at (N - k) % N (:145-154) wherever no index wraps
    rng = np.random.default_rng(9)
    n, K = 64, 9
    x = rng.random(n) + 1j * rng.random(n)
    f = rng.random(K) + 1j * rng.random(K)
    padded = np.zeros((1, n), dtype=complex)
    padded[0, (n - np.arange(K)) % n] = f
    circ = oracle_mod.circular_conv(x[None, :], padded)[0]
    assert np.allclose(oracle_mod.valid_correlation(x, f), circ[: n - K + 1])


def test_convolution_golden_vectors(oracle_mod):
    """The oracle's two convolution forms against fixtures computed from the definitions with plain numpy sums
    (tests/golden/make_golden.py: the reference's validation loops, conv1d_fullfft_l4096_thunder.cu:472-483 and
    sample/conv1d/main.cu:156-174)."""
    import os

    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "conv_golden.npz"))
    B, D, n = g["x"].shape
    for x, h, y in ((g["x"], g["h"], g["y"]), (g["x_real"] + 0j, g["h_real"] + 0j, g["y_real"])):
        want = y.reshape(B * D, n)
        assert oracle_mod.rel_l2(oracle_mod.circular_conv(x.reshape(B * D, n), h), want) < 1e-14
        assert oracle_mod.rel_l2(oracle_mod.circular_conv_fft(x.reshape(B * D, n), h), want) < 1e-12
    assert oracle_mod.rel_l2(oracle_mod.valid_correlation(g["tiles_x"], g["tiles_taps"]), g["tiles_y"]) < 1e-14
