"""GPU tests of the device-level API under the reference's own signatures
(thunderfft/include/thunderfft/thunderfft.cuh:55-93, 128-130): examples/reference_shape.hip holds user kernels
whose call sequences are those of the reference's bench and convolution samples
(sample/bench/baseline/thunderfft_bench.h:57-96, sample/conv1d/conv1d_fullfft_l4096_thunder.cu:148-194,
sample/conv1d/main.cu:38-129) -- registers of the storage type, no transform object, the block's shared memory as
staging array and workspace at once, two-dimensional blocks. Checked against the fp64 oracle.
Tolerances: relative L2 <= 1e-5 for fp32 storage, <= 1e-2 for fp16 / bf16 storage."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402

import fft_amd  # noqa: E402
from fft_amd import _lib, conv, synth  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TORCH_DT = {"fp32": torch.float32, "fp16": torch.float16, "bf16": torch.bfloat16}
TOL = {"fp32": 1e-5, "fp16": 1e-2, "bf16": 1e-2}
LENGTHS = [64, 128, 256, 512, 1024, 2048, 4096, 8192, 16384]


@pytest.fixture(scope="module")
def ref_lib():
    _lib.load()
    path = os.path.join(ROOT, "examples", "libreference_shape.so")
    assert os.path.exists(path), "build it with `make examples`"
    lib = ctypes.CDLL(path)
    vp, ll, i = ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int
    lib.refshape_fft.argtypes = [i, i, i, i, i, vp, vp, ll, vp]
    lib.refshape_fft.restype = i
    lib.refshape_unpublish.argtypes = []
    lib.refshape_unpublish.restype = None
    lib.refshape_filter.argtypes = [i, i, vp, vp, vp, ll, i, vp]
    lib.refshape_filter.restype = i
    lib.refshape_tiles64.argtypes = [vp, vp, vp, ll, ll, i, vp]
    lib.refshape_tiles64.restype = i
    return lib


def upload(x, dtype):
    a = np.ascontiguousarray(x.astype(np.complex64))
    return torch.from_numpy(a.view(np.float32).reshape(a.shape + (2,))).to("cuda").to(TORCH_DT[dtype]).contiguous()


def download(t):
    a = t.float().cpu().numpy().astype(np.float64)
    return a[..., 0] + 1j * a[..., 1]


def stream():
    return torch.cuda.current_stream().cuda_stream


def run_fft(lib, dtype, n, staged, forward, publish, d_in):
    d_out = torch.full_like(d_in, float("nan"))
    rc = lib.refshape_fft(fft_amd.api.dtype_code(dtype), n, staged, forward, publish, d_in.data_ptr(), d_out.data_ptr(),
                          d_in.shape[0], stream())
    assert rc == 0, f"refshape_fft rc={rc}"
    torch.cuda.synchronize()
    return d_out


def case(n, dtype, seed):
    batch = 4 * max(1, 1024 // n) * (16 if n <= 128 else 1)  # a multiple of every batch_per_block
    x = synth.uniform_complex(batch, n, seed=seed)
    if dtype != "fp32":
        x = x / np.sqrt(n)  # the reference's 16-bit input scale (sample/bench/main.cu:36)
    return upload(x, dtype)


@pytest.mark.parametrize("staged", [0, 1], ids=["gmem2reg", "five-stage"])
@pytest.mark.parametrize("forward", [1, 0], ids=["fwd", "inv"])
@pytest.mark.parametrize("n", LENGTHS)
def test_fp32_transform(oracle_mod, ref_lib, n, forward, staged):
    d_in = case(n, "fp32", seed=3 * n + forward)
    got = download(run_fft(ref_lib, "fp32", n, staged, forward, 1, d_in))
    err = oracle_mod.rel_l2(got, oracle_mod.fft(download(d_in), bool(forward)))
    assert err <= 1e-5, f"n={n} rel_l2={err:.3e}"
    assert err <= 2e-6, f"fp32 accuracy regressed: n={n} rel_l2={err:.3e}"


@pytest.mark.parametrize("dtype", ["fp16", "bf16"])
@pytest.mark.parametrize("n,staged,forward", [(64, 1, 1), (128, 0, 0), (256, 1, 0), (512, 1, 1), (1024, 0, 1),
                                              (2048, 1, 1), (4096, 1, 1), (4096, 0, 0), (8192, 1, 0), (16384, 1, 1)])
def test_16bit_transform(oracle_mod, ref_lib, dtype, n, staged, forward):
    """Half-size registers and a half-size workspace: the staging array of vec2_t<T> is the workspace."""
    d_in = case(n, dtype, seed=5 * n + forward)
    got = download(run_fft(ref_lib, dtype, n, staged, forward, 1, d_in))
    err = oracle_mod.rel_l2(got, oracle_mod.fft(download(d_in), bool(forward)))
    assert err <= TOL[dtype], f"{dtype} n={n} rel_l2={err:.3e}"


@pytest.mark.parametrize("n", [64, 512, 4096, 16384])
def test_without_published_table(oracle_mod, ref_lib, n):
    """A unit that never saw ThunderFFTInitialize evaluates its twiddle bases in the kernel; the result must
    hold the same tolerance and sit within rounding of the table-driven one."""
    d_in = case(n, "fp32", seed=7 * n)
    with_table = download(run_fft(ref_lib, "fp32", n, 0, 1, 1, d_in))
    ref_lib.refshape_unpublish()
    without = download(run_fft(ref_lib, "fp32", n, 0, 1, 0, d_in))
    ref = oracle_mod.fft(download(d_in), True)
    assert oracle_mod.rel_l2(without, ref) <= 2e-6
    assert oracle_mod.rel_l2(without, with_table) <= 5e-7


def test_rejects_partial_block(ref_lib):
    d_in = case(64, "fp32", seed=1)[:17]
    d_out = torch.empty_like(d_in)
    assert ref_lib.refshape_fft(0, 64, 1, 1, 1, d_in.data_ptr(), d_out.data_ptr(), 17, stream()) == -3
    assert ref_lib.refshape_fft(0, 96, 1, 1, 1, d_in.data_ptr(), d_out.data_ptr(), 16, stream()) != 0


@pytest.mark.parametrize("dtype,n", [("fp32", 64), ("fp32", 128), ("fp32", 256), ("fp32", 1024), ("fp32", 4096),
                                     ("fp16", 64), ("fp16", 256), ("fp16", 4096)])
def test_fused_filter(oracle_mod, ref_lib, dtype, n):
    """forward transform -> product in register order -> inverse transform, the reference's fused convolution
    shape: y[b] = x[b] (*) h[b % D], circular."""
    D, K = 3, 9
    batch = 4 * max(1, 1024 // n) * (16 if n <= 128 else 1)
    rng = np.random.default_rng(n)
    x = rng.random((batch, n)) + 1j * rng.random((batch, n))
    h = (rng.random((D, K)) + 1j * rng.random((D, K))) / n
    d_x = upload(x, dtype)
    hz = np.zeros((D, n), dtype=np.complex128)
    hz[:, :K] = h
    d_h = upload(hz, dtype)
    H = fft_amd.fft(d_h)  # the library's own batched transform of the zero-padded taps
    d_y = torch.full_like(d_x, float("nan"))
    rc = ref_lib.refshape_filter(fft_amd.api.dtype_code(dtype), n, d_x.data_ptr(), H.data_ptr(), d_y.data_ptr(), batch, D,
                                 stream())
    assert rc == 0
    torch.cuda.synchronize()
    xs, hs = download(d_x), download(d_h)[:, :K]
    ref = oracle_mod.circular_conv(xs, hs) * n  # signal b uses filter b % D; the inverse is unnormalised
    err = oracle_mod.rel_l2(download(d_y), ref)
    assert err <= TOL[dtype], f"{dtype} n={n} rel_l2={err:.3e}"


@pytest.mark.parametrize("L,K", [(65536, 9), (5000, 33), (777, 64), (100, 37)])
def test_tiles64_reversed_layout(oracle_mod, ref_lib, L, K):
    """Overlap-save tiles gathered by the user kernel into a padded, base-4 digit-reversed layout
    (layout_t<64, 16, 1, 64, 64, 4, true>, the reference's sample/conv1d/main.cu:54)."""
    rng = np.random.default_rng(L + K)
    x = (rng.random(L) + 1j * rng.random(L)).astype(np.complex64)
    taps = (rng.random(K) + 1j * rng.random(K)).astype(np.complex64)
    d_x, d_t = upload(x, "fp16"), upload(taps, "fp16")
    hf = conv.tiled_filter_spectrum(d_t, 64)
    out_len = L - K + 1
    d_y = torch.full((out_len + 64, 2), -7.0, device="cuda", dtype=torch.float16)
    rc = ref_lib.refshape_tiles64(d_x.data_ptr(), hf.data_ptr(), d_y.data_ptr(), L, out_len, 64 - K + 1, stream())
    assert rc == 0
    torch.cuda.synchronize()
    assert torch.all(d_y[out_len:] == -7.0)
    ref = oracle_mod.valid_correlation(download(d_x), download(d_t))
    err = oracle_mod.rel_l2(download(d_y[:out_len]), ref)
    assert err <= 1e-2, f"L={L} K={K} rel_l2={err:.3e}"
    lib_y = conv.conv1d_tiled(d_x, hf, K, 64)  # the library's own kernel on the same data
    assert oracle_mod.rel_l2(download(d_y[:out_len]), download(lib_y)) <= 2e-3
