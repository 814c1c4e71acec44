"""GPU test of the device-level API (include/thunderfft/thunderfft_device.hpp) through a user
kernel that is not part of the library: examples/spectral_mask.hip runs forward transform, a mask
on the bins and the inverse transform on registers, in one launch. Checked against the same three
steps done with the fp64 oracle. The reference exposes the same composition through
ThunderFFT_gmem2smem ... ThunderFFT_kernel_reg ... ThunderFFT_smem2gmem
(thunderfft/include/thunderfft/thunderfft.cuh:55-130)."""
import ctypes
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402

import fft_amd  # noqa: E402
from fft_amd import _lib, synth  # noqa: E402

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
TORCH_DT = {"fp32": torch.float32, "fp16": torch.float16}


@pytest.fixture(scope="module")
def user_lib():
    _lib.load()  # the example links against libthunderfft_hip.so
    path = os.path.join(ROOT, "examples", "libspectral_mask.so")
    assert os.path.exists(path), "build it with `make examples`"
    lib = ctypes.CDLL(path)
    lib.spectral_mask.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong,
                                  ctypes.c_int, ctypes.c_int, ctypes.c_void_p]
    lib.spectral_mask.restype = ctypes.c_int
    lib.spectral_mask_staged.argtypes = lib.spectral_mask.argtypes
    lib.spectral_mask_staged.restype = ctypes.c_int
    return lib


@pytest.mark.parametrize("dtype,n,tol", [("fp32", 64, 1e-5), ("fp32", 256, 1e-5), ("fp32", 1024, 1e-5),
                                          ("fp32", 4096, 1e-5), ("fp16", 256, 1e-2), ("fp16", 4096, 1e-2)])
def test_user_kernel_band_stop(oracle_mod, user_lib, dtype, n, tol):
    batch = 3 * max(1, 2048 // n) + 1  # leaves a partly filled workgroup
    lo, hi = n // 8, n // 2 + 3
    x = synth.uniform_complex(batch, n, seed=n)
    dt = TORCH_DT[dtype]
    d_in = torch.from_numpy(x.view(np.float32).reshape(batch, n, 2)).to("cuda").to(dt).contiguous()
    d_out = torch.full_like(d_in, float("nan"))
    fft_amd.ThunderFFTInitialize(dtype, n)
    rc = user_lib.spectral_mask(fft_amd.api.dtype_code(dtype), n, d_in.data_ptr(), d_out.data_ptr(), batch, lo, hi,
                                torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    stored = d_in.float().cpu().numpy().astype(np.float64)
    stored = stored[..., 0] + 1j * stored[..., 1]
    X = oracle_mod.fft(stored, True)
    X[:, lo:hi] = 0
    ref = oracle_mod.fft(X, False) / n
    got = d_out.float().cpu().numpy().astype(np.float64)
    got = got[..., 0] + 1j * got[..., 1]
    err = oracle_mod.rel_l2(got, ref)
    assert err <= tol, f"{dtype} n={n} rel_l2={err:.3e}"


@pytest.mark.parametrize("dtype,n,tol", [("fp32", 64, 1e-5), ("fp32", 512, 1e-5), ("fp32", 2048, 1e-5),
                                          ("fp32", 4096, 1e-5), ("fp16", 1024, 1e-2),
                                          ("fp32", 128, 1e-5), ("fp32", 1024, 1e-5), ("fp16", 256, 1e-2)])
def test_user_kernel_band_stop_staged(oracle_mod, user_lib, dtype, n, tol):
    """The staged pipeline gmem2smem -> smem2reg -> kernel_reg -> reg2smem -> smem2gmem (the reference's
    thunderfft.cuh:68-89 calls): the spectrum visits LDS in natural order, so lengths whose transforms do not
    chain in registers (512, 2048) work too. The last three cases use a padded layout_t (thunderfft.cuh:11-33)
    through the layout-aware overloads."""
    batch = 3 * max(1, 2048 // n) + 1
    lo, hi = n // 8, n // 2 + 3
    x = synth.uniform_complex(batch, n, seed=n + 1)
    dt = TORCH_DT[dtype]
    d_in = torch.from_numpy(x.view(np.float32).reshape(batch, n, 2)).to("cuda").to(dt).contiguous()
    d_out = torch.full_like(d_in, float("nan"))
    fft_amd.ThunderFFTInitialize(dtype, n)
    rc = user_lib.spectral_mask_staged(fft_amd.api.dtype_code(dtype), n, d_in.data_ptr(), d_out.data_ptr(), batch, lo, hi,
                                       torch.cuda.current_stream().cuda_stream)
    assert rc == 0
    torch.cuda.synchronize()
    stored = d_in.float().cpu().numpy().astype(np.float64)
    stored = stored[..., 0] + 1j * stored[..., 1]
    X = oracle_mod.fft(stored, True)
    X[:, lo:hi] = 0
    ref = oracle_mod.fft(X, False) / n
    got = d_out.float().cpu().numpy().astype(np.float64)
    got = got[..., 0] + 1j * got[..., 1]
    err = oracle_mod.rel_l2(got, ref)
    assert err <= tol, f"{dtype} n={n} rel_l2={err:.3e}"


def test_twiddle_table_accessor():
    fft_amd.ThunderFFTInitialize("fp32", 512)
    p = ctypes.c_void_p()
    assert _lib.load().tfft_get_twiddle_table(0, 512, ctypes.byref(p)) == 0 and p.value
    q = ctypes.c_void_p()
    assert _lib.load().tfft_get_twiddle_table(0, 32, ctypes.byref(q)) == 2  # never initialised
