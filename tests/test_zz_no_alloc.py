"""tfft_exec_c2c_static, the entry point that allocates nothing: its multi-pass sweeps go through a scratch that
is part of the library's code object. Called through the C ABI and compared bit for bit with tfft_exec_c2c (same
kernels in the same order, so the bar is equality), on two calls in a row, the second with a smaller input;
then the size limit and the forwarding of one-kernel lengths."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402

import fft_amd  # noqa: E402
from fft_amd import _lib, api  # noqa: E402


def exec_static(x, out, batch, n, forward=True, stream=None):
    stream = stream or torch.cuda.current_stream(x.device)
    with torch.cuda.device(x.device):
        api._check(_lib.load().tfft_exec_c2c_static(api.dtype_code(x.dtype), n, int(forward), x.data_ptr(), out.data_ptr(),
                                                    batch, stream.cuda_stream))
    return out


def test_static_scratch_matches_the_allocating_path(oracle_mod):
    g = torch.Generator(device="cuda").manual_seed(2024)
    # first call: 2^19 points x 5 (two transforms per round of the 2^20-point scratch, a ragged last round);
    # second call: a smaller length and batch, through the same scratch, which still holds the first call's data
    for n, batch in ((1 << 19, 5), (1 << 15, 3)):
        x = torch.rand(batch, n, 2, device="cuda", generator=g) * 2 - 1
        fft_amd.ThunderFFTInitialize("fp32", n)
        for forward in (True, False):
            want = torch.empty_like(x)
            fft_amd.ThunderFFT_global(x, want, batch, n, forward)
            got = torch.full_like(x, float("nan"))
            exec_static(x, got, batch, n, forward)
            torch.cuda.synchronize()
            assert torch.equal(got, want), (n, batch, forward)
        # and against the fp64 oracle, like every other entry point
        xs = x[:1].cpu().numpy().astype(np.float64)
        y = exec_static(x, torch.empty_like(x), batch, n, True)
        torch.cuda.synchronize()
        ys = y[:1].cpu().numpy().astype(np.float64)
        assert oracle_mod.rel_l2(ys[..., 0] + 1j * ys[..., 1], oracle_mod.fft(xs[..., 0] + 1j * xs[..., 1])) <= 3e-6


def test_static_entry_allocates_nothing():
    """The property the entry point exists for: across calls made after tfft_initialize the device's free memory
    does not move (nothing allocated, nothing freed), and the library-owned scratch of the allocating path is not
    touched: with that scratch poisoned by NaNs the static results are still exact."""
    n, batch = 1 << 18, 6
    g = torch.Generator(device="cuda").manual_seed(3)
    x = torch.rand(batch, n, 2, device="cuda", generator=g) * 2 - 1
    fft_amd.ThunderFFTInitialize("fp32", n)
    want = torch.empty_like(x)
    fft_amd.ThunderFFT_global(x, want, batch, n, True)
    got = torch.empty_like(x)
    exec_static(x, got, batch, n, True)  # first use on this device done: addresses resolved, code loaded
    torch.cuda.synchronize()
    free_before, _ = torch.cuda.mem_get_info()
    for forward in (True, False, True):
        exec_static(x, got, batch, n, forward)
    torch.cuda.synchronize()
    free_after, _ = torch.cuda.mem_get_info()
    assert free_after == free_before, (free_before, free_after)
    assert torch.equal(got, want)
    # The allocating path leaves its intermediate sweeps in the plan's own scratch; the static path must not read it.
    # Run the allocating path on NaN input (its scratch now holds NaNs), then the static path on real input.
    poison = torch.full_like(x, float("nan"))
    fft_amd.ThunderFFT_global(poison, torch.empty_like(x), batch, n, True)
    got.zero_()
    exec_static(x, got, batch, n, True)
    torch.cuda.synchronize()
    assert torch.equal(got, want)


def test_static_in_place_and_full_scratch():
    n, batch = 1 << 20, 2  # one transform fills the whole scratch
    g = torch.Generator(device="cuda").manual_seed(7)
    x = torch.rand(batch, n, 2, device="cuda", generator=g) * 2 - 1
    fft_amd.ThunderFFTInitialize("fp32", n)
    want = torch.empty_like(x)
    fft_amd.ThunderFFT_global(x, want, batch, n, True)
    y = x.clone()
    exec_static(y, y, batch, n, True)
    torch.cuda.synchronize()
    assert torch.equal(y, want)


def test_static_entry_under_graph_capture():
    """The callers this entry point exists for: captured into a HIP graph (no allocation, no synchronisation, no
    event of the eager world) and replayed, it gives the eager result every time."""
    n, batch = 1 << 16, 3
    g = torch.Generator(device="cuda").manual_seed(11)
    x = torch.rand(batch, n, 2, device="cuda", generator=g) * 2 - 1
    fft_amd.ThunderFFTInitialize("fp32", n)
    want = torch.empty_like(x)
    fft_amd.ThunderFFT_global(x, want, batch, n, True)
    torch.cuda.synchronize()
    out = torch.zeros_like(x)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        exec_static(x, out, batch, n, True)  # on the capture stream (torch's current stream)
    for _ in range(3):
        out.zero_()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, want)


def test_static_limits_and_forwarding():
    n = 1 << 21
    x = torch.zeros(1, n, 2, device="cuda")
    fft_amd.ThunderFFTInitialize("fp32", n)
    with pytest.raises(fft_amd.ThunderFFTError) as e:
        exec_static(x, x, 1, n)
    assert e.value.code == 1 and "static scratch" in str(e.value)
    with pytest.raises(fft_amd.ThunderFFTError) as e:
        exec_static(torch.zeros(1, 1 << 16, 2, device="cuda"), torch.zeros(1, 1 << 16, 2, device="cuda"), -1, 1 << 16)
    assert e.value.code == 3
    # one-kernel lengths use no scratch: forwarded, every storage type
    for dt, n in ((torch.float32, 4096), (torch.float16, 1024), (torch.bfloat16, 8192)):
        g = torch.Generator(device="cuda").manual_seed(n)
        x = ((torch.rand(7, n, 2, device="cuda", generator=g) * 2 - 1) / n ** 0.5).to(dt)
        fft_amd.ThunderFFTInitialize(dt, n)
        want = torch.empty_like(x)
        fft_amd.ThunderFFT_global(x, want, 7, n, True)
        got = exec_static(x, torch.empty_like(x), 7, n)
        torch.cuda.synchronize()
        assert torch.equal(got.view(torch.int16 if dt != torch.float32 else torch.int32),
                           want.view(torch.int16 if dt != torch.float32 else torch.int32))
