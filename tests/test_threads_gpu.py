"""Host-thread safety of the C ABI (the plan cache is shared by every caller of the process): two threads transform
on streams of their own while a third initialises other lengths, and later finalises another element type. ctypes
releases the interpreter lock around every library call, so the calls really overlap. The reference's twiddle
cache is a function-local static with no lock (thunderfft/include/thunderfft/detail/thunderfft_impl_twiddle.cuh:
30-34, 268); a thread-safe cache is this build's own contract (include/thunderfft_c.h)."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402

import fft_amd  # noqa: E402
from fft_amd import synth  # noqa: E402


def test_two_streams_two_threads_while_a_third_initialises(oracle_mod):
    n, batch, rounds = 2048, 96, 60
    x = synth.uniform_complex(2 * batch, n, seed=17)
    d = torch.from_numpy(x.view(np.float32).reshape(2 * batch, n, 2)).to("cuda")
    halves = [d[:batch].contiguous(), d[batch:].contiguous()]
    fft_amd.ThunderFFTInitialize("fp32", n)
    want = []
    for h in halves:
        y = torch.empty_like(h)
        fft_amd.ThunderFFT_global(h, y, batch, n, True)
        want.append(y)
    torch.cuda.synchronize()
    ref = oracle_mod.fft(x[:batch], True)
    got0 = want[0].cpu().numpy().astype(np.float64)
    assert oracle_mod.rel_l2(got0[..., 0] + 1j * got0[..., 1], ref) <= 2e-6

    errors, start = [], threading.Barrier(3)

    def worker(i):
        try:
            torch.cuda.set_device(0)
            stream = torch.cuda.Stream()
            out = torch.empty_like(halves[i])
            start.wait()
            for _ in range(rounds):
                out.fill_(float("nan"))  # on torch's current stream of this thread
                stream.wait_stream(torch.cuda.current_stream())
                fft_amd.ThunderFFT_global(halves[i], out, batch, n, True, stream)
                stream.synchronize()
                if not torch.equal(out, want[i]):
                    errors.append(f"thread {i}: result changed under concurrent calls")
                    return
        except Exception as exc:  # noqa: BLE001
            errors.append(f"thread {i}: {exc!r}")

    def initialiser():
        try:
            torch.cuda.set_device(0)
            start.wait()
            for _ in range(6):
                for m in (64, 128, 256, 512, 1024, 4096, 8192, 16384, 1 << 15):
                    fft_amd.ThunderFFTInitialize("fp32", m)
                for m in (256, 1024, 4096):
                    fft_amd.ThunderFFTInitialize("fp16", m)
                    fft_amd.plan_info("fp16", m)
                # another element type comes and goes while fp32 transforms are in flight
                fft_amd.ThunderFFTFinalize("fp16")
        except Exception as exc:  # noqa: BLE001
            errors.append(f"initialiser: {exc!r}")

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)] + [threading.Thread(target=initialiser)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=120)
    assert not any(t.is_alive() for t in threads), "a thread is stuck"
    assert not errors, errors


def test_finalize_races_exec_without_faulting():
    """tfft_finalize of the element type that is being used: every concurrent call either ran with live tables
    (right result) or was refused with TFFT_ERR_NOT_INITIALIZED; none reads a freed table."""
    n, batch = 1024, 64
    g = torch.Generator(device="cuda").manual_seed(4)
    x = torch.rand(batch, n, 2, device="cuda", generator=g) * 2 - 1
    fft_amd.ThunderFFTInitialize("fp32", n)
    want = torch.empty_like(x)
    fft_amd.ThunderFFT_global(x, want, batch, n, True)
    torch.cuda.synchronize()
    errors, outcomes, stop = [], {"ok": 0, "refused": 0}, threading.Event()

    def worker():
        try:
            torch.cuda.set_device(0)
            stream = torch.cuda.Stream()
            out = torch.empty_like(x)
            while not stop.is_set():
                try:
                    fft_amd.ThunderFFT_global(x, out, batch, n, True, stream)
                except fft_amd.ThunderFFTError as e:
                    if e.code != 2:
                        raise
                    outcomes["refused"] += 1
                    continue
                stream.synchronize()
                if not torch.equal(out, want):
                    errors.append("a transform ran on a table that was being freed")
                    return
                outcomes["ok"] += 1
        except Exception as exc:  # noqa: BLE001
            errors.append(repr(exc))

    t = threading.Thread(target=worker)
    t.start()
    for _ in range(40):
        fft_amd.ThunderFFTFinalize("fp32")
        fft_amd.ThunderFFTInitialize("fp32", n)
    stop.set()
    t.join(timeout=60)
    assert not t.is_alive() and not errors, errors
    assert outcomes["ok"] > 0
