"""Rate of the headline transform when the data starts and ends in pinned host memory (H2D + transform + D2H),
for the note beside the device-resident figure; not what bench.py reports."""
import math
import sys
import time

import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import fft_amd  # noqa: E402

n, batch = 4096, 65536
h_in = torch.empty(batch, n, 2, dtype=torch.float32).pin_memory()
h_in.uniform_(-1, 1)
h_out = torch.empty_like(h_in).pin_memory()
d_in = torch.empty(batch, n, 2, device="cuda")
d_out = torch.empty_like(d_in)
fft_amd.ThunderFFTInitialize("fp32", n)


def once():
    d_in.copy_(h_in, non_blocking=True)
    fft_amd.ThunderFFT_global(d_in, d_out, batch, n, True)
    h_out.copy_(d_out, non_blocking=True)
    torch.cuda.synchronize()


once()
ts = []
for _ in range(5):
    t0 = time.perf_counter()
    once()
    ts.append(time.perf_counter() - t0)
t = min(ts)
print(f"N={n} batch={batch}: host -> device -> transform -> host {t * 1e3:.1f} ms "
      f"= {5.0 * n * math.log2(n) * batch / t * 1e-9:.0f} GFLOP/s, {2 * h_in.numel() * 4 / t * 1e-9:.1f} GB/s over PCIe (both directions summed)")
