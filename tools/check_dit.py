"""Quick GPU check of the decimation-in-time kernel of N = 8192 (and its one-pass alternatives) against the fp64 oracle."""
import sys

import numpy as np
import torch

sys.path.insert(0, __file__.rsplit("/", 2)[0])
import fft_amd  # noqa: E402
import oracle  # noqa: E402
from fft_amd import synth  # noqa: E402

for n, plan in ((8192, 0), (8192, 3), (8192, 4)):
    for fwd in (True, False):
        for batch in (1, 7):
            x = synth.uniform_complex(batch, n, seed=n)
            d = torch.from_numpy(x.view(np.float32).reshape(batch, n, 2)).cuda()
            fft_amd.ThunderFFTInitialize("fp32", n)
            for inplace in (False, True):
                src = d.clone()
                out = src if inplace else torch.full_like(d, float("nan"))
                fft_amd.ThunderFFT_global(src, out, batch, n, fwd, options={"plan": plan})
                torch.cuda.synchronize()
                a = out.cpu().numpy().astype(np.float64)
                err = oracle.rel_l2(a[..., 0] + 1j * a[..., 1], oracle.fft(x, fwd))
                print(n, plan, fwd, batch, inplace, f"{err:.3e}")
                assert err < 2e-6
