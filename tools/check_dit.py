import numpy as np, torch, sys
sys.path.insert(0, '/root/repo')
import fft_amd, oracle
from fft_amd import synth
for n in (8192, 16384):
    for fwd in (True, False):
        for batch in (1, 7):
            x = synth.uniform_complex(batch, n, seed=n)
            d = torch.from_numpy(x.view(np.float32).reshape(batch, n, 2)).cuda()
            fft_amd.ThunderFFTInitialize("fp32", n)
            for inplace in (False, True):
                src = d.clone()
                out = src if inplace else torch.full_like(d, float('nan'))
                fft_amd.ThunderFFT_global(src, out, batch, n, fwd, options={"plan": 5})
                torch.cuda.synchronize()
                a = out.cpu().numpy().astype(np.float64)
                err = oracle.rel_l2(a[..., 0] + 1j * a[..., 1], oracle.fft(x, fwd))
                print(n, fwd, batch, inplace, f"{err:.3e}")
                assert err < 2e-6
