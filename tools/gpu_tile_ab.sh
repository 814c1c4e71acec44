#!/bin/bash
# Multi-pass lengths: the 64 KiB tiles (plan 0) against 128 KiB tiles for the 256- and 512-point sweeps (plans 3, 4, 5),
# first for equal results, then interleaved and settled (groups of BURST launches).
set -u
R=${GRAFT_REPO_ROOT:-$(pwd)}
timeout -k 10 200 python3 - <<'PY' || exit 1
import torch, fft_amd
for lg in (16, 17, 18, 22, 24, 26):
    n = 1 << lg
    batch = max(1, (1 << 26) // n)
    x = torch.rand(batch, n, 2, device="cuda") * 2 - 1
    fft_amd.ThunderFFTInitialize("fp32", n)
    want = torch.empty_like(x)
    fft_amd.ThunderFFT_global(x, want, batch, n, True)
    for plan in (3, 4, 5):
        got = torch.empty_like(x)
        fft_amd.ThunderFFT_global(x, got, batch, n, True, options={"plan": plan})
        torch.cuda.synchronize()
        print(f"2^{lg} plan {plan}: {'equal' if torch.equal(got, want) else 'DIFFERENT'}", flush=True)
        assert torch.equal(got, want)
PY
for lg in ${LOGS:-16 18 22 24 26}; do
  n=$((1 << lg))
  timeout -k 10 200 python3 $R/tools/gpu_opt_sweep.py --n $n --points $((1 << 26)) --rounds ${ROUNDS:-10} --burst ${BURST:-10} default= wide=plan:3 wide512=plan:4 wide256=plan:5 || exit 1
done
