"""GPU parity tests of the multi-pass path (N > 16384, fp32): strided tile sweeps with four-step
twiddles, natural order in and out. The reference has no transform longer than 4096."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

import torch  # noqa: E402

import fft_amd  # noqa: E402
from fft_amd import synth  # noqa: E402


def run_big(x, forward=True, inplace=False):
    n = x.shape[-1]
    d = torch.from_numpy(np.ascontiguousarray(x).view(np.float32).reshape(x.shape + (2,))).to("cuda")
    out = d if inplace else torch.full_like(d, float("nan"))
    fft_amd.ThunderFFTInitialize("fp32", n)
    fft_amd.ThunderFFT_global(d, out, x.shape[0], n, forward)
    torch.cuda.synchronize()
    a = out.cpu().numpy().astype(np.float64)
    return a[..., 0] + 1j * a[..., 1]


@pytest.mark.parametrize("lg", (15, 16, 17, 18, 19, 20, 21, 22))
@pytest.mark.parametrize("forward", (True, False))
def test_big_vs_oracle(oracle_mod, lg, forward):
    n = 1 << lg
    batch = 2 if lg <= 18 else 1
    x = synth.uniform_complex(batch, n, seed=lg)
    got = run_big(x, forward)
    err = oracle_mod.rel_l2(got, oracle_mod.fft(x, forward))
    assert err <= 1e-5, f"n=2^{lg} forward={forward} rel_l2={err:.3e}"
    assert err <= 3e-6, f"accuracy regressed: n=2^{lg} rel_l2={err:.3e}"


@pytest.mark.parametrize("lg", (24, 26, 27, 28))
def test_very_big_vs_fp64_fft_and_roundtrip(oracle_mod, lg):
    """Whole transforms at the large-transform configuration's size and beyond (2^26 is that configuration; 2^28
    is a three-sweep plan with the longest tile lengths). Up to 2^26 the expected spectrum is the fp64 oracle's,
    computed on the host; beyond, torch's fp64 FFT on the device stands in (the oracle would need minutes)."""
    n = 1 << lg
    g = torch.Generator(device="cuda").manual_seed(lg)
    x = torch.rand(1, n, 2, device="cuda", generator=g) * 2 - 1
    fft_amd.ThunderFFTInitialize("fp32", n)
    X = torch.empty_like(x)
    fft_amd.ThunderFFT_global(x, X, 1, n, True)
    back = torch.empty_like(x)
    fft_amd.ThunderFFT_global(X, back, 1, n, False)
    torch.cuda.synchronize()
    if lg <= 26:
        xh = x.cpu().numpy().astype(np.float64).reshape(n, 2)
        ref = torch.from_numpy(oracle_mod.fft(xh[:, 0] + 1j * xh[:, 1], True)).to("cuda")
        del xh
    else:
        ref = torch.fft.fft(torch.view_as_complex(x.double().contiguous()), dim=-1)
    got = torch.view_as_complex(X.double().contiguous()).reshape(-1)
    err = (torch.linalg.vector_norm(got - ref.reshape(-1)) / torch.linalg.vector_norm(ref)).item()
    assert err <= 1e-5, f"n=2^{lg} rel_l2={err:.3e}"
    rt = (torch.linalg.vector_norm((back / n - x).double()) / torch.linalg.vector_norm(x.double())).item()
    assert rt <= 1e-5, rt
    del ref, got, X, back, x
    fft_amd.ThunderFFTFinalize("fp32")  # releases the scratch of the big plans
    torch.cuda.empty_cache()


@pytest.mark.parametrize("lg", (16, 17, 18, 22))
@pytest.mark.parametrize("forward", (True, False))
def test_wide_tile_plans_equal_the_default(lg, forward):
    """Plans 3, 4, 5 of a multi-pass length run their 256- and 512-point sweeps on tiles of twice as many transforms;
    every transform goes through the same arithmetic, so the results are the default's bit for bit. Plans the
    multi-pass lengths do not have are refused."""
    n = 1 << lg
    batch = 3 if lg <= 18 else 1
    g = torch.Generator(device="cuda").manual_seed(100 + lg)
    x = torch.rand(batch, n, 2, device="cuda", generator=g) * 2 - 1
    fft_amd.ThunderFFTInitialize("fp32", n)
    want = torch.empty_like(x)
    fft_amd.ThunderFFT_global(x, want, batch, n, forward)
    for plan in (3, 4, 5):
        got = torch.full_like(x, float("nan"))
        fft_amd.ThunderFFT_global(x, got, batch, n, forward, options={"plan": plan})
        torch.cuda.synchronize()
        assert torch.equal(got, want), f"n=2^{lg} plan {plan}"
    with pytest.raises(fft_amd.ThunderFFTError) as e:
        fft_amd.ThunderFFT_global(x, want, batch, n, forward, options={"plan": 6})
    assert e.value.code == 1  # TFFT_ERR_UNSUPPORTED


def test_batch_larger_than_the_scratch_round(oracle_mod):
    """The library folds batches through a scratch of 2^26 points: 2049 transforms of 2^15 points take two
    rounds. Every transform equals the one a small call computes; scattered ones are checked against the oracle."""
    n, batch = 1 << 15, 2049
    g = torch.Generator(device="cuda").manual_seed(77)
    x = torch.rand(batch, n, 2, device="cuda", generator=g) * 2 - 1
    fft_amd.ThunderFFTInitialize("fp32", n)
    X = torch.empty_like(x)
    fft_amd.ThunderFFT_global(x, X, batch, n, True)
    idx = [0, 1, 1023, 2047, 2048]
    small_in = x[idx].contiguous()
    small = torch.empty_like(small_in)
    fft_amd.ThunderFFT_global(small_in, small, len(idx), n, True)
    torch.cuda.synchronize()
    assert torch.equal(X[idx], small)
    xs = small_in.cpu().numpy().astype(np.float64)
    got = small.cpu().numpy().astype(np.float64)
    assert oracle_mod.rel_l2(got[..., 0] + 1j * got[..., 1], oracle_mod.fft(xs[..., 0] + 1j * xs[..., 1])) <= 3e-6
    # Parseval for every transform: a round that missed transforms would leave them unwritten
    ex = (x.double() ** 2).sum(dim=(1, 2))
    eX = (X.double() ** 2).sum(dim=(1, 2)) / n
    assert torch.max(torch.abs(eX / ex - 1)).item() < 1e-5


def test_two_streams_share_the_library_scratch():
    """Multi-pass calls on different streams use the same library-owned scratch; the library orders them on the
    device, so interleaved launches from two streams still give each call's own result."""
    n, batch = 1 << 16, 8
    g = torch.Generator(device="cuda").manual_seed(5)
    xa = torch.rand(batch, n, 2, device="cuda", generator=g) * 2 - 1
    xb = torch.rand(batch, n, 2, device="cuda", generator=g) * 2 - 1
    fft_amd.ThunderFFTInitialize("fp32", n)
    want_a, want_b = torch.empty_like(xa), torch.empty_like(xb)
    fft_amd.ThunderFFT_global(xa, want_a, batch, n, True)
    fft_amd.ThunderFFT_global(xb, want_b, batch, n, True)
    torch.cuda.synchronize()
    sa, sb = torch.cuda.Stream(), torch.cuda.Stream()
    for _ in range(10):
        ya, yb = torch.zeros_like(xa), torch.zeros_like(xb)
        torch.cuda.synchronize()
        fft_amd.ThunderFFT_global(xa, ya, batch, n, True, sa)
        fft_amd.ThunderFFT_global(xb, yb, batch, n, True, sb)
        fft_amd.ThunderFFT_global(ya, ya, batch, n, False, sa)
        fft_amd.ThunderFFT_global(xb, yb, batch, n, True, sb)
        torch.cuda.synchronize()
        assert torch.equal(yb, want_b)
        assert (torch.linalg.vector_norm(ya / n - xa) / torch.linalg.vector_norm(xa)).item() < 2e-6


def test_caller_workspace(oracle_mod):
    """tfft_exec_c2c_opt with a caller-owned workspace: same bits as the library-owned scratch, for a workspace of
    the best size, of the minimum size (one transform per round) and on two streams with a workspace each."""
    n, batch = 1 << 17, 5
    lo, best = fft_amd.workspace_bytes("fp32", n, batch)
    assert lo == 8 * n and best == 8 * n * batch
    g = torch.Generator(device="cuda").manual_seed(9)
    x = torch.rand(batch, n, 2, device="cuda", generator=g) * 2 - 1
    fft_amd.ThunderFFTInitialize("fp32", n)
    want = torch.empty_like(x)
    fft_amd.ThunderFFT_global(x, want, batch, n, True)
    torch.cuda.synchronize()
    for nbytes in (best, lo, lo + 8 * n // 2):
        ws = torch.empty(nbytes, dtype=torch.uint8, device="cuda")
        got = torch.zeros_like(x)
        fft_amd.ThunderFFT_global(x, got, batch, n, True, options={"workspace": ws})
        torch.cuda.synchronize()
        assert torch.equal(got, want), nbytes
    with pytest.raises(fft_amd.ThunderFFTError) as e:
        fft_amd.ThunderFFT_global(x, got, batch, n, True,
                                  options={"workspace": torch.empty(lo - 16, dtype=torch.uint8, device="cuda")})
    assert e.value.code == 3
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    w1 = torch.empty(best, dtype=torch.uint8, device="cuda")
    w2 = torch.empty(best, dtype=torch.uint8, device="cuda")
    y1, y2 = torch.zeros_like(x), torch.zeros_like(x)
    for _ in range(5):
        fft_amd.ThunderFFT_global(x, y1, batch, n, True, s1, options={"workspace": w1})
        fft_amd.ThunderFFT_global(x, y2, batch, n, True, s2, options={"workspace": w2})
    torch.cuda.synchronize()
    assert torch.equal(y1, want) and torch.equal(y2, want)


def test_big_in_place_and_tone(oracle_mod):
    n = 1 << 17
    x = synth.uniform_complex(1, n, seed=3)
    a = run_big(x, True, inplace=False)
    b = run_big(x, True, inplace=True)
    assert np.array_equal(a, b)
    # a single tone lands in one bin
    k0 = 12345
    t = np.exp(2j * np.pi * k0 * np.arange(n) / n).astype(np.complex64)[None]
    X = run_big(t, True)
    assert abs(X[0, k0] - n) < 1e-2 * n
    X[0, k0] = 0
    assert np.max(np.abs(X)) < 1e-3 * n


def test_big_supported_table():
    for lg in range(15, 31):
        assert fft_amd.is_supported("fp32", 1 << lg, True)
    assert not fft_amd.is_supported("fp32", 1 << 31 if False else 3 << 20, True)
    assert not fft_amd.is_supported("fp16", 1 << 20, True)
    info = fft_amd.plan_info("fp32", 1 << 26)
    prod = 1
    for r in info["radix"]:
        prod *= r
    assert prod == 1 << 26


def _to_dev(z):
    return torch.from_numpy(np.stack([z.real, z.imag], axis=-1).astype(np.float32)).to("cuda")


def _to_c(t):
    a = t.cpu().numpy().astype(np.float64)
    return a[..., 0] + 1j * a[..., 1]


@pytest.mark.parametrize("world,lg", ((1, 14), (1, 24), (2, 14), (4, 16), (8, 18), (8, 22)))
def test_distributed_virtual_ranks_hip(oracle_mod, world, lg):
    """The distributed four-step transform with every local pass on the HIP kernels (tfft_dist_permute,
    tfft_dist_radix, tfft_exec_c2c): all `world` ranks stepped inside this process on the one GPU, the exchanges
    as chunk copies (run_virtual). Natural and transposed layouts, both directions, against the fp64 oracle."""
    from fft_amd.distributed import DistributedFFT, HipOps, run_virtual

    n = 1 << lg
    x = synth.uniform_complex(1, n, seed=40 + lg)[0]
    ops = HipOps()
    plans = [DistributedFFT(n, ops=ops, rank=r, world=world) for r in range(world)]
    per = n // world
    shards = [_to_dev(x[r * per:(r + 1) * per]) for r in range(world)]
    n1, n2, c2 = plans[0].n1, plans[0].n2, plans[0].c2
    cols = [_to_dev(np.ascontiguousarray(x.reshape(n1, n2)[:, r * c2:(r + 1) * c2].T)) for r in range(world)]
    for forward in (True, False):
        if lg <= 20:
            ref = oracle_mod.fft(x, forward)
        else:
            xc = torch.from_numpy(x.astype(np.complex128)).to("cuda")
            ref = (torch.fft.fft(xc) if forward else torch.fft.ifft(xc) * n).cpu().numpy()
        res = run_virtual(plans, shards, forward)
        torch.cuda.synchronize()
        nat = np.concatenate([_to_c(t).reshape(-1) for t in res])
        err = oracle_mod.rel_l2(nat, ref)
        assert err <= 1e-5, f"n=2^{lg} world={world} forward={forward} rel_l2={err:.3e}"
        assert err <= 2e-6, err
        assert plans[0].passes == (8 if world > 1 else 5)
        for r in range(world):
            assert torch.equal(shards[r], _to_dev(x[r * per:(r + 1) * per]))  # inputs untouched
        res = run_virtual(plans, cols, forward, input_layout="transposed", output_layout="transposed")
        torch.cuda.synchronize()
        assert plans[0].passes == (5 if world > 1 else 4)
        for r in range(world):
            err = oracle_mod.rel_l2(_to_c(res[r]), ref[plans[r].transposed_output_index()])
            assert err <= 2e-6, (r, err)


def _direct_bins_fp64(x, ks, chunk=1 << 24):
    """X[k] = sum_n x[n] exp(-2 pi i n k / N) for the bins ks, summed in fp64 on the device straight from the
    definition (n k mod N is exact in int64; every cos / sin is an fp64 evaluation). x: [N, 2] float32."""
    n = x.shape[0]
    kt = torch.tensor(ks, dtype=torch.int64, device=x.device)
    acc = torch.zeros(len(ks), 2, dtype=torch.float64, device=x.device)
    for s in range(0, n, chunk):
        idx = torch.arange(s, min(n, s + chunk), dtype=torch.int64, device=x.device)
        xr, xi = x[s:s + chunk, 0].double(), x[s:s + chunk, 1].double()
        for g in range(0, len(ks), 4):
            ang = ((idx[None, :] * kt[g:g + 4, None]) % n).double() * (-2.0 * np.pi / n)
            c, sn = torch.cos(ang), torch.sin(ang)
            acc[g:g + 4, 0] += (xr * c - xi * sn).sum(dim=1)
            acc[g:g + 4, 1] += (xr * sn + xi * c).sum(dim=1)
    return acc


def _rel(a, b):
    return (torch.linalg.vector_norm((a - b).double()) / torch.linalg.vector_norm(b.double())).item()


def test_distributed_2p30_on_eight_virtual_ranks():
    """The 8-GPU configuration's single large transform, 2^30 points over 8 ranks (1 GiB per rank, R1 = C2 = 4096),
    with all eight ranks' passes stepped on the one GPU and the exchanges as chunk copies. Checked three ways:
    against tfft_exec_c2c(n = 2^30) on the same input (a different algorithm: three strided sweeps), against fp64
    direct sums of the definition at 40 scattered bins (five per rank's shard), and by an inverse round trip.
    Natural and transposed layouts."""
    from fft_amd.distributed import DistributedFFT, HipOps, run_virtual

    world, n = 8, 1 << 30
    free, _ = torch.cuda.mem_get_info()
    if free < 64 * 2 ** 30:
        pytest.skip("needs about 50 GiB of free device memory")
    g = torch.Generator(device="cuda").manual_seed(30)
    x = torch.rand(n, 2, device="cuda", generator=g) * 2 - 1
    ops = HipOps()
    plans = [DistributedFFT(n, ops=ops, rank=r, world=world) for r in range(world)]
    per, n1, n2, r1, c2 = n // world, plans[0].n1, plans[0].n2, plans[0].r1, plans[0].c2
    assert (n1, n2, r1, c2) == (1 << 15, 1 << 15, 4096, 4096)

    fft_amd.ThunderFFTInitialize("fp32", n)
    X = torch.empty_like(x)
    fft_amd.ThunderFFT_global(x.view(1, n, 2), X.view(1, n, 2), 1, n, True)

    rng = np.random.default_rng(30)
    ks = sorted({r * per + int(o) for r in range(world) for o in [0, per - 1, *rng.integers(1, per - 1, 3)]})
    want = _direct_bins_fp64(x, ks)
    scale = torch.linalg.vector_norm(want).item()
    kt = torch.tensor(ks, device="cuda")
    assert torch.linalg.vector_norm(X[kt].double() - want).item() <= 2e-6 * scale

    # natural order in and out: 8 passes, 3 exchanges
    res = run_virtual(plans, [x[r * per:(r + 1) * per] for r in range(world)], True)
    torch.cuda.synchronize()
    assert plans[0].passes == 8
    got_bins = torch.stack([res[k // per].reshape(per, 2)[k % per] for k in ks])
    assert torch.linalg.vector_norm(got_bins.double() - want).item() <= 2e-6 * scale
    for r in range(world):
        # two fp32 results of different algorithms, each within about 8e-7 of the exact spectrum at this length
        err = _rel(res[r].reshape(per, 2), X[r * per:(r + 1) * per])
        assert err <= 2e-6, f"rank {r}: distributed vs single-GPU transform rel_l2={err:.3e}"

    # inverse of the distributed result, rank by rank from copies (the results live in the plans' work buffers)
    spectrum = [t.reshape(per, 2).clone() for t in res]
    back = run_virtual(plans, spectrum, False)
    torch.cuda.synchronize()
    for r in range(world):
        err = _rel(back[r].reshape(per, 2) / n, x[r * per:(r + 1) * per])
        assert err <= 2e-6, f"rank {r}: round trip rel_l2={err:.3e}"
    del spectrum, back

    # transposed layouts on both sides: 5 passes, ONE exchange
    cols = [x.view(n1, n2, 2)[:, r * c2:(r + 1) * c2].transpose(0, 1).contiguous() for r in range(world)]
    res = run_virtual(plans, cols, True, input_layout="transposed", output_layout="transposed")
    torch.cuda.synchronize()
    assert plans[0].passes == 5
    Xk = X.view(c2, world, n1, 2)  # X[k1 + N1 (ks + P kc)] as [kc][ks][k1]
    for r in range(world):
        expect = Xk[:, :, r * r1:(r + 1) * r1].permute(1, 2, 0, 3)  # [ks][k1l][kc], the rank's transposed output
        err = _rel(res[r], expect)
        assert err <= 2e-6, f"rank {r}: transposed layout rel_l2={err:.3e}"
    del cols, res, plans, ops, X, x
    fft_amd.ThunderFFTFinalize("fp32")
    torch.cuda.empty_cache()


def test_dist_pass_kernels_match_their_definition():
    """tfft_dist_permute and tfft_dist_radix against the fp64 evaluation of their definitions (NumpyOps) on
    shapes that are not multiples of the kernels' tiles (64 x 64 transposing, 8 x 512 copying), with batch strides, with and without the twiddle, both
    directions, in place for the radix pass."""
    from dist_reference import NumpyOps
    from fft_amd.distributed import HipOps, permute_spec, radix_spec

    hip, ref = HipOps(), NumpyOps()
    rng = np.random.default_rng(3)
    n_tw = 1 << 20
    hip.prepare(set(), {2 * 128, 4 * 128, 8 * 64}, {n_tw})
    specs = [
        permute_spec(nb1=3, nb2=2, ni=70, nj=46, in_b1=46, in_b2=3 * 46 * 80, in_i=3 * 46 + 2, out_b1=2 * 46 * 72,
                     out_b2=46 * 72, out_j=72),
        permute_spec(nb1=2, nb2=3, ni=64, nj=130, in_b1=3 * 64 * 130, in_b2=64 * 130, in_i=130, out_b1=3 * 130 * 64,
                     out_b2=130 * 64, out_j=64,
                     twiddle=dict(n=n_tw, p0=12345, pi=3, q0=7, qb1=1001, qb2=17, qj=5)),
        permute_spec(nb1=4, nb2=1, ni=33, nj=200, in_b1=33 * 200, in_i=200, out_b1=200, out_j=4 * 200, transpose=0),
        permute_spec(nb1=2, nb2=1, ni=10, nj=64, in_b1=640, in_i=64, out_b1=640, out_j=64, transpose=0,
                     twiddle=dict(n=n_tw, p0=1, pi=1, q0=0, qb1=64, qb2=0, qj=1)),
        # copies wider than one 512-point row segment, ragged in both directions, with a twiddle that depends on
        # every index
        permute_spec(nb1=2, nb2=2, ni=11, nj=1030, in_b1=2 * 11 * 1032, in_b2=11 * 1032, in_i=1032, out_b1=1030,
                     out_b2=2 * 1030 * 12, out_j=2 * 1030, transpose=0,
                     twiddle=dict(n=n_tw, p0=3, pi=7, q0=11, qb1=13, qb2=1000, qj=9)),
        # long rows, no twiddle: the kernel that moves runs of 4096 points of one row (ragged last run)
        permute_spec(nb1=2, nb2=1, ni=3, nj=4100, in_b1=3 * 4100, in_i=4100, out_b1=4102, out_j=2 * 4102, transpose=0),
        # transposing with the larger pitch on the output side (tiles then walk along i first)
        permute_spec(nb1=1, nb2=2, ni=130, nj=66, in_b1=0, in_b2=130 * 66, in_i=66, out_b2=132, out_j=2 * 132 + 4),
    ]
    for spec in specs:
        points = 1 << 16
        z = rng.standard_normal(points) + 1j * rng.standard_normal(points)
        for forward in (True, False):
            src_h, dst_h = _to_dev(z), torch.zeros(points, 2, device="cuda")
            src_r = torch.from_numpy(np.stack([z.real, z.imag], axis=-1))
            dst_r = ref.empty(points)
            hip.permute(src_h, dst_h, spec, forward)
            ref.permute(src_r, dst_r, spec, forward)
            torch.cuda.synchronize()
            got, want = _to_c(dst_h), dst_r.numpy()[:, 0] + 1j * dst_r.numpy()[:, 1]
            assert np.max(np.abs(got - want)) < 2e-6 * max(1.0, np.max(np.abs(want))), spec
    for parts, rows, row_len, pitch in ((2, 5, 128, 130), (4, 3, 128, 128), (8, 7, 64, 64)):
        chunk = rows * pitch + 6
        points = parts * chunk
        z = rng.standard_normal(points) + 1j * rng.standard_normal(points)
        spec = radix_spec(parts, rows, row_len, chunk, pitch, chunk, pitch)
        for forward in (True, False):
            buf_h = _to_dev(z)
            hip.radix(buf_h, buf_h, spec, forward)  # in place
            src_r = torch.from_numpy(np.stack([z.real, z.imag], axis=-1))
            out_r = src_r.clone()  # the gaps between rows keep their input values, as in place on the device
            ref.radix(src_r, out_r, spec, forward)
            torch.cuda.synchronize()
            got, want = _to_c(buf_h), out_r.numpy()[:, 0] + 1j * out_r.numpy()[:, 1]
            assert np.max(np.abs(got - want)) < 4e-6 * np.max(np.abs(want)), (parts, forward)


def test_fourstep_twiddle_kernel(oracle_mod):
    from fft_amd import _lib, api

    rows, cols, row0, col0, n = 37, 64, 1000, 6, 1 << 22
    x = synth.uniform_complex(rows, cols, seed=8)
    t = torch.from_numpy(x.view(np.float32).reshape(rows, cols, 2)).to("cuda")
    api._check(_lib.load().tfft_fourstep_twiddle(0, t.data_ptr(), rows, cols, row0, col0, n, 1,
                                                  torch.cuda.current_stream().cuda_stream))
    torch.cuda.synchronize()
    a = t.cpu().numpy().astype(np.float64)
    q = (row0 + np.arange(rows))[:, None] * (col0 + np.arange(cols))[None, :]
    ref = x.astype(np.complex128) * np.exp(-2j * np.pi * (q % n) / n)
    assert oracle_mod.rel_l2(a[..., 0] + 1j * a[..., 1], ref) < 2e-7
