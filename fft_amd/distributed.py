"""One large 1D complex FFT distributed over the ranks of a process group (four-step over RCCL).

N = N1 * N2 points over P ranks (P = 1, 2, 4, 8), fp32. Natural order in and out: rank r holds
x[r*N/P : (r+1)*N/P], i.e. rows n1 in [r*R1, (r+1)*R1) of the row-major [N1][N2] view (n = n1*N2 + n2,
R1 = N1/P, C2 = N2/P), and receives X[r*N/P : (r+1)*N/P] (k = k1 + N1*k2). The rank dimension is treated as a
radix-P stage of its own, so every local transform is a batch of short contiguous rows (length R1 or C2: one
LDS-resident kernel each for N up to 2^28 * P) and no local transpose of whole rows is needed:

  A-pack    x[r][n2]               -> S[d][c][r]          transpose, chunk d = columns of rank d       (permute)
  exchange                            T[s][c][r]          n1 = s*R1 + r, n2 = me*C2 + c
  1a        radix-P over s, times W_N1^(r*ks)             U[ks][c][r]                                   (radix)
  1b        P*C2 transforms of length R1 over r           V[ks][c][kr],  k1 = ks + P*kr                 (tfft_exec_c2c)
  B-pack    V times W_N^(n2*k1)    -> S[d][k1l][c]        transpose c <-> kr, k1 = d*R1 + k1l           (permute)
  exchange                            T[s][k1l][c]        n2 = s*C2 + c
  3a        radix-P over s, times W_N2^(c*ks)             U[ks][k1l][c]                                 (radix)
  3b        P*R1 transforms of length C2 over c           Z[ks][k1l][kc], k2 = ks + P*kc                (tfft_exec_c2c)
  C-pack    Z                      -> S[d][k2l][k1l]      transpose k1l <-> kc, k2 = d*C2 + k2l         (permute)
  exchange                            T[s][k2l][k1l]      k1 = s*R1 + k1l
  C-unpack                         -> X[k2l][k1]          chunks side by side                           (permute)

That is 8 passes over the rank's shard (each one read + one write) and 3 exchanges. Callers that can live with
other layouts skip work: `input_layout="transposed"` takes [C2][N1] (the rank's columns n2, all n1) and skips
A-pack and its exchange; `output_layout="transposed"` returns Z[ks][k1l][kc] = X[(me*R1 + k1l) + N1*(ks + P*kc)]
and skips C-pack, its exchange and C-unpack: 5 passes and ONE exchange. Each exchange moves the rank's whole
shard once; on 8 MI355X each rank sends 1/8 of it to every peer over its own xGMI link concurrently, so the
exchange is per-link bound. The reference has no distributed (or multi-GPU) code; this is new work on top of
its batched entry point.

The local passes are an object (`ops`): HipOps, the only implementation in this package, runs them through the
C ABI (tfft_dist_permute, tfft_dist_radix, tfft_exec_c2c). The tests bring their own fp64 evaluation of the same
pass descriptions to check the index maps with gloo ranks on the CPU. `run_virtual` steps P plans in lockstep
inside one process and copies the exchange chunks itself: the whole P-rank algorithm on one GPU.
"""
import numpy as np
import torch


def split_n(n, world):
    """N = N1 * N2 with N1 <= N2, both powers of two; R1 = N1/P and C2 = N2/P must leave even chunks per peer."""
    lg = n.bit_length() - 1
    if n <= 0 or (1 << lg) != n:
        raise ValueError(f"n={n} is not a power of two")
    if world not in (1, 2, 4, 8):
        raise ValueError(f"{world} ranks: the rank stage is a radix-1, 2, 4 or 8 butterfly")
    n1 = 1 << (lg // 2)
    n2 = n // n1
    if n1 % (2 * world * world) or n2 % (2 * world * world):
        raise ValueError(f"n={n} = {n1} x {n2} does not split over {world} ranks")
    return n1, n2


def permute_spec(nb1=1, nb2=1, ni=1, nj=1, in_b1=0, in_b2=0, in_i=0, out_b1=0, out_b2=0, out_j=0, transpose=1,
                 twiddle=None):
    """Arguments of one tfft_dist_permute pass (include/thunderfft_c.h); twiddle = dict(n, p0, pi, q0, qb1, qb2, qj)."""
    return dict(nb1=nb1, nb2=nb2, ni=ni, nj=nj, in_b1=in_b1, in_b2=in_b2, in_i=in_i, out_b1=out_b1, out_b2=out_b2,
                out_j=out_j, transpose=transpose, twiddle=twiddle)


def radix_spec(parts, rows, row_len, in_chunk, in_pitch, out_chunk, out_pitch):
    """Arguments of one tfft_dist_radix pass."""
    return dict(parts=parts, rows=rows, row_len=row_len, in_chunk=in_chunk, in_pitch=in_pitch, out_chunk=out_chunk,
                out_pitch=out_pitch)


def two_level_shift(n):
    return ((n.bit_length() - 1) + 1) // 2


class HipOps:
    """Local passes on the GPU through libthunderfft_hip.so. Buffers are float32 [points, 2] device tensors."""

    def __init__(self, device=None):
        import fft_amd

        self.f = fft_amd
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self._roots = {}
        self._two_level = {}
        self._initialized = set()

    def empty(self, points):
        return torch.empty(points, 2, dtype=torch.float32, device=self.device)

    def _upload(self, idx, n):
        a = -2.0 * np.pi * (idx % n).astype(np.float64) / n
        return torch.from_numpy(np.stack([np.cos(a), np.sin(a)], axis=-1).astype(np.float32)).to(self.device)

    def prepare(self, fft_lengths, root_tables, two_level_tables):
        """Everything that allocates or uploads, once per plan: twiddle tables of the local transforms, W_M tables
        of the radix passes, two-level W_N tables of the twiddling permute."""
        with torch.cuda.device(self.device):
            for n in fft_lengths:
                if n not in self._initialized:
                    self.f.ThunderFFTInitialize("fp32", n)
                    self._initialized.add(n)
        for m in root_tables:
            if m not in self._roots:
                self._roots[m] = self._upload(np.arange(m, dtype=np.int64), m)
        for n in two_level_tables:
            if n not in self._two_level:
                s = two_level_shift(n)
                hi = self._upload(np.arange(n >> s, dtype=np.int64) << s, n)
                lo = self._upload(np.arange(1 << s, dtype=np.int64), n)
                self._two_level[n] = (hi, lo, s)

    def fft_rows(self, buf, rows, n, forward):
        self.f.ThunderFFT_global(buf, buf, rows, n, forward)

    def permute(self, src, dst, spec, forward):
        from . import _lib, api

        a = _lib.DistPermuteArgs()
        for k in ("nb1", "nb2", "ni", "nj", "in_b1", "in_b2", "in_i", "out_b1", "out_b2", "out_j", "transpose"):
            setattr(a, k, int(spec[k]))
        hi = lo = None
        tw = spec["twiddle"]
        if tw:
            hi_t, lo_t, shift = self._two_level[tw["n"]]
            hi, lo = hi_t.data_ptr(), lo_t.data_ptr()
            a.twiddle, a.tw_n, a.tw_shift = 1, tw["n"], shift
            for k in ("p0", "pi", "q0", "qb1", "qb2", "qj"):
                setattr(a, "tw_" + k, int(tw[k]))
        stream = torch.cuda.current_stream(src.device)
        with torch.cuda.device(src.device):
            api._check(_lib.load().tfft_dist_permute(src.data_ptr(), dst.data_ptr(), a, hi, lo, int(bool(forward)),
                                                     stream.cuda_stream))

    def radix(self, src, dst, spec, forward):
        from . import _lib, api

        a = _lib.DistRadixArgs()
        for k, v in spec.items():
            setattr(a, k, int(v))
        table = self._roots[spec["parts"] * spec["row_len"]]
        stream = torch.cuda.current_stream(src.device)
        with torch.cuda.device(src.device):
            api._check(_lib.load().tfft_dist_radix(src.data_ptr(), dst.data_ptr(), a, table.data_ptr(),
                                                   int(bool(forward)), stream.cuda_stream))


class DistributedFFT:
    """Plan of one N-point transform over `world` ranks. `rank` / `world` default to the process group's; pass them
    (with group=None) to build the plan of a virtual rank for run_virtual."""

    def __init__(self, n, group=None, ops=None, rank=None, world=None):
        import torch.distributed as dist

        self.n = int(n)
        self.group = group
        live = dist.is_available() and dist.is_initialized() and world is None
        self.world = world if world is not None else (dist.get_world_size(group) if live else 1)
        self.rank = rank if rank is not None else (dist.get_rank(group) if live else 0)
        self.n1, self.n2 = split_n(self.n, self.world)
        self.r1, self.c2 = self.n1 // self.world, self.n2 // self.world
        self.points = self.n // self.world
        self.ops = ops if ops is not None else HipOps()
        P = self.world
        self.ops.prepare(fft_lengths={self.r1, self.c2}, root_tables={self.n1, self.n2} if P > 1 else set(),
                         two_level_tables={self.n})
        # two work buffers of one shard each, allocated once: every pass goes from one to the other or stays in place
        self.buf = [self.ops.empty(self.points), self.ops.empty(self.points)]
        self.passes = 0  # passes over the shard in the last transform (one read + one write each)

    def local_shape(self, layout="natural", side="input"):
        """Shape of this rank's shard: natural = contiguous run of N/P points ([R1][N2] in, [C2][N1] out);
        transposed input = [C2][N1] (columns n2 of the rank's block), transposed output = [P][R1][C2/1]:
        Z[ks][k1l][kc] = X[(rank*R1 + k1l) + N1*(ks + P*kc)]."""
        P = self.world
        if side == "input":
            return (self.r1, self.n2, 2) if layout == "natural" else (self.c2, self.n1, 2)
        return (self.c2, self.n1, 2) if layout == "natural" else (P, self.r1, self.c2, 2)

    def transposed_output_index(self):
        """k of every element of the transposed output, in its [P][R1][C2] order (for callers and tests)."""
        P = self.world
        ks, k1l, kc = np.meshgrid(np.arange(P), np.arange(self.r1), np.arange(self.c2), indexing="ij")
        return (self.rank * self.r1 + k1l) + self.n1 * (ks + P * kc)

    # ---- the pass descriptions (points; see the module docstring for the layouts)
    def _spec_a_pack(self):
        return permute_spec(nb1=self.world, ni=self.r1, nj=self.c2, in_b1=self.c2, in_i=self.n2,
                            out_b1=self.c2 * self.r1, out_j=self.r1)

    def _spec_1a(self, transposed_input):
        P, r1, c2 = self.world, self.r1, self.c2
        if transposed_input:
            return radix_spec(P, c2, r1, in_chunk=r1, in_pitch=self.n1, out_chunk=c2 * r1, out_pitch=r1)
        return radix_spec(P, c2, r1, in_chunk=c2 * r1, in_pitch=r1, out_chunk=c2 * r1, out_pitch=r1)

    def _spec_b_pack(self):
        P, r1, c2 = self.world, self.r1, self.c2
        tw = dict(n=self.n, p0=self.rank * c2, pi=1, q0=0, qb1=r1, qb2=1, qj=P)
        return permute_spec(nb1=P, nb2=P, ni=c2, nj=r1 // P, in_b1=r1 // P, in_b2=c2 * r1, in_i=r1,
                            out_b1=r1 * c2, out_b2=c2, out_j=P * c2, twiddle=tw)

    def _spec_3a(self):
        P, r1, c2 = self.world, self.r1, self.c2
        return radix_spec(P, r1, c2, in_chunk=r1 * c2, in_pitch=c2, out_chunk=r1 * c2, out_pitch=c2)

    def _spec_c_pack(self):
        P, r1, c2 = self.world, self.r1, self.c2
        return permute_spec(nb1=P, nb2=P, ni=r1, nj=c2 // P, in_b1=c2 // P, in_b2=r1 * c2, in_i=c2,
                            out_b1=c2 * r1, out_b2=r1, out_j=P * r1)

    def _spec_c_unpack(self):
        return permute_spec(nb1=self.world, ni=self.c2, nj=self.r1, in_b1=self.c2 * self.r1, in_i=self.r1,
                            out_b1=self.r1, out_j=self.n1, transpose=0)

    def steps(self, x_local, forward=True, input_layout="natural", output_layout="natural", out=None):
        """Generator over the transform: yields (send, recv) at every exchange point (chunk d of `send` goes to
        rank d, chunk s of `recv` comes from rank s; the caller performs it before resuming) and returns the
        rank's result. x_local is not modified. The result lives in `out` if given (natural output only), else
        in one of the plan's work buffers, valid until the next transform."""
        ops, P = self.ops, self.world
        if x_local.numel() != 2 * self.points:
            raise ValueError(f"shard holds {x_local.numel() // 2} points, expected {self.points}")
        x = x_local.reshape(self.points, 2)
        a, b = self.buf
        self.passes = 0

        def did(k=1):
            self.passes += k

        # ---- to T[s][c][r] in b (P = 1: [c][r] = the transposed matrix, no exchange)
        if input_layout == "natural":
            ops.permute(x, a, self._spec_a_pack(), True)  # a pure reordering: no direction
            did()
            if P > 1:
                yield a, b
                ops.radix(b, b, self._spec_1a(False), forward)
                did()
            else:
                a, b = b, a
        else:
            if P > 1:
                ops.radix(x, b, self._spec_1a(True), forward)
                did()
            else:
                b.copy_(x)
                did()
        ops.fft_rows(b, P * self.c2, self.r1, forward)
        did()
        # ---- twiddle + transpose, exchange, second half
        ops.permute(b, a, self._spec_b_pack(), forward)
        did()
        if P > 1:
            yield a, b
            ops.radix(b, b, self._spec_3a(), forward)
            did()
        else:
            a, b = b, a
        ops.fft_rows(b, P * self.r1, self.c2, forward)
        did()
        if output_layout == "transposed":
            return b.reshape(P, self.r1, self.c2, 2)
        # ---- back to natural order
        if P == 1:
            dst = out.reshape(self.points, 2) if out is not None else a
            ops.permute(b, dst, self._spec_c_pack(), True)
            did()
            return dst.reshape(self.c2, self.n1, 2)
        ops.permute(b, a, self._spec_c_pack(), True)
        did()
        yield a, b
        dst = out.reshape(self.points, 2) if out is not None else a
        ops.permute(b, dst, self._spec_c_unpack(), True)
        did()
        return dst.reshape(self.c2, self.n1, 2)

    def transform(self, x_local, forward=True, input_layout="natural", output_layout="natural", out=None):
        """Runs steps() with all_to_all_single over the plan's process group. Returns the rank's shard of the
        unnormalised transform (see steps() for where it lives)."""
        import torch.distributed as dist

        gen = self.steps(x_local, forward, input_layout, output_layout, out)
        try:
            while True:
                send, recv = next(gen)
                dist.all_to_all_single(recv, send, group=self.group)
        except StopIteration as done:
            return done.value


def run_virtual(plans, shards, forward=True, input_layout="natural", output_layout="natural"):
    """All ranks of one transform inside one process: plans[r] is the plan of virtual rank r (built with rank=r,
    world=P), shards[r] its input. The generators advance in lockstep and the exchanges are chunk copies. Returns
    the list of per-rank results (each valid until that plan's next transform)."""
    P = len(plans)
    gens = [p.steps(x, forward, input_layout, output_layout) for p, x in zip(plans, shards)]
    results = [None] * P
    while True:
        pending = []
        for r, g in enumerate(gens):
            try:
                pending.append(next(g))
            except StopIteration as done:
                results[r] = done.value
        if not pending:
            return results
        if len(pending) != P:
            raise RuntimeError("virtual ranks fell out of step")
        chunk = pending[0][0].shape[0] // P
        for d in range(P):
            for s in range(P):
                pending[d][1][s * chunk:(s + 1) * chunk].copy_(pending[s][0][d * chunk:(d + 1) * chunk])
