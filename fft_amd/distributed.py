"""One large 1D complex FFT distributed over the ranks of a process group (four-step over RCCL).

N = N1 * N2 points, block-distributed in natural order: rank r holds x[r*N/P : (r+1)*N/P], i.e.
rows n1 in [r*N1/P, (r+1)*N1/P) of the row-major [N1][N2] view (n = n1*N2 + n2). Steps:

  A  all-to-all transpose      -> rank holds columns n2 of its block, all n1      [N2/P][N1]
  1  N2/P local FFTs of length N1 (HIP path)                                      Y[n2][k1]
  2  twiddle Y[n2][k1] *= W_N^(n2*k1)                                             (tfft_fourstep_twiddle)
  B  all-to-all transpose      -> rank holds k1 of its block, all n2              [N1/P][N2]
  3  N1/P local FFTs of length N2                                                 Z[k1][k2] = X[k1 + N1*k2]
  C  all-to-all transpose      -> natural order, rank holds X[r*N/P : (r+1)*N/P]  [N2/P][N1]

`input_layout="transposed"` skips A (the caller already holds [N2/P][N1]) and
`output_layout="transposed"` skips C (the result stays as Z), so a transform that may live in
the transposed distribution on both sides costs ONE all-to-all (B). Each exchange moves the
rank's whole shard once; on 8 MI355X each rank sends 1/8 of its shard to every peer over its own
xGMI link concurrently, so the exchange is per-link bound. The reference has no distributed (or
multi-GPU) code; this is new work on top of its batched entry point.

The local numerics are injected (`ops`): on a GPU they are the HIP path through the C ABI; the
CPU tests substitute the fp64 oracle to check the index logic with gloo.
"""
import torch


def split_n(n, world):
    """N = N1 * N2 with N1 <= N2, both powers of two and multiples of `world`."""
    lg = n.bit_length() - 1
    if n <= 0 or (1 << lg) != n:
        raise ValueError(f"n={n} is not a power of two")
    n1 = 1 << (lg // 2)
    n2 = n // n1
    if n1 % world or n2 % world:
        raise ValueError(f"n={n} = {n1} x {n2} does not split over {world} ranks")
    return n1, n2


class HipOps:
    """Local numerics on the GPU through libthunderfft_hip.so."""

    def __init__(self):
        import fft_amd

        self.f = fft_amd

    def fft_rows(self, t, forward):
        rows, n = t.shape[0], t.shape[1]
        self.f.ThunderFFTInitialize("fp32", n)
        return self.f.ThunderFFT_global(t, t, rows, n, forward)

    def twiddle(self, t, row0, n_total, forward):
        from fft_amd import _lib, api

        rows, cols = t.shape[0], t.shape[1]
        stream = torch.cuda.current_stream(t.device)
        with torch.cuda.device(t.device):
            api._check(_lib.load().tfft_fourstep_twiddle(0, t.data_ptr(), rows, cols, row0, 0, n_total,
                                                          int(bool(forward)), stream.cuda_stream))
        return t


def _exchange_transpose(t, world, group):
    """t: [R, C, 2] on every rank, rows = this rank's block of a [world*R, C] matrix. Returns the
    rank's block of the transposed matrix, [C/world, world*R, 2]."""
    import torch.distributed as dist

    R, C = t.shape[0], t.shape[1]
    cb = C // world
    send = t.reshape(R, world, cb, 2).permute(1, 0, 2, 3).contiguous()  # [dest][R][cb]
    if world > 1:
        recv = torch.empty_like(send)
        dist.all_to_all_single(recv, send, group=group)
    else:
        recv = send
    # recv[s] = rows of rank s, my column block -> [world*R][cb] -> transpose
    return recv.reshape(world * R, cb, 2).transpose(0, 1).contiguous()


class DistributedFFT:
    def __init__(self, n, group=None, ops=None):
        import torch.distributed as dist

        self.n = int(n)
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.n1, self.n2 = split_n(self.n, self.world)
        self.ops = ops if ops is not None else HipOps()

    def local_shape(self, layout="natural", side="input"):
        """Shape of this rank's shard: natural = contiguous run of N/P points; transposed input =
        [N2/P][N1] (columns n2 of the rank's block), transposed output = [N1/P][N2] (rows k1)."""
        P = self.world
        if layout == "natural":
            return (self.n1 // P, self.n2, 2) if side == "input" else (self.n2 // P, self.n1, 2)
        return (self.n2 // P, self.n1, 2) if side == "input" else (self.n1 // P, self.n2, 2)

    def transform(self, x_local, forward=True, input_layout="natural", output_layout="natural"):
        """x_local: float32 tensor holding this rank's shard ([..., 2] interleaved, any shape with
        the right number of points). Returns the rank's shard of the unnormalised transform."""
        P, r = self.world, self.rank
        n1, n2 = self.n1, self.n2
        if x_local.numel() != 2 * self.n // P:
            raise ValueError(f"shard holds {x_local.numel() // 2} points, expected {self.n // P}")
        if input_layout == "natural":
            y = _exchange_transpose(x_local.reshape(n1 // P, n2, 2), P, self.group)  # A: [n2/P][n1]
        else:
            y = x_local.reshape(n2 // P, n1, 2).contiguous().clone()
        y = self.ops.fft_rows(y, forward)                                   # 1: over n1
        y = self.ops.twiddle(y, r * (n2 // P), self.n, forward)             # 2: W_N^(n2*k1)
        z = _exchange_transpose(y, P, self.group)                           # B: [n1/P][n2]
        z = self.ops.fft_rows(z, forward)                                   # 3: over n2
        if output_layout == "transposed":
            return z                                                        # Z[k1][k2] = X[k1 + n1*k2]
        return _exchange_transpose(z, P, self.group)                        # C: [n2/P][n1], natural order
