"""Test infrastructure, not product code: the local passes of the distributed transform evaluated in fp64 on the
CPU straight from their definitions (include/thunderfft_c.h: tfft_dist_permute, tfft_dist_radix). The gloo tests
run fft_amd.distributed.DistributedFFT with this object in place of HipOps to check the pass descriptions and
index maps; the GPU tests compare the HIP passes with it element by element."""
import numpy as np
import torch


class NumpyOps:
    """Buffers are float64 [points, 2] CPU tensors."""

    def __init__(self, fft=None):
        self._fft = fft  # callable(x[rows, n] complex128, forward) -> complex128; default numpy.fft

    def empty(self, points):
        return torch.zeros(points, 2, dtype=torch.float64)

    def prepare(self, fft_lengths, root_tables, two_level_tables):
        pass

    @staticmethod
    def _c(buf):
        a = buf.numpy()
        return a[:, 0] + 1j * a[:, 1]

    @staticmethod
    def _put(buf, z, idx=None):
        a = buf.numpy()
        if idx is None:
            a[:, 0], a[:, 1] = z.real, z.imag
        else:
            a[idx, 0], a[idx, 1] = z.real, z.imag

    def fft_rows(self, buf, rows, n, forward):
        x = self._c(buf)[:rows * n].reshape(rows, n)
        if self._fft is not None:
            y = self._fft(x, forward)
        else:
            y = np.fft.fft(x, axis=-1) if forward else np.fft.ifft(x, axis=-1) * n
        self._put(buf, np.asarray(y).reshape(-1), np.arange(rows * n))

    def permute(self, src, dst, spec, forward):
        b1, b2, i, j = np.meshgrid(*(np.arange(spec[k], dtype=np.int64) for k in ("nb1", "nb2", "ni", "nj")),
                                   indexing="ij")
        src_idx = b1 * spec["in_b1"] + b2 * spec["in_b2"] + i * spec["in_i"] + j
        if spec["transpose"]:
            dst_idx = b1 * spec["out_b1"] + b2 * spec["out_b2"] + j * spec["out_j"] + i
        else:
            dst_idx = b1 * spec["out_b1"] + b2 * spec["out_b2"] + i * spec["out_j"] + j
        z = self._c(src)[src_idx]
        tw = spec["twiddle"]
        if tw:
            p = tw["p0"] + tw["pi"] * i
            q = tw["q0"] + tw["qb1"] * b1 + tw["qb2"] * b2 + tw["qj"] * j
            e = (p * q) % tw["n"]
            z = z * np.exp((-2j if forward else 2j) * np.pi * e / tw["n"])
        self._put(dst, z.reshape(-1), dst_idx.reshape(-1))

    def radix(self, src, dst, spec, forward):
        P, rows, T = spec["parts"], spec["rows"], spec["row_len"]
        s, r, t = np.meshgrid(np.arange(P, dtype=np.int64), np.arange(rows, dtype=np.int64),
                              np.arange(T, dtype=np.int64), indexing="ij")
        x = self._c(src)[s * spec["in_chunk"] + r * spec["in_pitch"] + t]  # [s][r][t]
        sign = -2j if forward else 2j
        k = np.arange(P, dtype=np.int64)
        dft = np.exp(sign * np.pi * np.outer(k, k) / P)  # [k][s]
        y = np.einsum("ks,srt->krt", dft, x) * np.exp(sign * np.pi * (k[:, None, None] * t[:1]) / (P * T))
        self._put(dst, y.reshape(-1), (s * spec["out_chunk"] + r * spec["out_pitch"] + t).reshape(-1))
