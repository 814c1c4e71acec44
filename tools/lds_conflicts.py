"""LDS bank-conflict model for the in-LDS exchange stages of the batched FFT kernels.

Mirrors the index arithmetic of fft_amd/csrc/fft_lds_kernel.hpp (first / middle / last pass
butterfly -> logical exchange index -> padded physical index) and prices every
ds_write_b64 / ds_read_b64 wave-instruction with the gfx950 banking rules:

  ds_write_b64 : 4 groups of 16 consecutive lanes, bank = (byte/4) % 32
  ds_read_b64  : 2 groups of 32 consecutive lanes, bank = (byte/4) % 64

A group costs max-over-banks(number of distinct 8-byte words on that bank) LDS cycles.
The script searches the one-level padding  phys(i) = i + (i >> PS) * PW  per exchange and
prints the cheapest conflict-free (or least-conflicting) choice with the smallest LDS size.

Usage: python tools/lds_conflicts.py            # table for every shipped configuration
       python tools/lds_conflicts.py --split    # same for the split (real, then imaginary) exchange
"""
import itertools
import sys


def swz(tid):
    """Virtual thread id that puts lanes l and l+32 of a wave on adjacent butterflies."""
    return (tid & ~63) | ((tid & 31) << 1) | ((tid >> 5) & 1)


SPLIT = False  # True: exchanges move 4-byte halves (ds_write_b32 / ds_read_b32, 2 x 32 lanes, 32 banks)


def group_cost(addrs, nbanks):
    """addrs: element indices of the lanes of one lane group (8-byte elements, 4-byte when SPLIT)."""
    per_bank = {}
    for a in addrs:
        dw = a if SPLIT else 2 * a  # an 8-byte word covers banks dw, dw+1
        per_bank.setdefault(dw % nbanks, set()).add(a)
    return max(len(s) for s in per_bank.values())


def instr_cost(addrs64, kind):
    if SPLIT:
        groups = [addrs64[g * 32:(g + 1) * 32] for g in range(2)]
        return sum(group_cost(g, 32) for g in groups), 2
    if kind == "w":
        groups = [addrs64[g * 16:(g + 1) * 16] for g in range(4)]
        return sum(group_cost(g, 32) for g in groups), 4
    groups = [addrs64[g * 32:(g + 1) * 32] for g in range(2)]
    return sum(group_cost(g, 64) for g in groups), 2


class Cfg:
    def __init__(self, N, E, R, WG):
        self.N, self.E, self.R, self.WG = N, E, R, WG
        self.T = N // E
        self.TPB = max(1, WG // self.T)
        assert self.T * self.TPB == WG, (N, E, WG)
        self.P = len(R)

    # butterfly ids handled by thread `tid` in a pass with NB butterflies per thread
    def bfly_ids(self, tid, NB, paired):
        if NB == 1:
            return [swz(tid) if paired else tid]
        if paired:
            out = []
            for pi in range(NB // 2):
                g = 2 * (tid + self.WG * pi)
                out += [g, g + 1]
            return out
        return [tid + self.WG * b for b in range(NB)]

    # ---- exchange after the first pass: writer = first pass, reader = next pass
    def ex1_write(self, tid):
        N, R1 = self.N, self.R[0]
        C1 = N // R1
        out = []
        for g in self.bfly_ids(tid, self.E // R1, True):
            tr, c = divmod(g, C1)
            out.append([tr * N + k1 * C1 + c for k1 in range(R1)])
        return out  # list over butterflies of list over k1

    def ex1_read(self, tid):
        N, R1, R2 = self.N, self.R[0], self.R[1]
        C1 = N // R1
        if self.P == 2:
            # last pass: butterfly q = k1, elements n2 (stride 1)
            out = []
            for g in self.bfly_ids(tid, self.E // R2, True):
                tr, k1 = divmod(g, R1)
                out.append([tr * N + k1 * C1 + n2 for n2 in range(R2)])
            return out
        R3 = self.R[2]
        C2 = N // R2
        out = []
        for g in self.bfly_ids(tid, self.E // R2, False):
            tr, q = divmod(g, C2)
            k1, n3 = divmod(q, R3)
            out.append([tr * N + k1 * C1 + n2 * R3 + n3 for n2 in range(R2)])
        return out

    # ---- exchange after the middle pass (3-pass plans)
    def ex2_write(self, tid):
        N, R1, R2, R3 = self.N, *self.R
        C2 = N // R2
        out = []
        for g in self.bfly_ids(tid, self.E // R2, False):
            tr, q = divmod(g, C2)
            k1, n3 = divmod(q, R3)
            out.append([tr * N + k2 * (R1 * R3) + k1 * R3 + n3 for k2 in range(R2)])
        return out

    def ex2_read(self, tid):
        N, R1, R2, R3 = self.N, *self.R
        C3 = N // R3
        out = []
        for g in self.bfly_ids(tid, self.E // R3, True):
            tr, q3 = divmod(g, C3)
            k2, k1 = divmod(q3, R1)
            out.append([tr * N + k2 * (R1 * R3) + k1 * R3 + n3 for n3 in range(R3)])
        return out


def price(cfg, fn, kind, pad):
    """Sum the cost of every wave-instruction of one access phase for one workgroup."""
    per_thread = [fn(t) for t in range(cfg.WG)]
    nb = len(per_thread[0])
    ne = len(per_thread[0][0])
    tot = base = 0
    for w in range(cfg.WG // 64):
        for b in range(nb):
            for e in range(ne):
                addrs = [pad(per_thread[w * 64 + l][b][e]) for l in range(64)]
                c, b0 = instr_cost(addrs, kind)
                tot += c
                base += b0
    return tot, base


def make_pad(ps, pw):
    if pw == 0:
        return lambda i: i
    return lambda i: i + (i >> ps) * pw


def search(cfg, wfn, rfn, label):
    best = None
    total = cfg.N * cfg.TPB
    cands = [(0, 0)] + [(ps, pw) for ps in range(2, 12) for pw in (1, 2, 4, 8, 16)]
    for ps, pw in cands:
        pad = make_pad(ps, pw)
        size = pad(total - 1) + 1
        if size * (4 if SPLIT else 8) > 160 * 1024:
            continue
        wc, wb = price(cfg, wfn, "w", pad)
        rc, rb = price(cfg, rfn, "r", pad)
        # write conflicts only cost once the LDS-array cycles exceed the ~6-cycle issue cost
        score = (rc / rb) + 0.6 * max(wc / wb, 1.5) + 1e-3 * size / total
        rec = (score, ps, pw, size, wc / wb, rc / rb)
        if best is None or rec < best:
            best = rec
    _, ps, pw, size, wf, rf = best
    print(f"  {label}: PS={ps} PW={pw} lds_elems={size} ({size * (4 if SPLIT else 8)} B)  write x{wf:.2f}  read x{rf:.2f}")
    return ps, pw, size


CONFIGS = [
    (64, 16, (4, 4, 4), 256),
    (128, 16, (4, 8, 4), 256),
    (256, 16, (4, 16, 4), 256),
    (512, 16, (8, 16, 4), 256),
    (64, 16, (16, 4), 256),
    (128, 16, (16, 8), 256),
    (256, 16, (16, 16), 256),
    (512, 16, (16, 8, 4), 256),
    (1024, 16, (16, 16, 4), 256),
    (2048, 16, (16, 16, 8), 256),
    (4096, 16, (16, 16, 16), 256),
    (8192, 32, (32, 16, 16), 256),
    (16384, 32, (32, 32, 16), 512),
]


def main():
    global SPLIT
    SPLIT = "--split" in sys.argv
    for N, E, R, WG in CONFIGS:
        cfg = Cfg(N, E, R, WG)
        print(f"N={N} E={E} R={R} WG={WG} TPB={cfg.TPB}")
        search(cfg, cfg.ex1_write, cfg.ex1_read, "exchange 1")
        if cfg.P == 3:
            search(cfg, cfg.ex2_write, cfg.ex2_read, "exchange 2")


if __name__ == "__main__" and "--tile" not in sys.argv:
    main()


# ---------------------------------------------------------------------------------------------
# Tile kernels (fft_tile_kernel.hpp): two passes, LDS layout [element][transform]
# ---------------------------------------------------------------------------------------------
class TileCfg:
    def __init__(self, N, E, R1, R2, TPB):
        self.N, self.E, self.R1, self.R2, self.TPB = N, E, R1, R2, TPB
        self.WG = TPB * N // E
        self.C1, self.CL = N // R1, N // R2

    def ids(self, tid, NB):
        if NB == 1:
            return [swz(tid)]
        out = []
        for pi in range(NB // 2):
            g = 2 * (tid + self.WG * pi)
            out += [g, g + 1]
        return out

    def write(self, tid, in_col):
        out = []
        for g in self.ids(tid, self.E // self.R1):
            tr, c = (g % self.TPB, g // self.TPB) if in_col else (g // self.C1, g % self.C1)
            out.append([(k1 * self.C1 + c) * self.TPB + tr for k1 in range(self.R1)])
        return out

    def read(self, tid, out_col):
        out = []
        for g in self.ids(tid, self.E // self.R2):
            tr, q = (g % self.TPB, g // self.TPB) if out_col else (g // self.CL, g % self.CL)
            out.append([(q * self.C1 + n2) * self.TPB + tr for n2 in range(self.R2)])
        return out


def price_tile(cfg, per_thread, kind, pad):
    nb, ne = len(per_thread[0]), len(per_thread[0][0])
    tot = base = 0
    for w in range(cfg.WG // 64):
        for b in range(nb):
            for e in range(ne):
                c, b0 = instr_cost([pad(per_thread[w * 64 + l][b][e]) for l in range(64)], kind)
                tot += c
                base += b0
    return tot / base


TILE_CONFIGS = [
    (64, 16, 16, 4, 128),
    (128, 16, 16, 8, 64),
    (64, 16, 16, 4, 64),
    (128, 16, 16, 8, 32),
    (256, 16, 16, 16, 16),
    (256, 16, 16, 16, 32),
    (512, 32, 16, 32, 16),
    (1024, 32, 32, 32, 16),
    (512, 32, 16, 32, 32),
    (256, 32, 16, 16, 64),
]


def tile_main():
    global SPLIT
    for split in (False, True):
        SPLIT = split
        print(f"==== tile kernels, {'split' if split else 'full'} exchange")
        for N, E, R1, R2, TPB in TILE_CONFIGS:
            cfg = TileCfg(N, E, R1, R2, TPB)
            total = N * TPB
            w_col = [cfg.write(t, True) for t in range(cfg.WG)]
            w_row = [cfg.write(t, False) for t in range(cfg.WG)]
            r_col = [cfg.read(t, True) for t in range(cfg.WG)]
            best = None
            for ps, pw in [(0, 0)] + [(ps, pw) for ps in range(2, 12) for pw in (1, 2, 4, 8, 16)]:
                pad = make_pad(ps, pw)
                size = pad(total - 1) + 1
                if size * (4 if split else 8) > 160 * 1024:
                    continue
                wc, wr, rc = (price_tile(cfg, w_col, "w", pad), price_tile(cfg, w_row, "w", pad),
                              price_tile(cfg, r_col, "r", pad))
                score = 2 * rc + 0.6 * max(wc, 1.5) + 0.6 * max(wr, 1.5) + 1e-3 * size / total
                rec = (score, ps, pw, size, wc, wr, rc)
                if best is None or rec < best:
                    best = rec
            _, ps, pw, size, wc, wr, rc = best
            print(f"N={N} E={E} R=({R1},{R2}) TPB={TPB} WG={cfg.WG}: PS={ps} PW={pw} "
                  f"lds={size * (4 if split else 8)} B  write(col-in) x{wc:.2f}  write(row-in) x{wr:.2f}  read(col-out) x{rc:.2f}")


if __name__ == "__main__" and "--tile" in sys.argv:
    tile_main()
