"""Bank-conflict search for the LDS images of the matrix-core N = 4096 kernel (fft_mfma_kernel.hpp).

Images hold one dword (fp16 re, im) per point: 256 rows (the 256 columns of a radix-16 pass) of 16
points (the contracted axis), row pitch P dwords, the four 16-byte granules of a row rotated by
rot(row). Writers store single dwords (ds_write_b32: 2 groups of 32 lanes, 32 banks; 2-way is free),
readers fetch one granule (ds_read_b128: 4 groups of 16 lanes in the hardware's lane order, 64 banks).
"""
import itertools

B128_GROUPS = [
    list(range(0, 4)) + list(range(12, 16)) + list(range(20, 28)),
    list(range(4, 12)) + list(range(16, 20)) + list(range(28, 32)),
    list(range(32, 36)) + list(range(44, 48)) + list(range(52, 60)),
    list(range(36, 44)) + list(range(48, 52)) + list(range(60, 64)),
]


def cost_b32(addrs):  # addrs: 64 dword addresses
    tot = 0
    for grp in (range(0, 32), range(32, 64)):
        banks = {}
        for l in grp:
            banks.setdefault(addrs[l] % 32, set()).add(addrs[l])
        tot += max(len(s) for s in banks.values())
    return tot / 2.0


def cost_b128(addrs):  # addrs: 64 dword addresses of 16-byte reads
    tot = 0
    for grp in B128_GROUPS:
        banks = {}
        for l in grp:
            for d in range(4):
                banks.setdefault((addrs[l] + d) % 64, set()).add(addrs[l] + d)
        tot += max(len(s) for s in banks.values())
    return tot / 4.0


def slot(row, n, P, rot):
    return row * P + ((n + 4 * rot(row)) & 15)


def evaluate(P, rot2, rot3):
    res = {}
    # W1: P1 epilogue, lane (g, i) of wave w writes (k1 = 4g + r, n3 = 4(i&3) + j, n2 = 4w + (i>>2))
    worst = 0
    for w, r, j in itertools.product(range(4), range(4), range(4)):
        a = [slot((4 * (l >> 4) + r) * 16 + 4 * (l & 3) + j, 4 * w + ((l & 15) >> 2), P, rot2) for l in range(64)]
        worst = max(worst, cost_b32(a))
    res["W1"] = worst
    # R2: lane (g, i) reads granule g of row (4w + t) * 16 + i
    worst = 0
    for w, t in itertools.product(range(4), range(4)):
        a = [slot((4 * w + t) * 16 + (l & 15), 4 * (l >> 4), P, rot2) for l in range(64)]
        worst = max(worst, cost_b128(a))
    res["R2"] = worst
    # W2: lane (g, i) writes (row = (4g + r) * 16 + 4w + t, n3 = i)
    worst = 0
    for w, t, r in itertools.product(range(4), range(4), range(4)):
        a = [slot((4 * (l >> 4) + r) * 16 + 4 * w + t, l & 15, P, rot3) for l in range(64)]
        worst = max(worst, cost_b32(a))
    res["W2"] = worst
    worst = 0
    for w, t in itertools.product(range(4), range(4)):
        a = [slot((4 * w + t) * 16 + (l & 15), 4 * (l >> 4), P, rot3) for l in range(64)]
        worst = max(worst, cost_b128(a))
    res["R3"] = worst
    return res


def main():
    rots = {
        "0": lambda row: 0,
        "row>>2": lambda row: (row >> 2) & 3,
        "row>>4": lambda row: (row >> 4) & 3,
        "row>>6": lambda row: (row >> 6) & 3,
        "row": lambda row: row & 3,
        "(row>>4)+(row>>6)": lambda row: ((row >> 4) + (row >> 6)) & 3,
        "(row>>2)+(row>>6)": lambda row: ((row >> 2) + (row >> 6)) & 3,
        "row+(row>>2)": lambda row: (row + (row >> 2)) & 3,
        "row+(row>>4)": lambda row: (row + (row >> 4)) & 3,
        "row+(row>>6)": lambda row: (row + (row >> 6)) & 3,
        "(row>>1)": lambda row: (row >> 1) & 3,
        "(row>>3)": lambda row: (row >> 3) & 3,
        "(row>>5)": lambda row: (row >> 5) & 3,
    }
    for P in (16, 20, 24, 28):
        best2 = best3 = None
        for name, rot in rots.items():
            r = evaluate(P, rot, rot)
            s2 = (max(r["W1"], 2.0) + 2 * r["R2"], name, r["W1"], r["R2"])
            s3 = (max(r["W2"], 2.0) + 2 * r["R3"], name, r["W2"], r["R3"])
            best2 = min(best2, s2) if best2 else s2
            best3 = min(best3, s3) if best3 else s3
        print(f"P={P}: image 2 rot={best2[1]} write x{best2[2]:.2f} read x{best2[3]:.2f} | "
              f"image 3 rot={best3[1]} write x{best3[2]:.2f} read x{best3[3]:.2f}")
    # output image: natural order k with pad(k) = k + (k >> 8) * PW
    for PW in (0, 4, 8, 16, 20):
        worst = 0
        for w, t, r in itertools.product(range(4), range(4), range(4)):
            a = []
            for l in range(64):
                k = (l & 15) + 16 * (4 * w + t) + 256 * (4 * (l >> 4) + r)
                a.append(k + (k >> 8) * PW)
            worst = max(worst, cost_b32(a))
        rd = 0
        for q, w in itertools.product(range(4), range(4)):
            a = []
            for l in range(64):
                k = 4 * (64 * w + l + 256 * q)
                a.append(k + (k >> 8) * PW)
            rd = max(rd, cost_b128(a))
        print(f"output image pad {PW} dwords per 256: write x{worst:.2f} read x{rd:.2f}")


if __name__ == "__main__":
    main()
