"""Tabulate a tools/gpu_sweep*.sh log: best %HBM per (N, mode, in place / out of place)."""
import collections
import re
import sys

runs = collections.defaultdict(list)
key = None
for line in open(sys.argv[1]):
    if line.startswith("###"):
        m = re.search(r"--mode (\d+)", line)
        key = ("m" + (m.group(1) if m else "0")) + ("/oop" if "--inplace 0" in line else "/ip")
    if line.startswith("thunderfft"):
        tok = line.split()
        runs[(int(tok[1]), key)].append(float(tok[-1].rstrip("%")))
ns = sorted({k[0] for k in runs})
keys = sorted({k[1] for k in runs})
print("N     ", *[f"{k:>8s}" for k in keys])
for n in ns:
    print(f"{n:6d}", *[f"{max(runs[(n, k)]):8.1f}" if runs[(n, k)] else "       -" for k in keys])
