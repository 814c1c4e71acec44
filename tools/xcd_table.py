"""Mean %HBM per (N, L) from the logs of an XCD run-length sweep (tools/gpu_ab.sh with XCDS="0 2 4 6"). Usage: python tools/xcd_table.py gpurun_out/xcd*.log"""
import collections
import re
import sys

for f in sys.argv[1:]:
    d = collections.defaultdict(list)
    for line in open(f):
        m = re.match(r"rep=(\d) L=(-?\d+) (.*)", line)
        if not m:
            continue
        toks = m[3].split()
        for n, v in zip(toks[0::2], toks[1::2]):
            d[(int(n), int(m[2]))].append(float(v.rstrip("%")))
    ns = sorted({k[0] for k in d})
    ls = sorted({k[1] for k in d})
    print(f)
    print("N     " + " ".join(f"L={L:<5d}" for L in ls))
    for n in ns:
        print(f"{n:<6d}" + " ".join(f"{sum(d[(n, L)]) / len(d[(n, L)]):7.1f}" for L in ls))
