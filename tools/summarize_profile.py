"""Condenses a tools/gpu_profile.sh run (gpurun_out/prof) into the tracked files under profiles/:

  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats rows (kernel names shortened)
  profiles/<tag>_pmc.json           FETCH_SIZE / WRITE_SIZE per launch of the FFT kernel
  profiles/pmc_traffic.json         the record bench.py reports as roofline.traffic

HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (both counters are in KiB): on gfx950
FETCH_SIZE tallies the 128-byte requests of a wide coalesced stream at 64 bytes, WRITE_SIZE is exact.

Usage: python tools/summarize_profile.py <tag> [n] [batch] [dtype]     (dtype fp32 | fp16 | bf16, default fp32)
"""
import csv
import json
import os
import statistics
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "gpurun_out", "prof")
DST = os.path.join(ROOT, "profiles")


def short(name):
    if "fft_lds_kernel" in name or "fft_mfma" in name:
        return name.split("(")[0].replace("void ", "")
    return name.split("<")[0].replace("void ", "")[:80]


def main():
    tag = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
    dtype = sys.argv[4] if len(sys.argv) > 4 else "fp32"
    # fp32 runs the VALU kernel; 16-bit storage at n >= 256 the matrix-core kernels
    want = "fft_lds_kernel" if dtype == "fp32" else "fft_mfma"
    bytes_per_point = 16 if dtype == "fp32" else 8
    os.makedirs(DST, exist_ok=True)
    rows = list(csv.DictReader(open(os.path.join(SRC, "trace", "bench_kernel_stats.csv"))))
    with open(os.path.join(DST, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"],
                        r["MaxNs"], r["StdDev"]])
    pmc = {}
    for counter, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(os.path.join(SRC, sub, "bench_counter_collection.csv")))
                if want in r["Kernel_Name"] and r["Counter_Name"] == counter]
        pmc[counter] = {"launches": len(vals), "mean_kib": statistics.mean(vals), "min_kib": min(vals), "max_kib": max(vals)}
    hbm = int((2 * pmc["FETCH_SIZE"]["mean_kib"] + pmc["WRITE_SIZE"]["mean_kib"]) * 1024)
    fft = [r for r in rows if want in r["Name"]][0]
    rec = {
        "kernel": short(fft["Name"]),
        "n": n,
        "batch": batch,
        "avg_launch_ns_under_rocprofv3": float(fft["AverageNs"]),
        "counters": pmc,
        "correction": "HBM bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024; FETCH_SIZE reads half of a wide coalesced stream on gfx950",
        "hbm_bytes_per_launch": hbm,
        "algorithmic_bytes_per_launch": bytes_per_point * n * batch,
    }
    json.dump(rec, open(os.path.join(DST, f"{tag}_pmc.json"), "w"), indent=1)
    path = os.path.join(DST, "pmc_traffic.json")
    try:
        table = json.load(open(path))
    except (OSError, ValueError):
        table = {}
    key = f"fft_lds_kernel_f32_n{n}_b{batch}" if dtype == "fp32" else f"fft_kernel_{dtype}_n{n}_b{batch}"
    table[key] = {"hbm_bytes_per_launch": hbm, "source": f"profiles/{tag}_pmc.json"}
    json.dump(table, open(path, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
