"""Condenses a tools/gpu_profile.sh run (gpurun_out/prof) into the tracked files under profiles/:

  profiles/<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats rows (kernel names shortened)
  profiles/<tag>_pmc.json           FETCH_SIZE / WRITE_SIZE per launch of the FFT kernel
  profiles/pmc_traffic.json         the record bench.py reports as roofline.traffic

HBM bytes per launch = 2 * FETCH_SIZE + WRITE_SIZE (both counters are in KiB): on gfx950
FETCH_SIZE tallies the 128-byte requests of a wide coalesced stream at 64 bytes, WRITE_SIZE is exact.

Usage: python tools/summarize_profile.py <tag> [n] [batch] [dtype] [dir]   (dtype fp32 | fp16 | bf16, default fp32;
       dir = the run's directory under gpurun_out/, default prof)

The FFT kernel of a run is the library kernel (tfft::...) with the largest total duration; multi-pass lengths run
several (one per sweep): their counters are summed per transform.
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
    if "tfft::" in name:
        return name.split("(")[0].replace("void ", "")
    return name.split("<")[0].replace("void ", "")[:80]


def is_fft(name):
    return "tfft::fft_" in name


def launches_per_transform_hint(n, fft_rows):
    return len(fft_rows) if n > 16384 else 1


def main():
    tag = sys.argv[1]
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
    batch = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
    dtype = sys.argv[4] if len(sys.argv) > 4 else "fp32"
    src = os.path.join(ROOT, "gpurun_out", sys.argv[5]) if len(sys.argv) > 5 else SRC
    bytes_per_point = 16 if dtype == "fp32" else 8
    os.makedirs(DST, exist_ok=True)
    rows = list(csv.DictReader(open(os.path.join(src, "trace", "bench_kernel_stats.csv"))))
    fft_rows = sorted((r for r in rows if is_fft(r["Name"])), key=lambda r: -float(r["TotalDurationNs"]))
    sweeps = len(fft_rows) if n > 16384 else 1  # multi-pass: every sweep is its own kernel instantiation or launch
    fft_names = {r["Name"] for r in (fft_rows if n > 16384 else fft_rows[:1])}
    with open(os.path.join(DST, f"{tag}_kernel_stats.csv"), "w", newline="") as f:
        w = csv.writer(f)
        w.writerow(["Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs", "StdDev"])
        for r in rows:
            w.writerow([short(r["Name"]), r["Calls"], r["TotalDurationNs"], r["AverageNs"], r["Percentage"], r["MinNs"],
                        r["MaxNs"], r["StdDev"]])
    pmc = {}
    for counter, sub in (("FETCH_SIZE", "pmc_fetch"), ("WRITE_SIZE", "pmc_write")):
        vals = [float(r["Counter_Value"]) for r in csv.DictReader(open(os.path.join(src, sub, "bench_counter_collection.csv")))
                if r["Kernel_Name"] in fft_names and r["Counter_Name"] == counter]
        pmc[counter] = {"launches": len(vals), "mean_kib": statistics.mean(vals), "min_kib": min(vals), "max_kib": max(vals)}
    hbm = int((2 * pmc["FETCH_SIZE"]["mean_kib"] + pmc["WRITE_SIZE"]["mean_kib"]) * 1024)
    fft = fft_rows[0]
    launches_per_transform = 1
    if n > 16384:
        # kernel launches per transform: the mean above is per launch, the record per transform
        total_calls = sum(int(r["Calls"]) for r in fft_rows)
        steps = max(1, min(int(r["Calls"]) for r in fft_rows))
        launches_per_transform = max(1, round(total_calls / steps))
        hbm *= launches_per_transform
    # The launches of bench.py's timed region: the last `steps` launches of the kernel in the trace (the warm-up
    # launches come first; 16-bit kernels are still ramping up during them).
    timed = None
    try:
        steps = None
        for text in open(os.path.join(src, "trace_stdout.log")):
            if text.startswith('{"metric"'):
                steps = json.loads(text)["steps"]
        trace = [r for r in csv.DictReader(open(os.path.join(src, "trace", "bench_kernel_trace.csv")))
                 if r["Kernel_Name"] in fft_names]
        trace.sort(key=lambda r: int(r["Start_Timestamp"]))
        if steps and len(trace) >= steps * launches_per_transform_hint(n, fft_rows):
            k = steps * launches_per_transform_hint(n, fft_rows)
            last = trace[-k:]
            dur = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in last]
            span = int(last[-1]["End_Timestamp"]) - int(last[0]["Start_Timestamp"])
            timed = {"launches": k, "mean_kernel_ns": sum(dur) / k, "first_begin_to_last_end_per_launch_ns": span / k}
    except (OSError, ValueError, KeyError):
        timed = None
    rec = {
        "kernel": short(fft["Name"]),
        "timed_region_under_rocprofv3": timed,
        "n": n,
        "batch": batch,
        "kernel_launches_per_transform": launches_per_transform,
        "avg_launch_ns_under_rocprofv3": float(fft["AverageNs"]),
        "all_fft_kernels": [{"kernel": short(r["Name"]), "calls": int(r["Calls"]), "avg_ns": float(r["AverageNs"])}
                            for r in fft_rows],
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
    # the kernel the profiled bench.py run launched, in bench.py's own words: its line is in the trace pass's log
    signature = None
    try:
        for text in open(os.path.join(src, "trace_stdout.log")):
            if text.startswith('{"metric"'):
                signature = json.loads(text)["roofline"].get("kernel_signature")
    except (OSError, ValueError, KeyError):
        pass
    table[key] = {"hbm_bytes_per_launch": hbm, "source": f"profiles/{tag}_pmc.json", "signature": signature}
    json.dump(table, open(path, "w"), indent=1)
    print(json.dumps(rec, indent=1))


if __name__ == "__main__":
    main()
