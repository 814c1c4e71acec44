"""Registers, spills, LDS and occupancy of every kernel in one .hip file, from the compiler's own remarks.

Usage: python tools/kernel_resources.py fft_amd/csrc/inst_f32_a.hip [substring-of-demangled-plan]
"""
import re
import subprocess
import sys

FLAGS = ["-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "--offload-arch=gfx950",
         "-Rpass-analysis=kernel-resource-usage", "-c", "-o", "/dev/null"]


def short(name):
    m = re.search(r"PlanI((?:Li\d+E)+)E", name)
    plan = ",".join(re.findall(r"Li(\d+)E", m.group(1))) if m else "?"
    io = "f32" if "StoreF32" in name else "f16" if "StoreF16" in name else "bf16" if "StoreBF16" in name else "?"
    tail = re.search(r"Store\w+?ELb([01])ELi(\d+)E", name)
    kind = "tile" if "fft_tile_kernel" in name else "lds"
    if kind == "tile":
        tail = re.search(r"ELb([01])ELi(\d+)EEEv", name)
    fwd, mode = (tail.group(1), tail.group(2)) if tail else ("?", "?")
    return f"{kind} <{plan}> {io} fwd={fwd} mode={mode}"


def main():
    src = sys.argv[1]
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    out = subprocess.run(["/opt/rocm/bin/hipcc", *FLAGS, src], capture_output=True, text=True).stderr
    cur = {}
    rows = []
    for line in out.split("\n"):
        m = re.search(r"remark: (?:\s*)([A-Za-z ]+\[?[A-Za-z/]*\]?): (.*?) \[-Rpass", line)
        if not m:
            continue
        key, val = m.group(1).strip(), m.group(2).strip()
        if key == "Function Name":
            cur = {"name": val}
            rows.append(cur)
        else:
            cur[key] = val
    for r in rows:
        s = short(r["name"])
        if want and want not in s:
            continue
        print(f"{s:70s} vgpr={r.get('VGPRs', '?'):>4} agpr={r.get('AGPRs', '?'):>3} "
              f"spill={r.get('VGPR Spill', '?'):>3} scratch={r.get('ScratchSize [bytes/lane]', '?'):>4} "
              f"occ={r.get('Occupancy [waves/SIMD]', '?'):>2} lds={r.get('LDS Size [bytes/block]', '?')}")


if __name__ == "__main__":
    main()
