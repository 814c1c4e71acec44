"""Per-kernel instruction mix of a gfx950 assembly dump (hipcc -save-temps *.s).

Usage: python tools/isa_stats.py file.s [substring-of-kernel-name]
"""
import collections
import re
import sys


def main():
    text = open(sys.argv[1]).read()
    want = sys.argv[2] if len(sys.argv) > 2 else ""
    labels = [(m.start(), m.group(1)) for m in re.finditer(r"^(_Z\w+):", text, re.M)]
    for idx, (pos, name) in enumerate(labels):
        if want and want not in name:
            continue
        end = text.find("s_endpgm", pos)
        body = text[pos:end]
        ops = collections.Counter()
        for line in body.split("\n")[1:]:
            line = line.strip()
            if not line or line[0] in ".;/" or line.endswith(":"):
                continue
            ops[line.split()[0]] += 1
        cat = collections.Counter()
        for op, c in ops.items():
            if op.startswith("v_pk_"):
                cat["v_pk"] += c
            elif op.startswith("v_"):
                cat["valu"] += c
            elif op.startswith("s_waitcnt"):
                cat["waitcnt"] += c
            elif op.startswith("s_barrier"):
                cat["barrier"] += c
            elif op.startswith("s_"):
                cat["salu"] += c
            elif op.startswith("ds_"):
                cat["ds"] += c
            elif op.startswith(("global_", "buffer_", "flat_", "scratch_")):
                cat["vmem"] += c
            else:
                cat["other"] += c
        print(name[:110])
        print("   total", sum(ops.values()), dict(cat))
        print("   ", ops.most_common(16))


if __name__ == "__main__":
    main()
