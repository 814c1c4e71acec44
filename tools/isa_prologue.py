#!/usr/bin/env python3
"""Reads hipcc -S --cuda-device-only output and prints, per kernel, how its first loads are issued: L = a nontemporal
row load, l = another global load, Wk = s_waitcnt vmcnt(k), | = a basic-block label, br = a branch, from the kernel's
entry to its first LDS write / permlane / barrier. A row like "Lx14 W13 W5 W3 Lx2" is a thread whose last two loads
wait for the first ones to arrive.

    hipcc -O3 ... --offload-arch=gfx950 -S --cuda-device-only -o /tmp/k.s fft_amd/csrc/inst_f32_c.hip
    python tools/isa_prologue.py /tmp/k.s
"""
import re,sys
# per kernel: the sequence of row loads (L = nontemporal), other loads (l), vmcnt waits (Wk), labels (|) and branches
# from the kernel's entry to its first LDS write / permlane / barrier
for path in sys.argv[1:]:
    name=None; seq=[]; active=False
    def flush():
        if name and seq:
            out=[]
            for s in seq:
                if out and out[-1][0]==s: out[-1][1]+=1
                else: out.append([s,1])
            m=re.search(r'PlanILi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)ELi(\d+)E(.*?)EEENS_(\d)(\w+?)ELb(\d)ELi(\d)ELb(\d)',name)
            tag=f"N={m.group(1)} E={m.group(2)} wg={m.group(6)} {m.group(7)[:24]:24s} {m.group(9)[:9]:9s} fwd={m.group(10)} mode={m.group(11)} loop={m.group(12)}" if m else name[:100]
            print(tag+'  '+' '.join(f"{k}x{c}" if c>1 else k for k,c in out))
    for ln in open(path):
        m=re.match(r'^(_Z\w+):',ln)
        if m:
            flush(); name=m.group(1); seq=[]; active=True; continue
        if not active: continue
        t=ln.strip()
        if t.startswith('global_load') and ' nt' in t: seq.append('L')
        elif t.startswith('global_load'): seq.append('l')
        elif t.startswith('s_waitcnt') and 'vmcnt' in t: seq.append('W'+re.search(r'vmcnt\((\d+)\)',t).group(1))
        elif t.startswith('s_cbranch') or t.startswith('s_branch'): seq.append('br')
        elif t.startswith('.LBB'): seq.append('|')
        elif t.startswith('ds_write') or t.startswith('v_permlane') or t.startswith('s_barrier'):
            flush(); active=False; name=None
    flush()
