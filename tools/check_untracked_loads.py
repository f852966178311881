#!/usr/bin/env python3
"""ISA check for the untracked (inline-asm) loads of the packed chain kernels: between such a load and the SECOND
counted drain after it (asm `s_waitcnt vmcnt`), no instruction may read or overwrite its destination registers --
hipcc does not know the data is still in flight, so a copy / address computation scheduled there would use stale
registers.  Linear scan over the kernel's text (blocks are laid out in program order).
  hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -S --cuda-device-only -o /tmp/mgn.s graph-physics_amd/csrc/mgn_kernels.hip
  python tools/check_untracked_loads.py /tmp/mgn.s"""
import re, sys
text = open(sys.argv[1]).read().splitlines()
kern = None; body = {}
for ln in text:
    m = re.match(r"^(_Z\w*k_(mlp_fwd_x6|mlp_bwd_x6|wgrad_x6)\w*):", ln)
    if m: kern = m.group(1); body[kern] = []; continue
    if kern is not None:
        body[kern].append(ln)
        if "s_endpgm" in ln: kern = None
def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m: return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()
bad = 0; total = 0
for k, lines in body.items():
    in_asm = False
    i = 0
    loads = []  # (line index, dest regs)
    for i, ln in enumerate(lines):
        if "#ASMSTART" in ln: in_asm = True; continue
        if "#ASMEND" in ln: in_asm = False; continue
        if in_asm:
            m = re.match(r"\s*global_load_dword(x4)? (\S+), ", ln)
            if m and "lds" not in ln: loads.append((i, regs(m.group(2).rstrip(","))))
    for (i0, dst) in loads:
        total += 1
        drains = 0; in_asm = False
        for j in range(i0 + 1, len(lines)):
            ln = lines[j]
            if "#ASMSTART" in ln: in_asm = True; continue
            if "#ASMEND" in ln: in_asm = False; continue
            if in_asm and "s_waitcnt vmcnt" in ln:
                drains += 1
                if drains >= 2: break
                continue
            code = ln.split(";")[0].strip()
            if not code or code.endswith(":") or code.startswith("."): continue
            toks = re.findall(r"v\[\d+:\d+\]|v\d+", code)
            used = set().union(*[regs(t) for t in toks]) if toks else set()
            if used & dst:
                if in_asm and re.match(r"global_load_dword", code): continue  # a later untracked load into the same home
                print("%s: line +%d touches %s of the load at +%d before two drains: %s" % (k[:40], j, sorted(used & dst)[:4], i0, code))
                bad += 1
                break
print("%d untracked loads checked, %d violations" % (total, bad))
sys.exit(1 if bad else 0)
