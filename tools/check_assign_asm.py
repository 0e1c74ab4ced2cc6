#!/usr/bin/env python3
"""Build-time check of the E-step kernel's early label load (csrc/assign_kernel.h).

k_assign_mfma requests the previous labels with an inline-asm `global_load_dword` at its very start and only
consumes the register in the epilogue.  The compiler cannot see that the load is in flight: if it read, copied,
spilled or re-used the destination VGPR before the load has landed, the kernel would silently count wrong
"labels changed" values (which decide strict convergence).  The load has landed at the first
`s_waitcnt vmcnt(0)` (dma_wait() in front of the first stage barrier).  This script disassembles every
k_assign_mfma instantiation of assign.hip and fails if the destination register appears in any instruction
between the load and that first vmcnt(0).

    python tools/check_assign_asm.py <device assembly (.s) of assign.hip>   # made by csrc/Makefile
"""
import re
import sys


def uses_reg(line, reg):
    """does the instruction text mention VGPR `reg` (v57, or inside a range v[56:59])?"""
    body = line.split(";")[0]
    for m in re.finditer(r"\bv(\d+)\b", body):
        if int(m.group(1)) == reg:
            return True
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", body):
        if int(m.group(1)) <= reg <= int(m.group(2)):
            return True
    return False


def check(path):
    lines = open(path).read().splitlines()
    kernels = {}
    cur = None
    for ln in lines:
        m = re.match(r"^(_Z13k_assign_mfmaI\w*):", ln)  # k_assign_mfma<...> only: the grouped kernel loads its labels plainly
        if m:
            cur = m.group(1)
            kernels[cur] = []
            continue
        if re.match(r"^\s*\.end_amdhsa_kernel|^\.Lfunc_end", ln):
            cur = None
        if cur is not None:
            kernels[cur].append(ln)
    if not kernels:
        print(f"check_assign_asm: no k_assign_mfma instantiation found in {path}", file=sys.stderr)
        return 1
    bad = 0
    for name, body in kernels.items():
        loads = [i for i, ln in enumerate(body) if re.search(r"\bglobal_load_dword\s+v\d+\s*,", ln)]
        if len(loads) != 1:
            print(f"check_assign_asm: {name}: expected exactly one global_load_dword (the label load), found {len(loads)}",
                  file=sys.stderr)
            bad += 1
            continue
        i0 = loads[0]
        reg = int(re.search(r"global_load_dword\s+v(\d+)", body[i0]).group(1))
        landed = None
        for i in range(i0 + 1, len(body)):
            ln = body[i].split(";")[0]
            if re.search(r"\bs_waitcnt\b.*vmcnt\(0\)", ln):
                landed = i
                break
            if re.search(r"\bs_(c)?branch|\bs_endpgm|\bs_setpc", ln):
                print(f"check_assign_asm: {name}: control flow before the first vmcnt(0) after the label load: {ln.strip()}",
                      file=sys.stderr)
                bad += 1
                landed = -1
                break
            if uses_reg(ln, reg):
                print(f"check_assign_asm: {name}: v{reg} (label load in flight) is touched before vmcnt(0): {ln.strip()}",
                      file=sys.stderr)
                bad += 1
                landed = -1
                break
        if landed is None:
            print(f"check_assign_asm: {name}: no s_waitcnt vmcnt(0) after the label load", file=sys.stderr)
            bad += 1
    if bad == 0:
        print(f"check_assign_asm: {len(kernels)} k_assign_mfma instantiations OK")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(check(sys.argv[1]))
