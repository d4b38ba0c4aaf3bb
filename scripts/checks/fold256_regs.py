#!/usr/bin/env python3
"""Listing check for the fold256 consumer prologue (must3r_amd/csrc/gemm.hip fold256_issue / fold256_landed): the statistics are requested by inline-asm loads whose
destination registers are only valid once the prologue's counted `s_waitcnt vmcnt(N)` has retired them; the compiler does not know that.  For every kernel that holds the
six-load block this walks the listing from the block to the first instruction that names one of its destination registers and checks that (a) a counted vmcnt wait lies in
between, (b) nothing in between writes, copies or spills one of them.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off --cuda-device-only -S must3r_amd/csrc/gemm.hip -o /tmp/gemm.s && python scripts/checks/fold256_regs.py /tmp/gemm.s
"""
import re
import sys


def regs_of(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def all_vregs(line):
    out = set()
    for tok in re.findall(r"v\[\d+:\d+\]|v\d+", line):
        out |= regs_of(tok)
    return out


def main(path):
    lines = open(path).read().split("\n")
    kern, bad, seen = None, 0, 0
    i = 0
    while i < len(lines):
        l = lines[i]
        if l.startswith("_ZN3m3r"):
            kern = l.split(":")[0]
        block = [x.strip() for x in lines[i:i + 6]]
        if (len(block) == 6 and all(b.startswith("global_load_dwordx4 ") for b in block[:4]) and block[4].startswith("global_load_dword ")
                and block[5].startswith("global_load_dwordx4 ") and "off" in block[0]):
            dest = set()
            for b in block:
                dest |= regs_of(b.split()[1].rstrip(","))
            seen += 1
            waited, j = False, i + 6
            while j < len(lines):
                t = lines[j].strip()
                if t.startswith("s_waitcnt") and "vmcnt" in t:
                    waited = True
                if t.startswith("s_endpgm"):
                    break
                if not t.startswith(";") and not t.startswith(".") and all_vregs(t) & dest:
                    break
                j += 1
            first = lines[j].strip() if j < len(lines) else "(none)"
            ok = waited and not first.startswith(("scratch_store", "v_mov", "v_accvgpr_write"))
            print(f"{'ok ' if ok else 'BAD'} {kern}: {len(dest)} registers, first touched {j - i - 6} lines later by `{first}` (counted wait in between: {waited})")
            bad += 0 if ok else 1
            i += 6
            continue
        i += 1
    print(f"{seen} fold256 prologue(s) checked, {bad} bad")
    return 1 if bad or not seen else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
