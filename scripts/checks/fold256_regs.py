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
    kern, bad, seen, n_ix = None, 0, 0, 0
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
        # gemm256s_kernel's load_ix: two dword loads of the 2:4 positions, the second at offset:256 of the same address -- same rule: nothing names the two
        # destination registers before a counted vmcnt wait (take_ix's v_mov sits behind it; in text order the next iteration's wait or the last K-tile's)
        b2 = [x.strip() for x in lines[i:i + 2]]
        if (len(b2) == 2 and b2[0].startswith("global_load_dword ") and b2[1].startswith("global_load_dword ") and b2[1].endswith("offset:256")
                and "gemm256s" in (kern or "") and b2[0].split()[2] == b2[1].split()[2]):
            dest = regs_of(b2[0].split()[1].rstrip(",")) | regs_of(b2[1].split()[1].rstrip(","))
            waited, j = False, i + 2
            while j < len(lines):
                t = lines[j].strip()
                if t.startswith("s_waitcnt") and "vmcnt" in t:
                    waited = True
                if t.startswith("s_endpgm"):
                    break
                if not t.startswith(";") and not t.startswith(".") and all_vregs(t) & dest:
                    break
                j += 1
            n_ix += 1
            if not waited:
                bad += 1
                print(f"BAD {kern}: load_ix registers {sorted(dest)} touched by `{lines[j].strip()}` before any counted wait")
            i += 2
            continue
        i += 1
    print(f"{n_ix} load_ix sites checked in the gemm256s instantiations")
    print(f"{seen} fold256 prologue(s) checked, {bad} bad")
    return 1 if bad or not seen else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1]))
