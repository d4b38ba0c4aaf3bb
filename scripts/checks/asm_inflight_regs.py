"""Static check of hand-counted LDS waits in the generated assembly (build-time helper, not part of the product).

Kernels that issue their fragment reads from inline asm (gemm256k_kernel, attn4_kernel; first written for the r03 gemm256s_kernel experiment) count `lgkmcnt` by hand.  The property to hold on EVERY
path through the kernel: between a `ds_read_b128 vDST, ...` and an `s_waitcnt lgkmcnt(N)` that guarantees its arrival, no
instruction reads or writes a register of vDST (the register allocator may otherwise copy a fragment -- at a join, for a tied asm
operand -- before the data has landed, or reuse the register).  LDS operations return in order: a read has arrived once a wait with
N <= (number of younger LDS reads in flight) has executed.

The check is a forward dataflow walk over the kernel's control-flow graph (basic blocks from the labels and s_branch / s_cbranch
instructions of the .s file); the state is the ordered list of in-flight destination register sets.  States are memoised per block.

  python scripts/checks/asm_inflight_regs.py /tmp/gemm.s gemm256k_kernel      (tests/test_asm_checks.py runs it on every build)
"""
import re
import sys


def regs(tok):
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return frozenset(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return frozenset({int(m.group(1))}) if m else frozenset()


def parse_blocks(lines):
    """-> (blocks: {label: [(lineno, op, toks, raw)]}, order: [labels], succ: {label: [labels]})"""
    blocks, order, cur = {}, [], "__entry__"
    blocks[cur] = []
    order.append(cur)
    for ln, raw in lines:
        l = raw.split(";")[0].strip()
        if not l:
            continue
        m = re.match(r"^(\.LBB\w+):", l)
        if m:
            cur = m.group(1)
            blocks[cur] = []
            order.append(cur)
            continue
        if l.startswith(".") or l.endswith(":"):
            continue
        op, *rest = l.split(None, 1)
        toks = [t.strip().split()[0] for t in (rest[0].split(",") if rest else []) if t.strip()]
        blocks[cur].append((ln, op, toks, raw.strip()))
    succ = {}
    for i, b in enumerate(order):
        ins = blocks[b]
        nxt = order[i + 1] if i + 1 < len(order) else None
        out = []
        ended = False
        for _, op, toks, _ in ins:
            if op == "s_branch":
                out.append(toks[0]); ended = True; break
            if op.startswith("s_cbranch"):
                out.append(toks[0])
            if op == "s_endpgm":
                ended = True; break
        if not ended and nxt:
            out.append(nxt)
        succ[b] = out
    return blocks, order, succ


def step(state, ins, name, report):
    """state: tuple of frozensets (oldest first).  Returns the new state."""
    st = list(state)
    for ln, op, toks, raw in ins:
        if op.startswith("ds_read") or op.startswith("ds_load"):
            addr = frozenset().union(*[regs(t) for t in toks[1:]]) if len(toks) > 1 else frozenset()
            for rs in st:
                if rs & addr:
                    report(f"{name}:{ln}: address register still in flight: {raw}")
            st.append(regs(toks[0]))
            continue
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", raw)
            if m:
                n = int(m.group(1))
                st = st[len(st) - n:] if n < len(st) else st
                if n == 0:
                    st = []
            continue
        if op in ("s_branch", "s_endpgm"):
            break
        touched = frozenset().union(*[regs(t) for t in toks]) if toks else frozenset()
        for k, rs in enumerate(st):
            if rs & touched:
                report(f"{name}:{ln}: v{sorted(rs & touched)[0]}.. touched while in flight ({len(st) - 1 - k} younger reads): {raw}")
    return tuple(st)


def check_kernel(lines, name):
    blocks, order, succ = parse_blocks(lines)
    msgs = set()
    seen = set()
    work = [(order[0], ())]
    n_states = 0
    while work:
        b, st = work.pop()
        if (b, st) in seen:
            continue
        seen.add((b, st))
        n_states += 1
        if n_states > 20000:
            msgs.add(f"{name}: state explosion"); break
        out = step(st, blocks[b], name, msgs.add)
        for s in succ[b]:
            if s in blocks:
                work.append((s, out))
    return sorted(msgs), n_states


def main():
    path, pat = sys.argv[1], sys.argv[2]
    txt = open(path).read().split("\n")
    starts = [i for i, l in enumerate(txt) if re.match(rf"^_ZN\S*{pat}\S*:", l)]
    total = 0
    for s in starts:
        name = txt[s].split(":")[0]
        e = next(i for i in range(s, len(txt)) if ".amdhsa_kernel" in txt[i])
        body = list(enumerate(txt[s + 1:e], s + 2))
        msgs, n = check_kernel(body, name[-44:])
        nr = sum("ds_read_b128" in l for _, l in body)
        print(f"{name}: {'OK' if not msgs else str(len(msgs)) + ' VIOLATIONS'} ({nr} ds_read_b128, {n} block states)")
        for m in msgs[:6]:
            print("   ", m)
        total += len(msgs)
    sys.exit(1 if total else 0)


if __name__ == "__main__":
    main()
