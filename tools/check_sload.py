"""Static check of the hand-written scalar loads in the compiled kernels (tools/build/pt_hip.s from `make asm`).

pt_device.h issues `s_load_dwordx16` from inline asm and waits for it with a separate statement, so hipcc does
not know that the destination registers are in flight in between. This script builds the control-flow graph of
every kernel and follows EVERY path from such a load to the `s_waitcnt lgkmcnt(0)` that ends its flight; it fails
if an instruction on the way reads or writes one of the registers being loaded (a spill, a copy or a reuse there
would move garbage), if the kernel can end or jump through a register with the load in flight, or if a second
hand-written load is issued before the wait.
"""
import re
import sys


def regs(tok):
    """Scalar registers named by an operand token: s12 -> {12}, s[4:7] -> {4,5,6,7}."""
    out = set()
    for m in re.finditer(r"\bs\[(\d+):(\d+)\]", tok):
        out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    for m in re.finditer(r"\bs(\d+)\b", tok):
        out.add(int(m.group(1)))
    return out


def kernels(path):
    """Yields (name, instructions, labels): instructions = [(line number, text, hand_written)], labels = {name: index}."""
    name, ins, labels, in_asm = None, [], {}, False
    for i, raw in enumerate(open(path).read().split("\n"), 1):
        line = raw.strip()
        m = re.match(r"^(_Z\w+):", line)
        if m:
            if name and ins:
                yield name, ins, labels
            name, ins, labels, in_asm = m.group(1), [], {}, False
            continue
        if name is None:
            continue
        if line.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"^(\.LBB[0-9_]+):", line)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        if not line or line.startswith(";") or line.startswith(".") or line.endswith(":"):
            continue
        code = line.split(";")[0].strip()
        if code:
            ins.append((i, code, in_asm))
            if code.startswith("s_endpgm"):
                yield name, ins, labels
                name, ins, labels = None, [], {}
    if name and ins:
        yield name, ins, labels


def successors(ins, labels, k):
    code = ins[k][1]
    op = code.split()[0]
    if op == "s_endpgm":
        return []
    if op == "s_branch":
        return [labels[code.split()[1]]]
    if op.startswith("s_cbranch"):
        return [labels[code.split()[1]]] + ([k + 1] if k + 1 < len(ins) else [])
    return [k + 1] if k + 1 < len(ins) else []


def check(path):
    problems, loads = [], 0
    for name, ins, labels in kernels(path):
        for k, (ln, code, hand) in enumerate(ins):
            if not (hand and code.startswith("s_load_dwordx16")):
                continue
            loads += 1
            dst = regs(code.split(",")[0])
            seen, stack = set(), list(successors(ins, labels, k))
            while stack:
                j = stack.pop()
                if j in seen:
                    continue
                seen.add(j)
                ln2, c2, hand2 = ins[j]
                if c2.startswith("s_waitcnt") and "lgkmcnt(0)" in c2:
                    continue                                    # the flight ends on this path
                if c2.startswith(("s_endpgm", "s_setpc", "s_swappc")):
                    problems.append("%s:%d: leaves while the load at line %d is in flight: %s" % (name, ln2, ln, c2))
                    continue
                if hand2 and c2.startswith("s_load_dwordx16"):
                    problems.append("%s:%d: second hand-written load before the wait for the one at line %d" % (name, ln2, ln))
                touched = regs(c2) & dst
                if touched:
                    problems.append("%s:%d: touches s%s while the load at line %d is in flight: %s" % (name, ln2, sorted(touched), ln, c2))
                stack.extend(successors(ins, labels, j))
    return loads, sorted(set(problems))


if __name__ == "__main__":
    n, probs = check(sys.argv[1] if len(sys.argv) > 1 else "tools/build/pt_hip.s")
    print("hand-written scalar loads checked:", n)
    for p in probs:
        print("PROBLEM", p)
    sys.exit(1 if probs or n == 0 else 0)
