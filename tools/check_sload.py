"""Static check of the hand-written scalar loads in the compiled kernels (tools/build/pt_hip.s from `make asm`).

pt_device.h issues `s_load_dwordx16` from inline asm and waits for it with a separate statement, so hipcc does
not know that the destination registers are in flight in between. This script walks every kernel's
instruction stream and fails if any instruction between such a load and the next `s_waitcnt lgkmcnt(0)`
reads or writes one of the registers being loaded (a spill, a copy or a reuse there would move garbage).
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


def check(path):
    lines = open(path).read().split("\n")
    problems, loads = [], 0
    kernel = None
    pending = None          # (line number, set of destination registers)
    in_asm = False
    for i, raw in enumerate(lines, 1):
        line = raw.strip()
        if re.match(r"^_Z\w+:", line):
            kernel, pending = line[:-1], None
        if line.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not line or line.startswith(";") or line.startswith(".") or line.endswith(":"):
            continue
        code = line.split(";")[0]
        if in_asm and code.startswith("s_load_dwordx16"):
            dst = regs(code.split(",")[0])
            if pending:
                problems.append("%s:%d: second hand-written load before the wait for the one at line %d" % (kernel.split(":")[0], i, pending[0]))
            pending = (i, dst)
            loads += 1
            continue
        if pending:
            if code.startswith("s_waitcnt") and "lgkmcnt(0)" in code:
                pending = None
                continue
            # Branches inside the window are expected (the tier-2 blocks sit between a load and its wait and are
            # laid out inline), so the text between the load and the next wait covers the paths taken; leaving
            # the kernel or jumping through a register with a load in flight is not.
            if code.startswith(("s_endpgm", "s_setpc", "s_swappc")):
                problems.append("%s:%d: leaves while the load at line %d is in flight: %s" % (kernel.split(":")[0], i, pending[0], code))
            touched = regs(code) & pending[1]
            if touched:
                problems.append("%s:%d: touches s%s while the load at line %d is in flight: %s" % (kernel.split(":")[0], i, sorted(touched), pending[0], code))
    return loads, problems


if __name__ == "__main__":
    n, probs = check(sys.argv[1] if len(sys.argv) > 1 else "tools/build/pt_hip.s")
    print("hand-written scalar loads checked:", n)
    for p in probs:
        print("PROBLEM", p)
    sys.exit(1 if probs or n == 0 else 0)
