"""Static check of the register allocation of the sample-loop kernels (the -Rpass-analysis=kernel-resource-usage
remarks `make asm` writes to tools/build/pt_hip.resources.txt).

The kernels run at the edge of their register budgets (k_render<0>: 80 vector registers at 6 waves per SIMD; the grid
variants: 96 at 5), and a harmless-looking edit can push hipcc into spilling a loop-carried value to scratch memory,
which costs tens of percent. This script fails if any of k_render<0|1|2> or k_tail uses scratch, spills, or loses
occupancy. Usage: python3 tools/check_resources.py tools/build/pt_hip.resources.txt
"""
import re
import sys

WANT = {                                  # kernel (mangled prefix) -> fewest waves per SIMD
    "_Z8k_renderILi0E": 6,
    "_Z8k_renderILi1E": 5,
    "_Z8k_renderILi2E": 5,
    "_Z6k_tailILb1E": 6,                  # k_tail<true>: render_core sizes its launch for six waves per SIMD (scene copies in LDS)
    "_Z6k_tailILb0E": 5,
}


def parse(path):
    """{function name: {field: int}} from the remarks."""
    out, cur = {}, None
    for line in open(path, errors="ignore"):
        m = re.search(r"remark: Function Name: (\S+)", line)
        if m:
            cur = out.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z \[\]/]+): (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return out


def check(path):
    """Returns (kernels checked, list of problems)."""
    res, problems, seen = parse(path), [], 0
    for prefix, waves in WANT.items():
        names = [n for n in res if n.startswith(prefix)]
        if not names:
            problems.append("%s: not found in %s" % (prefix, path))
            continue
        for n in names:
            r = res[n]
            seen += 1
            if r.get("ScratchSize [bytes/lane]", -1) != 0:
                problems.append("%s: uses %s bytes of scratch per lane" % (n, r.get("ScratchSize [bytes/lane]")))
            if r.get("VGPRs Spill", 0):                      # (scalar registers parked in vector lanes are reported, not refused)
                problems.append("%s: spills %s vector registers" % (n, r.get("VGPRs Spill")))
            if r.get("Occupancy [waves/SIMD]", 0) < waves:
                problems.append("%s: %s waves per SIMD, expected at least %d" % (n, r.get("Occupancy [waves/SIMD]"), waves))
    return seen, problems


if __name__ == "__main__":
    n, problems = check(sys.argv[1])
    for p in problems:
        print("PROBLEM:", p)
    for name, r in sorted(parse(sys.argv[1]).items()):
        if any(name.startswith(k) for k in WANT):
            print("%-40s VGPRs %3d  SGPRs %3d  scalar registers parked in vector lanes %2d  scratch %d  waves/SIMD %d" % (
                name[:40], r.get("VGPRs", -1), r.get("TotalSGPRs", -1), r.get("SGPRs Spill", -1),
                r.get("ScratchSize [bytes/lane]", -1), r.get("Occupancy [waves/SIMD]", -1)))
    print("kernels checked: %d" % n)
    sys.exit(1 if problems or n < len(WANT) else 0)
