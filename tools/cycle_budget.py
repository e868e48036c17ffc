"""Issue-cycle budget of k_render<0> per wave and ray iteration, from the compiled assembly (make asm) and the wave-level
frequencies the PT_STATS build reports (tools/stats_run.py). Every basic block of the kernel's main loop is assigned to a
region by what it contains, weighted by how often it runs per iteration, and its vector instructions are priced in
issue cycles with the cost model of DESIGN.md section 3.1, measured with tools/valu_peak.hip (profiles/r03_valu_peak.log):
4 cycles per wave64 instruction (packed fp32, anything with a scalar or literal operand, max/min/compare/select, conversions,
fp64 add/mul/fma, 32- and 64-bit integer multiplies), 2.4 cycles for the all-VGPR forms of v_mov / v_add / v_sub / v_mul /
v_fma / v_fmac / v_xor / v_bitop3 / v_lshrrev / v_add_u32 ("fast" below), 8 for fp32 transcendentals, 16 for v_rsq_f64.

Usage: python3 tools/cycle_budget.py [asm file] [quads flagged_quads resolved_spheres sky_share finish_share]
"""
import re
import sys

asm = sys.argv[1] if len(sys.argv) > 1 and not sys.argv[1][0].isdigit() else "tools/build/pt_hip.s"
nums = [float(a) for a in sys.argv[1:] if a[0].isdigit()]
QUADS, FLAGGED, RESOLVED, SKY, FINISH = (nums + [126.94, 18.74, 21.75, 0.998, 0.891][len(nums):])[:5]
JOB_SWITCH = 0.04      # job switches per iteration at configs[1]: 324,000 jobs / 8.51e6 wave iterations

text = open(asm).read()
start = text.index("_Z8k_renderILi0EEv12RenderParams:")
body = text[start:text.index("s_endpgm", start)]

blocks, cur = [], {"name": "entry", "ins": [], "depth": 0}
for ln in body.split("\n"):
    m = re.match(r"^(\.LBB\d+_\d+):\s*(;.*)?$", ln) or re.match(r"^; (%bb\.\d+):\s*(;.*)?$", ln)
    if m:
        blocks.append(cur)
        depth = 2 if "Depth=2" in ln else 1 if "Depth=1" in ln else 0
        cur = {"name": m.group(1), "ins": [], "depth": depth}
        continue
    t = ln.strip()
    if t and not t.startswith(";") and not t.startswith("."):
        cur["ins"].append(t)
blocks.append(cur)

TRANS = ("v_sqrt_f32", "v_rsq_f32", "v_rcp_f32", "v_rcp_iflag_f32", "v_sin_f32", "v_cos_f32", "v_exp_f32", "v_log_f32")
SLOW16 = ("v_rsq_f64", "v_rcp_f64", "v_sqrt_f64")
FAST = ("v_mov_b32", "v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_sub_f32", "v_subrev_f32", "v_xor_b32", "v_bitop3_b32",
        "v_lshrrev_b32", "v_lshlrev_b32", "v_add_u32", "v_sub_u32")


def opname(ins):
    return ins.split()[0].replace("_e32", "").replace("_e64", "")


def is_fast(ins):
    """An all-VGPR form of one of the FAST opcodes: no scalar register, no literal, no inline constant among the sources."""
    if opname(ins) not in FAST:
        return False
    ops = [o.strip() for o in ins.split(None, 1)[1].split(",")] if len(ins.split(None, 1)) > 1 else []
    srcs = [o.split()[0] for o in ops[1:]]
    return all(re.match(r"^-?\|?v(\d+|\[)", o) for o in srcs)


def cost(ins):
    base = opname(ins)
    if base in TRANS:
        return 8
    if base in SLOW16:
        return 16
    return 2.4 if is_fast(ins) else 4


def region(b):
    ops = [opname(o) for o in b["ins"]]
    c = lambda name: sum(1 for o in ops if o == name)
    if c("v_pk_fma_f32") >= 14:
        return "tier 1 (14 v_pk_fma_f32 + max, max3, cmp per quad)"
    if b["depth"] == 2:
        if c("v_sqrt_f32") and c("v_min_u32"):
            return "tier 2: roots of a flagged sphere"
        if c("v_mov_b32") == 2 and len([o for o in ops if o.startswith("v_")]) == 2:
            return "tier 2: closest-hit update (two moves under the mask)"
        return "tier 2: per-sphere compares of a flagged quad"
    if any(o.startswith("global_atomic") for o in ops) or c("ds_write_b128") or c("ds_write2_b64") or c("ds_read_b64") or c("s_setprio"):
        return "job switch: sums of a finished job to the frame, next job from the counter; stragglers (rare)"
    if c("v_cvt_f64_f32") >= 4:
        return "finish: hit record of the winner in fp64"
    if c("v_mad_u64_u32") >= 10:
        return "philox4x32-10"
    if c("v_sin_f32") and any(o.startswith("s_load") for o in ops):
        return "camera ray (polar lens point, 19 constants through the scalar cache)"
    if c("v_sin_f32"):
        return "shade (three materials side by side), normalise, far root of the own sphere"
    if c("v_fract_f32") or c("ds_add_u64") or c("global_atomic_add_x2"):
        return "sky colour, Q32.32 conversion, sums"
    if c("v_mbcnt_lo_u32_b32") or c("v_rcp_iflag_f32") or c("ds_read_b128") or c("ds_read2_b64"):
        return "regeneration (item -> pixel, sample)"
    return "other blocks of the loop (foot point, self root, job fetch, copies)"


# frequency of a block per ray iteration
def freq(b, reg):
    if reg.startswith("tier 1"):
        return QUADS / 2.0                      # two copies of the quad body (banks A and B), each runs for half the quads
    if reg.startswith("tier 2: per-sphere compares"):
        return None                             # priced below from the totals
    if reg.startswith("tier 2: roots"):
        return None
    if reg.startswith("tier 2: closest"):
        return None
    if reg.startswith("finish"):
        return FINISH
    if reg.startswith("sky"):
        return SKY
    if reg.startswith("job switch"):
        return JOB_SWITCH
    return 1.0


in_loop = False
table = {}
tier2 = {"cmp": [], "roots": [], "upd": []}
for b in blocks:
    if b["depth"] >= 1:
        in_loop = True
    elif in_loop and b["name"].startswith(".LBB") and b["depth"] == 0:
        in_loop = False if "ds_write" in " ".join(b["ins"]) or "global_store" in " ".join(b["ins"]) else in_loop
    if b["depth"] == 0:
        continue
    v = [o for o in b["ins"] if o.startswith("v_")]
    if not v:
        continue
    reg = region(b)
    n, cyc, fast = len(v), sum(cost(o) for o in v), sum(1 for o in v if is_fast(o))
    if reg.startswith("tier 2: per-sphere"):
        tier2["cmp"].append((n, cyc, fast)); continue
    if reg.startswith("tier 2: roots"):
        tier2["roots"].append((n, cyc, fast)); continue
    if reg.startswith("tier 2: closest"):
        tier2["upd"].append((n, cyc, fast)); continue
    f = freq(b, reg)
    t = table.setdefault(reg, [0.0, 0.0, 0.0])
    t[0] += n * f; t[1] += cyc * f; t[2] += fast * f

# tier 2 from the totals: a flagged quad runs its four compares; a resolved sphere its roots block; the update block runs
# when a lane's closest hit improves (bounded above by the resolved spheres: priced at half of them)
def mean(lst, k):
    return sum(x[k] for x in lst) / max(1, len(lst))
if tier2["cmp"]:
    # the compare blocks: 8 per two quads in the code; each flagged quad runs 4 of them (the first sits in the tier-1 block's tail)
    per_quad = sum(x[0] for x in tier2["cmp"]) / 2.0, sum(x[1] for x in tier2["cmp"]) / 2.0, sum(x[2] for x in tier2["cmp"]) / 2.0
    table["tier 2: per-sphere compares of a flagged quad"] = [per_quad[0] * FLAGGED, per_quad[1] * FLAGGED, per_quad[2] * FLAGGED]
if tier2["roots"]:
    table["tier 2: roots of a flagged sphere"] = [mean(tier2["roots"], k) * RESOLVED for k in range(3)]
if tier2["upd"]:
    table["tier 2: closest-hit update (two moves under the mask)"] = [mean(tier2["upd"], k) * RESOLVED * 0.5 for k in range(3)]

tot = [sum(t[k] for t in table.values()) for k in range(3)]
print("%-84s %10s %10s %8s" % ("region (per wave and ray iteration)", "vector ins", "cycles", "share"))
for reg, t in sorted(table.items(), key=lambda kv: -kv[1][1]):
    print("%-84s %10.0f %10.0f %7.1f%%" % (reg, t[0], t[1], 100 * t[1] / tot[1]))
print("%-84s %10.0f %10.0f   (%d of the instructions are fast forms at 2.4 cycles)" % ("total", tot[0], tot[1], tot[2]))
t1 = table[[k for k in table if k.startswith("tier 1")][0]][1]
print("tier 1 is %.3f of the cycles. One test = 17 flop and tier 1 spends 17 cycles per sphere and wave (64 flop per cycle and SIMD = the peak's rate),"
      " so at the clock the peak assumes the kernel would run at %.3f of the roofline; at a sustained clock f it runs at %.3f x f / 2.4 GHz." % (t1 / tot[1], t1 / tot[1], t1 / tot[1]))
