#!/usr/bin/env python3
"""Static instruction mix of a kernel's hot loop, from hipcc's gfx950 assembly.

    hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S forge_ec_amd/csrc/fecgpu.hip -o /tmp/fecgpu.s
    python tools/isa_mix.py /tmp/fecgpu.s <mangled-kernel-substring> [--loop outer|all] [--md]
    python tools/isa_mix.py /tmp/kernels_p256.s <mangled-kernel-substring> --region task_add [--md]

--region NAME (the scheduler kernels, whose loop body is "pop a batch, run ONE task"): the hot path between the
comment markers `; FEC_MARK NAME_begin` and `; FEC_MARK NAME_end`, which the sources emit when compiled with
-DFEC_ISA_MARKERS (limbs.hpp: FEC_MARK; the shipped library is built without them).  Also splits the region at
every `;;#ASMSTART` block of field_asm.inc, so the table shows what sits INSIDE the generated field products and
what the compiler put between them.

The hot loop is the largest natural loop of the kernel (the 256-step ladder): the span from the
target label of a backward branch to that branch.  Blocks the compiler placed outside that span
(cold continuations: the rare ripple / equal-points / non-canonical paths) are reported separately
as "outside the loop span".  Classes follow the measured issue costs (profiles/valu_rates*_r01.txt).
"""
import collections
import re
import sys

CLASSES = [
    ("v_mad_u64_u32", lambda m: m.startswith("v_mad_u64_u32")),
    ("v_mul_lo/hi_u32", lambda m: m.startswith("v_mul_lo_u32") or m.startswith("v_mul_hi_u32")),
    ("v_addc/v_subb (carry in)", lambda m: m.startswith(("v_addc_co", "v_subb_co", "v_subbrev_co"))),
    ("v_add_co/v_sub_co (carry out)", lambda m: m.startswith(("v_add_co", "v_sub_co", "v_subrev_co"))),
    ("v_cndmask", lambda m: m.startswith("v_cndmask")),
    ("v_cmp", lambda m: m.startswith("v_cmp")),
    ("v_mov", lambda m: m.startswith("v_mov") or m.startswith("v_accvgpr")),
    ("v_alignbit/shift/logic/other VALU", lambda m: m.startswith("v_")),
    ("s_nop", lambda m: m.startswith("s_nop")),
    ("s_waitcnt/s_barrier", lambda m: m.startswith(("s_waitcnt", "s_barrier"))),
    ("s_branch/s_cbranch", lambda m: m.startswith(("s_branch", "s_cbranch"))),
    ("other SALU", lambda m: m.startswith("s_")),
    ("ds_read/ds_write/ds_*", lambda m: m.startswith("ds_")),
    ("scratch_*", lambda m: m.startswith("scratch_")),
    ("global/buffer/flat", lambda m: m.startswith(("global_", "buffer_", "flat_"))),
]


def classify(m):
    for name, pred in CLASSES:
        if pred(m):
            return name
    return "other"


def region_mix(insts, asm_flags, label_at, marks, region, kernel, md):
    """Hot path (rare lane masks = 0, like the ladder walk) from marker <region>_begin to <region>_end."""
    b, e = marks.get(region + "_begin"), marks.get(region + "_end")
    if b is None or e is None:
        raise SystemExit("markers %s_begin / %s_end not found (compile with -DFEC_ISA_MARKERS)" % (region, region))
    path, pc, scc, steps = [], b, None, 0
    while pc != e and steps < 200000:
        m, ops = insts[pc]
        path.append((m, asm_flags[pc]))
        steps += 1
        if m in ("s_cmp_eq_u64", "s_cmp_lg_u64") and ops.replace(" ", "").endswith(",0"):
            scc = 1 if m == "s_cmp_eq_u64" else 0
        elif m.startswith(("s_cmp", "s_add", "s_sub", "s_and", "s_or", "s_xor", "s_lshl", "s_lshr", "s_bitcmp", "s_andn2", "s_orn2", "s_not")):
            scc = None
        taken = m == "s_branch" or (m == "s_cbranch_scc0" and scc == 0) or (m == "s_cbranch_scc1" and scc == 1) or \
            m in ("s_cbranch_vccz", "s_cbranch_execnz")
        pc = label_at[ops.split()[0].rstrip(",")] if taken else pc + 1
    if pc != e:
        raise SystemExit("the walk from %s_begin did not reach %s_end" % (region, region))
    print("kernel: %s" % kernel)
    for title, sel in (("whole region", lambda a: True), ("inside the field_asm.inc statements", lambda a: a),
                       ("compiler-scheduled code between them", lambda a: not a)):
        c = collections.Counter(classify(m) for m, a in path if sel(a))
        valu = sum(v for k, v in c.items() if k.startswith("v_"))
        print("region %s, %s: %d instructions, %d VALU" % (region, title, sum(c.values()), valu))
        if md:
            print("| class | count |\n|---|---|")
        for name, _ in CLASSES + [("other", None)]:
            if c.get(name, 0):
                print(("| %s | %d |" if md else "%-36s %8d") % (name, c[name]))
    print("top mnemonics: " + ", ".join("%s %d" % t for t in collections.Counter(m for m, _ in path).most_common(25)))


def main():
    path, needle = sys.argv[1], sys.argv[2]
    md = "--md" in sys.argv
    lines = open(path).read().split("\n")
    start = None
    for i, l in enumerate(lines):
        if re.match(r"^[A-Za-z_][\w$.]*:", l) and needle in l.split(":")[0]:
            start = i
            break
    if start is None:
        raise SystemExit("kernel not found: " + needle)
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    body = lines[start:end]
    label_at, insts, headers = {}, [], []  # instruction index of each label; (mnemonic, operands)
    marks, in_asm, asm_flags = {}, False, []   # FEC_MARK name -> instruction index; per instruction: inside an asm statement?
    for l in body:
        s = l.strip()
        if "FEC_MARK" in s:
            marks.setdefault(s.split("FEC_MARK", 1)[1].split()[0], len(insts))
            continue
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        mm = re.match(r"^(\.LBB\d+_\d+):", s)
        if mm:
            label_at[mm.group(1)] = len(insts)
            if "Loop Header: Depth=1" in s:
                headers.append(mm.group(1))
            continue
        s = s.split(";")[0].strip()
        if not s or s.startswith((".", "//")) or s.endswith(":"):
            continue
        parts = s.split(None, 1)
        insts.append((parts[0], parts[1] if len(parts) > 1 else ""))
        asm_flags.append(in_asm)
    region = sys.argv[sys.argv.index("--region") + 1] if "--region" in sys.argv else None
    if region:
        return region_mix(insts, asm_flags, label_at, marks, region, body[0].rstrip(":"), md)
    loops = []
    for idx, (m, ops) in enumerate(insts):
        if m.startswith(("s_cbranch", "s_branch")):
            t = ops.split()[0].rstrip(",")
            if t in label_at and label_at[t] <= idx:
                loops.append((label_at[t], idx, m.startswith("s_cbranch")))
    if not loops:
        raise SystemExit("no loop found")
    # the ladder loop's back edge is a conditional branch; unconditional backward branches are the
    # returns of out-of-line cold blocks
    # the ladder loop = the depth-1 loop (LLVM's own "Loop Header: Depth=1" annotation) with the
    # widest span; its back edges are all branches to that header
    cands = [r for r in loops if any(label_at[h] == r[0] for h in headers)] or [r for r in loops if r[2]] or loops
    lo, hi, _ = max(cands, key=lambda r: r[1] - r[0])
    inner = [r for r in loops if r[:2] != (lo, hi) and r[0] >= lo and r[1] <= hi]

    def mix(rng):
        c = collections.Counter()
        for m, _ in rng:
            c[classify(m)] += 1
        return c

    inside, outside = mix(insts[lo:hi + 1]), mix(insts[:lo] + insts[hi + 1:])
    total_in = sum(inside.values())
    valu = sum(v for k, v in inside.items() if k.startswith("v_"))
    print("kernel: %s" % body[0].rstrip(":"))
    print("hot loop span: %d instructions (%d VALU), %d instructions outside the span, %d nested loops inside"
          % (total_in, valu, sum(outside.values()), len(inner)))
    sep = " | " if md else "  "
    if md:
        print("| class | in loop span | outside |\n|---|---|---|")
    for name, _ in CLASSES + [("other", None)]:
        if inside.get(name, 0) or outside.get(name, 0):
            row = (name, inside.get(name, 0), outside.get(name, 0))
            print(("| %s | %d | %d |" if md else "%-36s %8d %8d") % row)
    top = collections.Counter(m for m, _ in insts[lo:hi + 1]).most_common(25)
    print("top mnemonics in the loop span: " + ", ".join("%s %d" % t for t in top))
    # Hot path of one iteration: walk from the loop head evaluating each conditional branch under
    # the assumption that every rare-path lane mask is zero (that is what the rare paths are: lane
    # masks tested with s_cmp_{eq,lg}_u64 mask, 0 or v_cmp + s_cbranch_vcc*), EXEC is non-zero, and
    # any other condition falls through; unconditional branches are followed.  The walk ends at the
    # loop's back edge.
    outer = [h for h in headers if any(h + ":" in l and "Inner" not in l for l in body)]

    def walk(start_pc):
        path, pc, steps, scc, first_visit = [], start_pc, 0, None, {}
        while steps < 400000:
            if pc in first_visit:  # the cycle closed: one iteration = the walk since the first visit
                return path[first_visit[pc]:]
            first_visit[pc] = len(path)
            m, ops = insts[pc]
            path.append(m)
            steps += 1
            if m in ("s_cmp_eq_u64", "s_cmp_lg_u64") and ops.replace(" ", "").endswith(",0"):
                scc = 1 if m == "s_cmp_eq_u64" else 0
            elif m.startswith(("s_cmp", "s_add", "s_sub", "s_and", "s_or", "s_xor", "s_lshl", "s_lshr", "s_bitcmp", "s_andn2", "s_orn2", "s_not")):
                scc = None  # SCC rewritten by something that is not a rare-mask test
            taken = False
            if m == "s_branch":
                taken = True
            elif m == "s_cbranch_scc0":
                taken = scc == 0
            elif m == "s_cbranch_scc1":
                taken = scc == 1
            elif m in ("s_cbranch_vccz", "s_cbranch_execnz"):
                taken = True
            if m.startswith("s_cbranch") and label_at.get(ops.split()[0].rstrip(",")) == start_pc:
                taken = True  # the loop's own back edge (its condition is the trip counter)
            if taken:
                pc = label_at[ops.split()[0].rstrip(",")]
                continue
            pc += 1
            if pc >= len(insts):
                return []
        return path

    # the ladder loop = the depth-1 loop whose iteration is longest
    path = max((walk(label_at[h]) for h in (outer or headers)), key=len, default=[]) or walk(lo)
    hot = collections.Counter(classify(m) for m in path)
    hv = sum(v for k, v in hot.items() if k.startswith("v_"))
    print("hot path of one iteration (rare masks = 0): %d instructions, %d VALU" % (len(path), hv))
    if md:
        print("| class | hot path, per iteration |\n|---|---|")
    for name, _ in CLASSES + [("other", None)]:
        if hot.get(name, 0):
            print(("| %s | %d |" if md else "%-36s %8d") % (name, hot[name]))
    print("top mnemonics on the hot path: " + ", ".join("%s %d" % t for t in collections.Counter(path).most_common(25)))


if __name__ == "__main__":
    main()
