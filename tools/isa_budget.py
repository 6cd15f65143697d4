"""Instruction budget of a kernel from the ISA the build kept (primia_amd/csrc/_build/isa/<object>.s — the code that ships):
static counts per LOOP (a back-edge's span) and for the code outside every loop, by class — MFMA, other VALU, SALU, LDS (ds_*),
vector memory (loads, LDS-DMA, stores), waits, barriers, branches.  For a steady-state loop without inner conditionals the
static body IS the per-iteration dynamic count; the table also gives VALU / SALU / DS per MFMA for every loop.
    python tools/isa_budget.py conv3x3_lh2 'conv3x3_lh2_kernelILi392ELb0ELb0'      (object, regex on the mangled kernel name)
    python tools/isa_budget.py --round6          (the three kernels VERDICT r05 item 7 names -> profiles/r06_isa_budget.txt)"""
import os, re, sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import check_asm_hazards as H

CLASSES = ["mfma", "valu", "salu", "ds", "vmem_ld", "lds_dma", "vmem_st", "wait", "barrier", "branch", "other"]


def classify(t):
    op = t.split()[0]
    if op.startswith("v_mfma") or op.startswith("v_smfma"):
        return "mfma"
    if op.startswith("v_"):
        return "valu"
    if op.startswith("ds_"):
        return "ds"
    if re.match(r"(buffer|global|flat|scratch)_", op):
        if "store" in op or "atomic" in op:
            return "vmem_st"
        return "lds_dma" if (re.search(r"\blds\b", t) or "_lds_" in op) else "vmem_ld"
    if op.startswith("s_waitcnt") or op.startswith("s_nop") or op.startswith("s_sleep"):
        return "wait"
    if op.startswith("s_barrier"):
        return "barrier"
    if re.match(r"s_c?branch|s_endpgm|s_setpc|s_call", op):
        return "branch"
    if op.startswith("s_"):
        return "salu"
    return "other"


def functions(path):
    """[(name, first instruction index, last + 1)] from the symbol labels of a kept .s, with the parsed instruction list."""
    ins, labels = H.parse(path)
    starts = sorted((i, n) for n, i in labels.items() if n.startswith("_Z"))
    out = []
    for k, (i, n) in enumerate(starts):
        end = starts[k + 1][0] if k + 1 < len(starts) else len(ins)
        if end > i:
            out.append((n, i, end))
    return ins, labels, out


def budget(ins, labels, lo, hi):
    cls = [classify(t) for t, _ in ins[lo:hi]]
    loops = []
    for j in range(lo, hi):
        m = re.match(r"s_c?branch\w*\s+(\S+)", ins[j][0])
        if m and m.group(1) in labels and lo <= labels[m.group(1)] <= j:
            loops.append((labels[m.group(1)], j + 1))
    loops = sorted(set(loops))

    def count(a, b, holes=()):
        c = dict.fromkeys(CLASSES, 0)
        for i in range(a, b):
            if any(x <= i < y for x, y in holes):
                continue
            c[cls[i - lo]] += 1
        return c

    rows = []
    for a, b in loops:
        inner = [(x, y) for x, y in loops if (x, y) != (a, b) and a <= x and y <= b]
        depth = sum(1 for x, y in loops if (x, y) != (a, b) and x <= a and b <= y)
        rows.append((a, b, depth, count(a, b, inner), bool(inner)))
    top = [(x, y) for x, y in loops if not any((p, q) != (x, y) and p <= x and y <= q for p, q in loops)]
    return rows, count(lo, hi, top), count(lo, hi)


def fmt(c):
    m = max(c["mfma"], 1)
    per = f"{c['valu'] / m:5.2f} {c['salu'] / m:5.2f} {c['ds'] / m:5.2f}" if c["mfma"] else "    -     -     -"
    return (f"{c['mfma']:6d} {c['valu']:6d} {c['salu']:6d} {c['ds']:5d} {c['vmem_ld']:5d} {c['lds_dma']:5d} {c['vmem_st']:5d} "
            f"{c['wait']:5d} {c['barrier']:4d} {c['branch']:5d} | {per}")


def report(obj, pattern, out=sys.stdout):
    path = H.isa_path(obj)[0]
    ins, labels, funcs = functions(path)
    for name, lo, hi in funcs:
        if not re.search(pattern, name):
            continue
        rows, outside, total = budget(ins, labels, lo, hi)
        print(f"\n== {name}  ({hi - lo} instructions; {os.path.relpath(path)})", file=out)
        print("   span (instr index)  depth    MFMA   VALU   SALU    DS  VMld LDSdma VMst  wait barr   br | VALU/ SALU/  DS/ per MFMA", file=out)
        for a, b, depth, c, has_inner in rows:
            tag = "loop" + (" (own code, inner loops excluded)" if has_inner else "")
            print(f"   [{a - lo:6d},{b - lo:6d})  {depth:3d}   {fmt(c)}   {tag}", file=out)
        print(f"   outside every loop       {fmt(outside)}   prologue / epilogue / tile set-up", file=out)
        print(f"   whole kernel             {fmt(total)}", file=out)


ROUND6 = [("conv3x3_lh2", r"conv3x3_lh2_kernelILi392ELb0ELb0"), ("conv3x3_lh2", r"conv3x3_lh2_kernelILi392ELb1ELb0"),
          ("conv3x3_lh4", r"conv3x3_lh4_kernelILi196ELb0ELb0"), ("conv_wgrad_patch", r"conv_wgrad_patch33lw_kernelILi8ELi8ELi3")]

if __name__ == "__main__":
    if sys.argv[1:2] == ["--round6"]:
        for obj, pat in ROUND6:
            report(obj, pat)
    else:
        report(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else ".")
