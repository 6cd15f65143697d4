"""Static check of the gfx950 ISA for hazards the compiler's hazard recognizer / wait-count insertion do not handle INSIDE inline
asm (it treats an asm block as opaque):
  (a) a VALU instruction writes an SGPR (v_readlane / v_readfirstlane / v_cmp / carry-outs ...) and an asm vector-memory
      instruction reads that SGPR (descriptor, soffset) fewer than 5 wait states later;
  (b) an asm vector-memory STORE of more than 8 bytes per lane is followed within 2 wait states by an instruction that
      overwrites its data registers (round 5: conv3x3_lh4's unrolled write-back stored the NEXT fragment's values);
  (c) an asm LDS-DMA instruction (buffer_load ... lds / global_load_lds_*) reads M0 in the wait state right after an SALU
      instruction wrote it (1 wait state required);
  (d) an asm instruction LOADS into registers (buffer / global / ds load issued by asm: the compiler does not know it is a load
      and counts nothing for it) and some instruction reads or overwrites those registers before an s_waitcnt of that counter
      has retired the load (round 5: mask words copied by the compiler right after the asm load was issued).
The scan follows the control flow: fall-through ACROSS labels, both arms of conditional branches, and branch back-edges into
a label (round 5's version stopped at labels).

Input: the device ISA of the objects `python -m primia_amd.build` actually links — it compiles with -save-temps and keeps
primia_amd/csrc/_build/isa/<name>.s next to <name>.o (the flag does not change the code).  A missing or stale .s is an error,
not a reason to recompile something else.
Usage: python tools/check_asm_hazards.py [name.hip | name.s ...]   (default: every object of the build)."""
import glob, os, re, sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "primia_amd", "csrc")
ISA = os.path.join(CSRC, "_build", "isa")
OBJ = os.path.join(CSRC, "_build")


def _regs(tok, kind):
    tok = tok.strip().rstrip(",")
    m = re.fullmatch(kind + r"\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(kind + r"(\d+)", tok)
    return {int(m.group(1))} if m else set()


def sregs(tok):
    return _regs(tok, "s")


def vregs(tok):
    return _regs(tok, "v")


def operands(t):
    ops = t.split(None, 1)[1] if " " in t else ""
    return [x.strip().split()[0] if x.strip() else "" for x in ops.split(",")]


def all_vregs(t):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", t):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1)) if m.group(1) else {int(m.group(3))}
    return out


def wait_states(ins):
    m = re.match(r"s_nop (\d+)", ins)
    return int(m.group(1)) + 1 if m else 1


_SDST_SECOND = re.compile(r"v_(add_co|sub_co|subrev_co|addc_co|subb_co|subbrev_co|mad_u64_u32|mad_i64_i32|div_scale)")
_TWO_DST = re.compile(r"v_(swap_b32|permlane16_swap|permlane32_swap)")
_UNCOND = re.compile(r"s_(branch|endpgm|setpc_b64)\b")


def valu_sgpr_dst(t):
    if re.match(r"v_(readlane|readfirstlane)_b32", t) or re.match(r"v_cmp", t):
        ops = operands(t)
        return sregs(ops[0]) if ops else set()
    if _SDST_SECOND.match(t):
        ops = operands(t)
        return sregs(ops[1]) if len(ops) > 1 else set()
    return set()


def vgpr_dsts(t):
    """VGPRs an instruction writes (vector ALU / matrix / memory loads into registers)."""
    if not re.match(r"(v_|ds_read|ds_load|buffer_load|global_load|flat_load|scratch_load)", t):
        return set()
    if re.search(r"\blds\b", t) or "_lds_" in t.split()[0]:
        return set()
    if re.match(r"v_(cmp|cmpx|readlane|readfirstlane|nop)", t):
        return set()
    ops = operands(t)
    d = vregs(ops[0]) if ops else set()
    if _TWO_DST.match(t) and len(ops) > 1:
        d |= vregs(ops[1])
    return d


def mem_class(t):
    """Which counter an instruction increments: 'vm' (vector memory, loads AND stores on gfx9), 'lgkm' (LDS, scalar memory)."""
    op = t.split()[0] if t else ""
    if re.match(r"(buffer|global|flat|scratch)_", op):
        return "vm"
    if re.match(r"(ds_|s_load|s_buffer_load|s_sendmsg)", op):
        return "lgkm"
    return None


def waitcnt(t):
    """{'vm': N, 'lgkm': N} of an s_waitcnt (only the counters it names)."""
    if not t.startswith("s_waitcnt"):
        return {}
    out = {}
    m = re.search(r"vmcnt\((\d+)\)", t)
    if m:
        out["vm"] = int(m.group(1))
    m = re.search(r"lgkmcnt\((\d+)\)", t)
    if m:
        out["lgkm"] = int(m.group(1))
    if not out and re.fullmatch(r"s_waitcnt\s+(0|0x0)", t):
        out = {"vm": 0, "lgkm": 0}
    return out


def parse(path):
    """[(text, in_asm)], {label: index of the first instruction after it}."""
    ins, labels, pending = [], {}, []
    in_asm = False
    for l in open(path):
        l = l.strip()
        if l.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if l.startswith(";;#ASMEND"):
            in_asm = False
            continue
        l = l.split(";")[0].strip()
        if not l:
            continue
        if l.endswith(":") and not l.startswith("."):
            pending.append(l[:-1])
            continue
        if re.match(r"\.L[\w$.]*:", l):
            pending.append(l[:-1])
            continue
        if l.startswith("."):
            continue
        for name in pending:
            labels[name] = len(ins)
        pending = []
        ins.append((l, in_asm))
    return ins, labels


def flow(ins, labels):
    n = len(ins)
    succ = [[] for _ in range(n)]
    pred = [[] for _ in range(n)]
    for k, (t, _) in enumerate(ins):
        tgt = None
        m = re.match(r"s_c?branch\w*\s+(\S+)", t)
        if m and m.group(1) in labels and labels[m.group(1)] < n:
            tgt = labels[m.group(1)]
        if not _UNCOND.match(t) and k + 1 < n:
            succ[k].append(k + 1)
        if tgt is not None:
            succ[k].append(tgt)
    for k in range(n):
        for j in succ[k]:
            pred[j].append(k)
    return succ, pred


def isa_path(arg):
    name = os.path.basename(arg)
    name = name[:-4] if name.endswith(".hip") else (name[:-2] if name.endswith(".s") else name)
    return os.path.join(ISA, name + ".s"), os.path.join(OBJ, name + ".o"), os.path.join(CSRC, name + ".hip")


def check(arg):
    path, obj, src = isa_path(arg)
    base = os.path.basename(src)
    if not os.path.exists(path) or not os.path.exists(obj):
        return [f"{base}: no ISA of the built object ({os.path.relpath(path)}): run `python -m primia_amd.build`"]
    if os.path.exists(src) and os.path.getmtime(path) < os.path.getmtime(src):
        return [f"{base}: the kept ISA is older than the source: run `python -m primia_amd.build`"]
    return check_isa(path, base)


def check_isa(path, base=None):
    """The hazard scan proper over one file of device assembly (tests/test_asm_hazards.py feeds it hand-written cases)."""
    base = base or os.path.basename(path)
    ins, labels = parse(path)
    succ, pred = flow(ins, labels)
    problems = set()
    for k, (t, a) in enumerate(ins):
        if not a:
            continue
        op = t.split()[0]
        is_vmem = re.match(r"(buffer|global)_(load|store)", op) is not None
        is_lds_dma = is_vmem and (re.search(r"\blds\b", t) is not None or "_lds_" in op)
        ops = operands(t)
        if is_vmem:
            used_s = set()
            for x in ops:
                used_s |= sregs(x)
            # (a) VALU-written SGPRs, any path backwards, < 5 wait states
            stack, seen = [(j, 0) for j in pred[k]], set()
            while stack:
                j, ws = stack.pop()
                if ws >= 5 or (j, ws) in seen:
                    continue
                seen.add((j, ws))
                pt = ins[j][0]
                if valu_sgpr_dst(pt) & used_s:
                    problems.add(f"{base}: VALU-written SGPR of `{pt[:50]}` read by asm `{t[:60]}` after {ws} wait states")
                for i in pred[j]:
                    stack.append((i, ws + wait_states(pt)))
            # (c) M0 written by the SALU in the wait state before an LDS-DMA instruction
            if is_lds_dma:
                for j in pred[k]:
                    pt = ins[j][0]
                    if re.match(r"s_\w+\s+m0\b", pt):
                        problems.add(f"{base}: `{pt[:40]}` directly before asm `{t[:60]}` (M0 needs 1 wait state)")
        # (b) wide stores: data registers overwritten within 2 wait states, any path forwards
        if re.match(r"(buffer|global)_store_dwordx[34]", op):
            data = vregs(ops[0])
            stack, seen = [(j, 0) for j in succ[k]], set()
            while stack:
                j, ws = stack.pop()
                if ws >= 2 or (j, ws) in seen:
                    continue
                seen.add((j, ws))
                nt = ins[j][0]
                if vgpr_dsts(nt) & data:
                    problems.add(f"{base}: `{nt[:50]}` overwrites the data of asm `{t[:50]}` after {ws} wait states")
                for i in succ[j]:
                    stack.append((i, ws + wait_states(nt)))
        # (d) asm-issued loads into registers: nobody may touch the destination before a wait retires the load
        cls = mem_class(t)
        dst = vgpr_dsts(t) if cls and not is_lds_dma and re.match(r"(buffer_load|global_load|flat_load|ds_read|ds_load)", op) else set()
        if dst:
            stack, seen = [(j, 0) for j in succ[k]], set()
            while stack:
                j, younger = stack.pop()
                if (j, min(younger, 64)) in seen or len(seen) > 4000:
                    continue
                seen.add((j, min(younger, 64)))
                nt = ins[j][0]
                w = waitcnt(nt)
                if cls in w and w[cls] <= younger:
                    continue                      # retired on this path
                if all_vregs(nt) & dst and not nt.startswith("s_waitcnt"):
                    problems.add(f"{base}: `{nt[:50]}` touches the destination of asm load `{t[:50]}` before an s_waitcnt "
                                 f"retires it ({younger} younger {cls} operations)")
                    continue
                if _UNCOND.match(nt) and not succ[j]:
                    continue
                for i in succ[j]:
                    stack.append((i, younger + (1 if mem_class(nt) == cls else 0)))
    return sorted(problems)


def built_objects():
    return sorted(os.path.basename(p)[:-2] for p in glob.glob(os.path.join(OBJ, "*.o")))


if __name__ == "__main__":
    files = sys.argv[1:] or built_objects()
    bad = []
    for f in files:
        bad += check(f)
    print("\n".join(bad) if bad else f"no asm hazards found in the ISA of {len(files)} built objects")
    sys.exit(1 if bad else 0)
