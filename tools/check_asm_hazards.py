"""Static check of the gfx950 ISA for two hazards the compiler's hazard recognizer does not handle INSIDE inline asm:
  (a) a VALU instruction writes an SGPR (v_readlane / v_readfirstlane / v_cmp ... to SGPRs) and an asm vector-memory instruction reads
      that SGPR (descriptor, soffset) fewer than 5 wait states later;
  (b) an asm vector-memory STORE of more than 8 bytes per lane is followed within 2 wait states by an instruction that overwrites its
      data registers.
Usage: python tools/check_asm_hazards.py [file.hip ...]   (default: every primia_amd/csrc/*.hip); compiles each to ISA with hipcc."""
import glob, os, re, subprocess, sys, tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "..", "primia_amd", "csrc")


def sregs(tok):
    tok = tok.strip().rstrip(",")
    m = re.fullmatch(r"s\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"s(\d+)", tok)
    return {int(m.group(1))} if m else set()


def vregs(tok):
    tok = tok.strip().rstrip(",")
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def wait_states(ins):
    m = re.match(r"s_nop (\d+)", ins)
    return int(m.group(1)) + 1 if m else 1


def check(path):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        r = subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                            "-I", CSRC, "-I", os.path.join(HERE, "..", "include"), path, "-o", out], capture_output=True, text=True)
        if r.returncode != 0:
            return [f"{os.path.basename(path)}: does not compile stand-alone"]
        lines = [l.strip() for l in open(out)]
    problems = []
    ins = []          # (text, in_asm)
    in_asm = False
    for l in lines:
        if l.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if l.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not l or l.startswith(";") or l.startswith(".") or l.endswith(":"):
            if l.endswith(":") and not l.startswith(";"):
                ins.append(("<label>", False))
            continue
        ins.append((l.split(";")[0].strip(), in_asm))
    for k, (t, a) in enumerate(ins):
        if not a or not re.match(r"(buffer|global)_(load|store)", t):
            continue
        ops = t.split(None, 1)[1] if " " in t else ""
        toks = [x.strip() for x in ops.split(",")]
        used_s = set()
        for x in toks:
            used_s |= sregs(x.split()[0] if x else "")
        # (a) look back for VALU writes of those SGPRs
        ws = 0
        j = k - 1
        while j >= 0 and ws < 5:
            pt, _ = ins[j]
            if pt == "<label>":
                break
            if re.match(r"v_(readlane|readfirstlane)_b32", pt) or re.match(r"v_cmp", pt):
                dst = pt.split(None, 1)[1].split(",")[0]
                if sregs(dst) & used_s:
                    problems.append(f"{os.path.basename(path)}: VALU-written SGPR {dst} read by asm `{t[:60]}` after {ws} wait states")
            ws += wait_states(pt)
            j -= 1
        # (b) stores wider than 8 bytes
        if re.match(r"(buffer|global)_store_dwordx[34]", t):
            data = vregs(toks[0])
            ws = 0
            j = k + 1
            while j < len(ins) and ws < 2:
                nt, _ = ins[j]
                if nt == "<label>":
                    break
                if re.match(r"v_", nt):
                    dst = nt.split(None, 1)[1].split(",")[0] if " " in nt else ""
                    if vregs(dst) & data:
                        problems.append(f"{os.path.basename(path)}: `{nt[:50]}` overwrites the data of asm `{t[:50]}` after {ws} wait states")
                ws += wait_states(nt)
                j += 1
    return problems


if __name__ == "__main__":
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    bad = []
    for f in files:
        bad += check(f)
    print("\n".join(bad) if bad else f"no asm hazards found in {len(files)} files")
    sys.exit(1 if bad else 0)
