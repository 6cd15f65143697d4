"""Build libprimia_hip.so (gfx950) with hipcc.  `python -m primia_amd.build [--force]`.

Each csrc/*.hip is compiled to an object (in parallel) and linked into ONE shared library that
lives in-tree next to this file, so it travels with the repo snapshot to the GPU box.  hipcc
cross-compiles without a GPU.
"""
import concurrent.futures
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(CSRC, "_build")
ISA = os.path.join(OBJ, "isa")      # device assembly of every object (kept by the build, see compile_one)
LIB = os.path.join(HERE, "libprimia_hip.so")
ARCH = "gfx950"
FLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wno-unused-result"]


def source_digest(files=None):
    """sha256 over the library's sources (csrc/*.hip, csrc/*.h, include/primia_hip.h; or the named csrc files only):
    profiles recorded with rocprofv3 carry it, and bench.py quotes a counter from a profile only when the digest of the
    code it is running equals the profile's."""
    import hashlib

    names = sorted(f for f in os.listdir(CSRC) if f.endswith((".hip", ".h"))) if files is None else sorted(files)
    paths = [os.path.join(CSRC, f) for f in names]
    if files is None:
        paths.append(os.path.join(HERE, "..", "include", "primia_hip.h"))
    h = hashlib.sha256()
    for q in paths:
        h.update(os.path.basename(q).encode() + b"\0")
        # CODE only: comments (`//` and `/* */`), trailing blanks and empty lines do not change what a profile measured (no
        # string literal of these sources contains a comment marker)
        import re

        text = re.sub(r"/\*.*?\*/", "", open(q, "r", encoding="utf-8", errors="replace").read(), flags=re.S)
        for line in text.splitlines():
            code = line.split("//", 1)[0].rstrip()
            if code:
                h.update(code.encode() + b"\n")
    return h.hexdigest()


def _hipcc():
    for c in ("/opt/rocm/bin/hipcc", "hipcc"):
        if os.path.isabs(c) and os.path.exists(c):
            return c
    return "hipcc"


def _newer(src, dst, deps=()):
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(p) > t for p in (src, *deps))


def build(force=False, verbose=True):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".h")]
    headers.append(os.path.join(HERE, "..", "include", "primia_hip.h"))
    sources = sorted(f for f in os.listdir(CSRC) if f.endswith(".hip"))
    jobs = []
    for f in sources:
        src = os.path.join(CSRC, f)
        obj = os.path.join(OBJ, f[:-4] + ".o")
        if force or _newer(src, obj, headers) or not os.path.exists(os.path.join(ISA, f[:-4] + ".s")):
            jobs.append((src, obj))

    def compile_one(job):
        # -save-temps: the device ISA of THIS object is kept as csrc/_build/isa/<name>.s (with the ;;#ASMSTART / ;;#ASMEND
        # markers of inline asm) — what tools/check_asm_hazards.py and tools/isa_budget.py read.  The flag does not change
        # the code (checked: the unbundled gfx950 code objects disassemble identically with and without it).
        src, obj = job
        name = os.path.basename(src)[:-4]
        tmp = os.path.join(OBJ, "tmp_" + name)
        shutil.rmtree(tmp, ignore_errors=True)
        os.makedirs(tmp)
        cmd = [_hipcc(), *FLAGS, "-save-temps=obj", "-c", src, "-o", os.path.join(tmp, name + ".o")]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode == 0:
            os.makedirs(ISA, exist_ok=True)
            os.replace(os.path.join(tmp, f"{name}-hip-amdgcn-amd-amdhsa-{ARCH}.s"), os.path.join(ISA, name + ".s"))
            os.replace(os.path.join(tmp, name + ".o"), obj)
        shutil.rmtree(tmp, ignore_errors=True)
        return src, r.returncode, r.stdout + r.stderr

    if jobs:
        with concurrent.futures.ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for src, rc, out in ex.map(compile_one, jobs):
                if verbose:
                    print(f"[primia build] {os.path.basename(src)}: {'ok' if rc == 0 else 'FAILED'}")
                if rc != 0:
                    raise RuntimeError(f"hipcc failed for {src}:\n{out}")
    objs = [os.path.join(OBJ, f[:-4] + ".o") for f in sources]
    if jobs or force or not os.path.exists(LIB):
        cmd = [_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
        if verbose:
            print(f"[primia build] linked {LIB}")
    return LIB


PROBE_LIB = os.path.join(HERE, "..", "tools", "micro", "libprimia_probe.so")
PROBE_SOURCES = ("conv3x3_c64.hip", "conv_s2lh.hip", "options.hip")


def build_probe(extra_flags=()):
    """tools/micro/libprimia_probe.so: the library with the timing-experiment switches compiled IN (-DPRIMIA_PROBE=1: the
    c64_dbg / s2lh_dbg options skip parts of a kernel — wrong results, phase timings only).  The shipped library refuses
    those options; tools load this one explicitly (bench.py --lib, `_lib.LIB_PATH = ...`).  Never loaded by the product."""
    build(verbose=False)
    objs = []
    for f in sorted(x for x in os.listdir(CSRC) if x.endswith(".hip")):
        if f in PROBE_SOURCES:
            os.makedirs(os.path.join(OBJ, "probe"), exist_ok=True)      # (its own directory: tools glob OBJ/*.o)
            obj = os.path.join(OBJ, "probe", f[:-4] + ".o")
            cmd = [_hipcc(), *FLAGS, "-DPRIMIA_PROBE=1", *extra_flags, "-c", os.path.join(CSRC, f), "-o", obj]
            r = subprocess.run(cmd, capture_output=True, text=True)
            if r.returncode != 0:
                raise RuntimeError(f"hipcc failed for {f} (probe):\n{r.stdout}{r.stderr}")
        else:
            obj = os.path.join(OBJ, f[:-4] + ".o")
        objs.append(obj)
    r = subprocess.run([_hipcc(), "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", PROBE_LIB, *objs],
                       capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("link failed:\n" + r.stdout + r.stderr)
    print(f"[primia build] linked {os.path.normpath(PROBE_LIB)}")
    return PROBE_LIB


if __name__ == "__main__":
    if "--probe" in sys.argv:
        build_probe()
    else:
        build(force="--force" in sys.argv)
