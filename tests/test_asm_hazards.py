"""The two inline-asm hazards the compiler does not guard (tools/check_asm_hazards.py): a vector-memory asm instruction reading an
SGPR that a vector instruction wrote fewer than 5 wait states earlier, and an asm store of more than 8 bytes per lane whose data
registers the next instruction overwrites (round 5: conv3x3_lh4's write-back stored the NEXT fragment's values in 2.5 % of a
tile's elements until its asm stores carried their own s_nop).  Checked on the ISA of the kernels that store or request through
asm, in parallel; no GPU needed."""
import concurrent.futures
import importlib.util
import os

import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
FILES = ["conv3x3_c64.hip", "conv3x3_lh2.hip", "conv3x3_lh4.hip", "conv_s2lh.hip", "conv_igemm.hip", "conv_wgrad_patch.hip",
         "stem_conv.hip", "stem_bwd_fused.hip", "stem_fwd_fused.hip", "conv_wgrad_tap.hip", "bn.hip", "gn.hip"]


def test_no_unguarded_asm_hazards_in_the_kernels():
    if not os.path.exists("/opt/rocm/bin/hipcc"):
        pytest.skip("no hipcc")
    spec = importlib.util.spec_from_file_location("check_asm_hazards", os.path.join(ROOT, "tools", "check_asm_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    paths = [os.path.join(ROOT, "primia_amd", "csrc", f) for f in FILES if os.path.exists(os.path.join(ROOT, "primia_amd", "csrc", f))]
    with concurrent.futures.ThreadPoolExecutor(max_workers=4) as ex:
        problems = [p for ps in ex.map(mod.check, paths) for p in ps]
    assert not problems, "\n".join(problems)
