"""The inline-asm hazards the compiler does not guard (tools/check_asm_hazards.py): a vector-memory asm instruction reading an
SGPR that a vector instruction wrote fewer than 5 wait states earlier; an asm store of more than 8 bytes per lane whose data
registers the next instruction overwrites (round 5: conv3x3_lh4's write-back stored the NEXT fragment's values in 2.5 % of a
tile's elements until its asm stores carried their own s_nop); M0 written right before an asm LDS-DMA instruction; a register
the asm LOADED into, touched before a wait retired the load.  Checked on the ISA the build kept for the objects it links
(csrc/_build/isa/*.s), following fall-through across labels and branch back-edges; no GPU needed.  The checker itself is held
to hand-written cases of every class — hazard present / hazard cured — so that "no hazards found" means something."""
import importlib.util
import os

import pytest

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")


def _checker():
    spec = importlib.util.spec_from_file_location("check_asm_hazards", os.path.join(ROOT, "tools", "check_asm_hazards.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_no_unguarded_asm_hazards_in_the_built_objects():
    """Every object of the shipped library: the ISA is the one `primia_amd.build` kept while compiling that object."""
    from primia_amd import build

    build.build(verbose=False)          # (no-op when the tree is built: what __graft_entry__.build() leaves behind)
    mod = _checker()
    names = mod.built_objects()
    assert len(names) >= 25 and "conv3x3_c64" in names and "conv_s2lh" in names
    problems = [p for n in names for p in mod.check(n)]
    assert not problems, "\n".join(problems)
    # it looked at something: the kernels that request and store through asm are in there
    ins, _ = mod.parse(mod.isa_path("conv3x3_c64")[0])
    asm = [t for t, a in ins if a]
    assert sum("store_dwordx4" in t for t in asm) >= 8 and sum(t.endswith(" lds") for t in asm) >= 50


ASM = "\t;;#ASMSTART\n\t{}\n\t;;#ASMEND\n"
CASES = [
    # (a) VALU-written SGPR read by an asm buffer instruction
    ("sgpr close", "v_readfirstlane_b32 s4, v1\ns_nop 1\n" + ASM.format("buffer_load_dwordx4 v2, s[8:11], s4 offen lds"), 1),
    ("sgpr far enough", "v_readfirstlane_b32 s4, v1\ns_nop 4\n" + ASM.format("buffer_load_dwordx4 v2, s[8:11], s4 offen lds"), 0),
    ("sgpr carry-out pair", "v_add_co_u32_e64 v1, s[8:9], v2, v3\n" + ASM.format("buffer_store_dword v2, v3, s[8:11], 0 offen"), 1),
    ("sgpr across a label (fall-through)",
     "v_cmp_gt_i32_e64 s[4:5], s25, v51\n.LBB0_3:\ns_nop 0\n" + ASM.format("buffer_load_dword v2, v3, s[4:7], 0 offen"), 1),
    ("sgpr through a back-edge",
     ".LBB0_1:\n" + ASM.format("buffer_load_dwordx4 v2, s[8:11], s4 offen lds")
     + "v_add_u32_e32 v2, 1, v2\nv_readfirstlane_b32 s4, v9\ns_cbranch_scc1 .LBB0_1\ns_endpgm\n", 1),
    ("no path: writer behind an unconditional branch",
     "v_readfirstlane_b32 s4, v1\ns_branch .LBB0_9\n" + ASM.format("buffer_load_dword v2, v3, s[4:7], 0 offen") + ".LBB0_9:\ns_endpgm\n", 0),
    # (b) wide asm store, data overwritten
    ("store data overwritten next", ASM.format("buffer_store_dwordx4 v[4:7], v1, s[8:11], 0 offen") + "v_mov_b32_e32 v5, 0\n", 1),
    ("store data overwritten by a swap's second operand",
     ASM.format("buffer_store_dwordx4 v[4:7], v1, s[8:11], 0 offen") + "v_permlane16_swap_b32_e32 v20, v6\n", 1),
    ("store followed by its own nop", ASM.format("buffer_store_dwordx4 v[4:7], v1, s[8:11], 0 offen\n\ts_nop 1") + "v_mov_b32_e32 v5, 0\n", 0),
    ("store data overwritten at a branch target",
     ASM.format("buffer_store_dwordx4 v[4:7], v1, s[8:11], 0 offen") + "s_cbranch_vccz .LBB0_2\ns_nop 3\ns_endpgm\n.LBB0_2:\nv_mov_b32_e32 v4, 0\n", 1),
    ("narrow store is interlocked", ASM.format("buffer_store_dwordx2 v[4:5], v1, s[8:11], 0 offen") + "v_mov_b32_e32 v5, 0\n", 0),
    # (c) M0
    ("m0 written directly before the lds load", ASM.format("s_mov_b32 m0, s2\n\tbuffer_load_dwordx4 v2, s[36:39], s62 offen lds"), 1),
    ("m0 with its wait state", ASM.format("s_mov_b32 m0, s2\n\ts_nop 0\n\tbuffer_load_dwordx4 v2, s[36:39], s62 offen lds"), 0),
    ("m0 written by the compiler right before an asm global_load_lds",
     "s_add_u32 m0, s3, 64\n" + ASM.format("global_load_lds_dwordx4 v[2:3], off"), 1),
    # (d) registers an asm load writes
    ("asm load consumed without a wait",
     ASM.format("buffer_load_dword v7, v1, s[8:11], 0 offen") + "v_mov_b32_e32 v30, v7\ns_waitcnt vmcnt(0)\n", 1),
    ("asm load, counted wait too loose",
     ASM.format("buffer_load_dword v7, v1, s[8:11], 0 offen") + "buffer_load_dword v8, v1, s[8:11], 0 offen offset:4\n"
     "buffer_load_dword v9, v1, s[8:11], 0 offen offset:8\ns_waitcnt vmcnt(3)\nv_add_u32_e32 v7, v7, v7\n", 1),
    ("asm load, counted wait exact",
     ASM.format("buffer_load_dword v7, v1, s[8:11], 0 offen") + "buffer_load_dword v8, v1, s[8:11], 0 offen offset:4\n"
     "buffer_load_dword v9, v1, s[8:11], 0 offen offset:8\ns_waitcnt vmcnt(2)\nv_add_u32_e32 v7, v7, v7\n", 0),
    ("asm ds_read into the data of an asm store, lgkm wait missing",
     ASM.format("ds_read_b128 v[4:7], v1") + "s_waitcnt vmcnt(0)\n" + ASM.format("buffer_store_dwordx4 v[4:7], v1, s[8:11], 0 offen\n\ts_nop 1"), 1),
    ("asm ds_read then the right wait",
     ASM.format("ds_read_b128 v[4:7], v1") + "s_waitcnt lgkmcnt(0)\n" + ASM.format("buffer_store_dwordx4 v[4:7], v1, s[8:11], 0 offen\n\ts_nop 1"), 0),
    ("asm load consumed in the next iteration of a loop",
     ".LBB0_1:\nv_mov_b32_e32 v30, v7\n" + ASM.format("buffer_load_dword v7, v1, s[8:11], 0 offen") + "s_cbranch_scc1 .LBB0_1\ns_waitcnt vmcnt(0)\ns_endpgm\n", 1),
]


@pytest.mark.parametrize("name,text,expected", CASES, ids=[c[0] for c in CASES])
def test_checker_finds_each_hazard_class_and_accepts_the_cure(tmp_path, name, text, expected):
    mod = _checker()
    path = tmp_path / "k.s"
    path.write_text("kernel:\n" + "\n".join("\t" + l.strip() if not l.strip().endswith(":") else l.strip() for l in text.splitlines())
                    + "\n\ts_endpgm\n")
    problems = mod.check_isa(str(path), "k.hip")
    assert (len(problems) > 0) == bool(expected), problems
