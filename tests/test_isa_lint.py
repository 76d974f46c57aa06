"""CPU-side ISA lint of the tiled scan kernels (tools/isa_lint.py): the asynchronous scalar loads issued from inline asm
may not be touched before the hand-placed `s_waitcnt lgkmcnt(0)`, on any path of the compiled control-flow graph, and the
kernels keep the register / LDS budget their occupancy depends on.  No GPU needed: hipcc cross-compiles gfx950."""
import os
import shutil
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import isa_lint  # noqa: E402

needs_hipcc = pytest.mark.skipif(not (os.path.exists(isa_lint.HIPCC) or shutil.which("hipcc")), reason="hipcc not installed")


def test_lint_flags_a_read_of_an_inflight_register():
    bad = """
        s_load_dwordx4 s[20:23], s[4:5], 0x0
        v_sub_f32 v1, s21, v2
        s_waitcnt lgkmcnt(0)
        s_endpgm
    """.split("\n")
    rep = isa_lint.inflight_violations([s.strip() for s in bad if s.strip()])
    assert len(rep) == 1 and rep[0][1] == [21]


def test_lint_follows_branches_and_back_edges():
    # the load is issued at the END of the loop body; the use sits at its top: only the back edge connects them
    loop = """
        s_mov_b32 s9, 0
        .LBB0_1:
        v_add_f32 v1, s30, v1
        s_waitcnt lgkmcnt(0)
        s_load_dword s30, s[4:5], s9
        s_add_i32 s9, s9, 4
        s_cmp_lt_i32 s9, 64
        s_cbranch_scc1 .LBB0_1
        s_waitcnt lgkmcnt(0)
        s_endpgm
    """.split("\n")
    rep = isa_lint.inflight_violations([s.strip() for s in loop if s.strip()])
    assert [r[1] for r in rep] == [[30]]
    # a partial wait does not release scalar loads (they return out of order)
    partial = ["s_load_dwordx2 s[10:11], s[4:5], 0x0", "s_waitcnt lgkmcnt(1)", "s_mov_b32 s12, s10", "s_waitcnt lgkmcnt(0)", "s_endpgm"]
    assert len(isa_lint.inflight_violations(partial)) == 1


def test_lint_finds_spill_traffic_only_inside_arithmetic_loops():
    prologue = ["v_writelane_b32 v68, s8, 0", ".LBB0_1:", "v_fmac_f32 v1, v2, v2", "s_cbranch_scc1 .LBB0_1", "v_readlane_b32 s8, v68, 0", "s_endpgm"]
    assert isa_lint.spills_in_hot_loops(prologue) == []
    hot = ["v_writelane_b32 v68, s8, 3", ".LBB0_1:", "v_fmac_f32 v1, v2, v2", "v_readlane_b32 s8, v68, 3", "s_cbranch_scc1 .LBB0_1", "s_endpgm"]
    assert isa_lint.spills_in_hot_loops(hot) == ["v_readlane_b32 s8, v68, 3"]
    reduction = [".LBB0_1:", "v_fmac_f32 v1, v2, v2", "v_readlane_b32 s8, v24, 63", "s_cbranch_scc1 .LBB0_1", "s_endpgm"]   # ordinary code, not a reload
    assert isa_lint.spills_in_hot_loops(reduction) == []
    other = [".LBB0_1:", "v_add_f32 v1, v2, v2", "v_writelane_b32 v68, s8, 3", "s_cbranch_scc1 .LBB0_1", "s_endpgm"]     # a loop without the arithmetic
    assert isa_lint.spills_in_hot_loops(other) == []


def test_lint_allows_only_the_one_dword_warm_up_idiom():
    warm = ["s_load_dword s40, s[4:5], s9", "s_load_dword s40, s[6:7], s9", "s_waitcnt lgkmcnt(0)", "s_mov_b32 s40, 0", "s_endpgm"]
    assert isa_lint.inflight_violations(warm) == []
    wide = ["s_load_dwordx4 s[40:43], s[4:5], s9", "s_load_dwordx4 s[40:43], s[6:7], s9", "s_waitcnt lgkmcnt(0)", "s_endpgm"]
    assert len(isa_lint.inflight_violations(wide)) == 1
    mixed = ["s_load_dword s40, s[4:5], s9", "s_load_dwordx4 s[40:43], s[6:7], s9", "s_waitcnt lgkmcnt(0)", "s_endpgm"]
    assert len(isa_lint.inflight_violations(mixed)) == 1


@needs_hipcc
def test_shipped_tiled_scan_kernels_are_clean():
    rep = isa_lint.lint("scan_bucket.hip", "bscan3_kernel")
    assert len(rep) == 3, sorted(rep)                      # L2 (exact), L2 (folded eps, opt-in) and cosine instantiations
    for name, r in rep.items():
        assert r["scalar_loads"] > 100, name               # the hand-placed loads are really in there
        assert r["violations"] == [], (name, r["violations"][:5])
        res = r["resources"]
        # what the 7 waves per SIMD rest on: no scratch, no VGPR spills, the register budget, the 20 KB tile + the
        # in-scan merge's 2 KB scratch (7 workgroups x 22.5 KB fit the CU's 160 KB).  SGPR spills are tolerated where they are
        # harmless -- kernel arguments the epilogue needs, parked in a VGPR's lanes at the prologue -- and nowhere else: none inside a
        # loop that holds the distance arithmetic (a v_writelane of an in-flight register would also be a violation above).
        assert res["ScratchSize"] == 0 and res["VGPRs Spill"] == 0, (name, res)
        # (one 64-bit pointer of one task body is RE-loaded from its spill lanes once per k-block by this compiler: two instructions
        # per ~700, and a reload into an in-flight register would be a violation above; a spill WRITE in there is refused outright)
        assert not [x for x in r["hot_loop_spills"] if x.startswith("v_writelane")], (name, r["hot_loop_spills"][:5])
        assert len(r["hot_loop_spills"]) <= 4, (name, r["hot_loop_spills"][:8])
        assert res["SGPRs Spill"] <= 32, (name, res)
        assert res["Occupancy"] >= 7 and res["VGPRs"] <= 72, (name, res)
        assert res["LDS Size"] <= 22528, (name, res)


@needs_hipcc
def test_diagnostic_switches_exist_only_under_nlsh_diag():
    """The NLSH_ABLATE / NLSH_NO_STAGE_BARRIER builds remove pieces of the kernel to time the rest: wrong results by design.  They live
    behind -DNLSH_DIAG (csrc/scan_common.h): a stray -DNLSH_ABLATE on the shipped flags must not compile, and the diagnostic build itself
    must keep compiling (compile-only: the ablation code does not rot unseen)."""
    import subprocess
    src = os.path.join(isa_lint.CSRC, "scan_bucket.hip")
    base = [isa_lint.HIPCC, *isa_lint.BASE_FLAGS, "--cuda-device-only", "-c", src, "-o", os.devnull]
    stray = subprocess.run(base + ["-DNLSH_ABLATE=5"], capture_output=True, text=True)
    assert stray.returncode != 0 and "diagnostic builds only" in stray.stderr
    diag = subprocess.run(base + ["-DNLSH_DIAG", "-DNLSH_ABLATE=13", "-DNLSH_NO_STAGE_BARRIER=1"], capture_output=True, text=True)
    assert diag.returncode == 0, diag.stderr[-2000:]
