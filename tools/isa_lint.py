#!/usr/bin/env python3
"""ISA lint of the hand-scheduled scan kernels (CPU only: hipcc cross-compiles gfx950 without a GPU).

The tiled scan issues `s_load_dword*` from inline asm (`load_qset`, `warm_query_lines` in csrc/scan_bucket.hip).  The
compiler believes their outputs are written when the asm statement ends; the data lands later, at a hand-placed
`s_waitcnt lgkmcnt(0)`.  That is correct only while nothing reads, copies or reassigns those SGPRs in between -- r02 hit
exactly this (a throw-away destination was reused, the late write corrupted a live value, the kernel faulted).  This
lint pins the property on the compiled code instead of on the source:

  for every kernel whose name matches, over its control-flow graph (basic blocks, branch edges, fixpoint), no instruction
  may name an SGPR that a scalar load issued on ANY path since the last `s_waitcnt lgkmcnt(0)` still owns.

Scalar loads return out of order, so only `lgkmcnt(0)` releases them.  One idiom is allowed: several ONE-dword loads
into the same destination (the scalar-cache warm-up: the value is never read), i.e. a write-after-write between
`s_load_dword` instructions; every other overlap (a read of an in-flight register by any instruction, a wider load
landing on one) is a violation.  Resource figures come from `-Rpass-analysis=kernel-resource-usage` of the same compile.

    python tools/isa_lint.py [--extra=-DNLSH_TILED_KB=8 ...] [--match bscan3_kernel] [--src scan_bucket.hip]
"""
import argparse
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "neural-locality-sensitive-hashing_amd", "csrc")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
# the flags of csrc/Makefile that shape code generation
BASE_FLAGS = ["-O3", "-std=c++17", "--offload-arch=gfx950", "-fno-gpu-rdc", "-ffp-contract=off"]

_SREG = re.compile(r"\bs(\d+)\b|\bs\[(\d+):(\d+)\]")
_LABEL = re.compile(r"^([.\w$]+):")
_WAIT_LGKM0 = re.compile(r"lgkmcnt\(0\)")
_BRANCH = re.compile(r"^s_(branch|cbranch_\w+)\s+([.\w$]+)")


def compile_asm(src="scan_bucket.hip", extra=()):
    """-> (asm text, remarks text) of the device code of csrc/<src> for gfx950."""
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        cmd = [HIPCC, *BASE_FLAGS, *extra, "-S", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage",
               os.path.join(CSRC, src), "-o", out]
        p = subprocess.run(cmd, capture_output=True, text=True)
        if p.returncode != 0:
            raise RuntimeError("hipcc failed:\n" + p.stderr[-4000:])
        return open(out).read(), p.stderr


def resources(remarks):
    """{mangled kernel name: {"VGPRs": n, "TotalSGPRs": n, "ScratchSize": n, "Occupancy": n, "SGPRs Spill": n, "VGPRs Spill": n, "LDS Size": n}}"""
    res, cur = {}, None
    for line in remarks.splitlines():
        m = re.search(r"remark:\s+Function Name: (\S+)", line)
        if m:
            cur = res.setdefault(m.group(1), {})
            continue
        m = re.search(r"remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)", line)
        if m and cur is not None:
            cur[m.group(1).strip()] = int(m.group(2))
    return res


def functions(asm):
    """{name: [instruction / label lines]} for every function of the module (text between `name:` and `.Lfunc_end`)."""
    out, cur, name = {}, None, None
    for raw in asm.splitlines():
        line = raw.split(";")[0].rstrip()
        if not line.strip():
            continue
        if cur is None:
            m = re.match(r"^(_Z\w+):\s*$", line)
            if m:
                name, cur = m.group(1), []
            continue
        if line.startswith(".Lfunc_end"):
            out[name] = cur
            cur = None
            continue
        s = line.strip()
        if s.startswith(".") and not _LABEL.match(s):
            continue                      # directives
        cur.append(s)
    return out


def _sregs(text):
    regs = set()
    for m in _SREG.finditer(text):
        if m.group(1) is not None:
            regs.add(int(m.group(1)))
        else:
            regs.update(range(int(m.group(2)), int(m.group(3)) + 1))
    return regs


def _blocks(lines):
    """Basic blocks: [(label | None, [instructions])], plus {label: block index}."""
    blocks, cur, label = [], [], None
    for s in lines:
        m = _LABEL.match(s)
        if m:
            if cur or label is not None:
                blocks.append((label, cur))
            label, cur = m.group(1), []
            continue
        cur.append(s)
        if _BRANCH.match(s) or s.startswith("s_endpgm") or s.startswith("s_setpc"):
            blocks.append((label, cur))
            label, cur = None, []
    if cur or label is not None:
        blocks.append((label, cur))
    index = {lab: i for i, (lab, _) in enumerate(blocks) if lab is not None}
    return blocks, index


def _successors(blocks, index, i):
    ins = blocks[i][1]
    last = ins[-1] if ins else ""
    m = _BRANCH.match(last)
    succ = []
    if m:
        if m.group(2) in index:
            succ.append(index[m.group(2)])
        if m.group(1) != "branch" and i + 1 < len(blocks):
            succ.append(i + 1)
    elif last.startswith("s_endpgm") or last.startswith("s_setpc"):
        pass
    elif i + 1 < len(blocks):
        succ.append(i + 1)
    return succ


def _transfer(ins_list, state, report=None):
    """state: {sgpr: is_one_dword_load}.  Walks one block; returns the state at its end."""
    state = dict(state)
    for s in ins_list:
        op = s.split()[0]
        if op.startswith("s_waitcnt"):
            if _WAIT_LGKM0.search(s) or re.match(r"^s_waitcnt\s+(0|0x0)\s*$", s):
                state.clear()
            continue
        if op.startswith("s_load_dword") or op.startswith("s_buffer_load_dword"):
            ops = s[len(op):].split(",")
            dest, srcs = _sregs(ops[0]), _sregs(",".join(ops[1:]))
            one = op in ("s_load_dword", "s_buffer_load_dword")
            bad = (srcs & state.keys()) | {r for r in dest if r in state and not (one and state[r])}
            if bad and report is not None:
                report.append((s, sorted(bad)))
            for r in dest:
                state[r] = one
            continue
        if state:
            bad = _sregs(s) & state.keys()
            if bad and report is not None:
                report.append((s, sorted(bad)))
    return state


def inflight_violations(lines):
    """[(instruction, [sgprs])] that touch an SGPR owned by a scalar load in flight on some path."""
    blocks, index = _blocks(lines)
    n = len(blocks)
    state_in = [dict() for _ in range(n)]
    work = list(range(n))
    while work:
        i = work.pop()
        out = _transfer(blocks[i][1], state_in[i])
        for j in _successors(blocks, index, i):
            merged = dict(state_in[j])
            changed = False
            for r, one in out.items():
                if r not in merged or (merged[r] and not one):
                    merged[r] = one if r not in merged else False
                    changed = True
            if changed:
                state_in[j] = merged
                work.append(j)
    report = []
    for i in range(n):
        _transfer(blocks[i][1], state_in[i], report)
    return report


def spills_in_hot_loops(lines, hot="v_fmac_f32"):
    """SGPR spill traffic (v_writelane_b32, or a v_readlane_b32 of a constant lane into an SGPR: the compiler's spill / reload
    forms) inside any loop whose body holds the distance arithmetic.  Loop = a backward branch to a label; body = the lines between.
    Spills elsewhere (kernel arguments parked in a VGPR at the prologue, reloaded in the epilogue) cost nothing that matters."""
    pos = {}
    for i, s_ in enumerate(lines):
        m = _LABEL.match(s_)
        if m:
            pos[m.group(1)] = i
    # the VGPRs the compiler spills SGPRs into = the destinations of the function's v_writelane_b32 (a v_readlane of a constant
    # lane from any other VGPR is ordinary code: wave reductions read lane 63)
    spill_vgprs = {x.split()[1].rstrip(",") for x in lines if x.startswith("v_writelane_b32")}
    bad = []
    for i, s_ in enumerate(lines):
        m = _BRANCH.match(s_)
        if m and m.group(2) in pos and pos[m.group(2)] < i:
            body = lines[pos[m.group(2)]:i]
            if any(x.startswith(hot) for x in body):
                for x in body:
                    r = re.match(r"^v_readlane_b32 s\d+, (v\d+), \d+$", x)
                    if x.startswith("v_writelane_b32") or (r and r.group(1) in spill_vgprs):
                        bad.append(x)
    return bad


def lint(src="scan_bucket.hip", match="bscan3_kernel", extra=()):
    """-> {kernel: {"resources": {...}, "violations": [...], "scalar_loads": n, "writelanes": n, "hot_loop_spills": [...]}}"""
    asm, remarks = compile_asm(src, extra)
    res = resources(remarks)
    out = {}
    for name, lines in functions(asm).items():
        if match not in name:
            continue
        out[name] = {"resources": res.get(name, {}), "violations": inflight_violations(lines),
                     "scalar_loads": sum(1 for s in lines if s.startswith("s_load_dword")),
                     "writelanes": sum(1 for s in lines if s.startswith("v_writelane")),
                     "hot_loop_spills": spills_in_hot_loops(lines)}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--src", default="scan_bucket.hip")
    ap.add_argument("--match", default="bscan3_kernel")
    ap.add_argument("--extra", action="append", default=[])
    args = ap.parse_args()
    rep = lint(args.src, args.match, tuple(args.extra))
    rc = 0
    for name, r in rep.items():
        print(name, r["resources"], f"scalar loads {r['scalar_loads']}, v_writelane {r['writelanes']} ({len(r['hot_loop_spills'])} in hot loops), violations {len(r['violations'])}")
        for ins, regs in r["violations"][:20]:
            print("   ", ins, "<- in flight:", regs)
        rc |= bool(r["violations"]) or any(x.startswith("v_writelane") for x in r["hot_loop_spills"])
    if not rep:
        print("no kernel matches", args.match)
        rc = 1
    sys.exit(rc)


if __name__ == "__main__":
    main()
