"""Where the cooperative kernels' loops lie in the instruction cache's 32-byte fetch windows is pinned per wave role
and instantiation (step_coop.hip, coopCodePhase: `.p2align 5` + a measured number of `s_nop`; sweeps in
profiles/r04_phase_sweep*.txt: best against worst phase 1.5 - 5 %).  This test DISASSEMBLES the built library and checks
(1) that the code behind every pinned point starts at the measured offset and (2) -- for the instantiations the
workloads launch, the ones the sweeps measured -- that every LOOP HEAD behind a pinned point (the target of a backward
branch) still lies at the offset it had when the sweep was run: tests/golden/code_placement.json, written by
`python tests/test_code_placement.py --write` after a sweep.  A compiler bump (or an edit) that moves a loop head fails
here instead of silently costing up to 2.7 %; the remedy is tools/gpu_phase_sweep.sh for the kernel named, the new table in
coopCodePhase, and a regenerated fixture.  A pinned point is found by its marker, `s_nop 8 + role` (role: 0 carbon,
1 water, 2 light, 3 factor / soil wave, 4 the common prologue), which the compiler never emits."""
import json
import sys
import os
import re
import shutil
import subprocess

import pytest

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(REPO, "sipnet_amd", "libsipnet_amd.so")
LLVM = "/opt/rocm/lib/llvm/bin"

pytestmark = pytest.mark.skipif(not os.path.exists(os.path.join(LLVM, "llvm-objdump")), reason="needs ROCm's llvm-objdump")


def expected_phase(kernel, R, plain_exp, ring_lds, full, NP, ncyc, ext, role):
    """the table of step_coop.hip's coopCodePhase as measured (profiles/r04_phase_sweep.txt, _roles.txt), restated"""
    f64 = R == "double"
    if kernel == "stepCoopSumsKernel" and ring_lds and role in (0, 1):
        return 3          # round 6: profiles/r06_sums_phase_sweep.txt
    if role == 0:
        return 0 if ncyc else 7 if ext else 2 if NP == 2 else (4 if full else 6) if (NP == 1 and ring_lds) else 0
    if role == 1:
        return 0 if ncyc else 1 if ext else 4 if NP == 2 else (5 if full else 6) if (NP == 1 and ring_lds) else 0
    if role == 2:
        return 3 if (NP == 1 and ring_lds and not full and not ncyc and not ext) else 0
    if role == 3:
        return 4 if (ncyc and NP == 1) else 0
    if ncyc or ext:
        return 4
    if NP == 4:
        return 6 if f64 else 3
    if NP == 2:
        return 0 if full else 4 if f64 else 7
    if not ring_lds:
        return 0
    return 3 if (full or f64) else 4


def instantiation(demangled):
    """'void sipnet::stepCoopPairKernel<double, true, false>(sipnet::FastArgs)' -> template arguments of coopBody"""
    # (round 6: the in-kernel sums builds -- fp64, lean, default physics; placed like their plain relatives, not measured)
    m = re.match(r"void sipnet::stepCoopSumsKernel<(\w+), (\w+)>", demangled)
    if m:
        return dict(kernel="stepCoopSumsKernel", R="double", plain_exp=m.group(1) == "true", ring_lds=m.group(2) == "true", full=False,
                    NP=1, ncyc=False, ext=False)
    m = re.match(r"void sipnet::stepCoopPairSumsKernel<(\w+)>", demangled)
    if m:
        return dict(kernel="stepCoopPairSumsKernel", R="double", plain_exp=m.group(1) == "true", ring_lds=False, full=False, NP=2,
                    ncyc=False, ext=False)
    m = re.match(r"void sipnet::stepCoopXSumsKernel<(\w+), (\w+)>", demangled)
    if m:
        return dict(kernel="stepCoopXSumsKernel", R="double", plain_exp=m.group(1) == "true", ring_lds=m.group(2) == "true", full=False,
                    NP=1, ncyc=False, ext=True)
    m = re.match(r"void sipnet::stepCoopXPairSumsKernel<(\w+)>", demangled)
    if m:
        return dict(kernel="stepCoopXPairSumsKernel", R="double", plain_exp=m.group(1) == "true", ring_lds=False, full=False, NP=2,
                    ncyc=False, ext=True)
    m = re.match(r"void sipnet::stepCoopN(Pair)?SumsKernel<(\w+), (\w+)>", demangled)
    if m:
        return dict(kernel="stepCoopN%sSumsKernel" % (m.group(1) or ""), R="double", plain_exp=m.group(2) == "true", ring_lds=False,
                    full=False, NP=2 if m.group(1) else 1, ncyc=True, ext=m.group(3) == "true")
    # (step_coop_sums.hip: the sums builds of fp32-mixed batches on every layout and of the four-chunk layout; Layout = CoopLayout)
    m = re.match(r"void sipnet::sums2::stepCoopSumsAtKernel<(\w+), (\w+), (?:\(sipnet::CoopLayout\))?(\d), (\w+)>", demangled)
    if m:
        lay = int(m.group(3))
        return dict(kernel="stepCoopSumsAtKernel", R=m.group(1),     # (placed like their plain relatives: the fp64 one-chunk sums' own phases are not theirs)
                    plain_exp=m.group(2) == "true", ring_lds=lay == 0, full=False, NP={2: 2, 5: 2, 3: 4}.get(lay, 1), ncyc=lay in (4, 5),
                    ext=m.group(4) == "true")
    m = re.match(r"void sipnet::(?:bounded::)?(stepCoop\w*Kernel)<(\w+), (\w+)(?:, (\w+))?(?:, (\w+))?>", demangled)
    if not m:
        return None
    name, R = m.group(1), m.group(2)
    b = [x == "true" for x in m.groups()[2:] if x is not None]
    t = dict(kernel=name, R=R, plain_exp=b[0], ring_lds=False, full=False, NP=1, ncyc=False, ext=False)
    if name in ("stepCoopKernel", "stepCoopXKernel"):
        t.update(ring_lds=b[1], full=b[2], ext=name == "stepCoopXKernel")
    elif name in ("stepCoopPairKernel", "stepCoopXPairKernel"):
        t.update(full=b[1], NP=2, ext=name == "stepCoopXPairKernel")
    elif name in ("stepCoopQuadKernel", "stepCoopXQuadKernel"):
        t.update(NP=4, ext=name == "stepCoopXQuadKernel")
    elif name in ("stepCoopNKernel", "stepCoopNFullKernel", "stepCoopNXKernel", "stepCoopNXFullKernel"):
        t.update(ncyc=True, full="Full" in name, ext="NX" in name)
    elif name in ("stepCoopNPairKernel", "stepCoopNPairFullKernel", "stepCoopNXPairKernel", "stepCoopNXPairFullKernel",
                  "stepCoopNPairDiagKernel", "stepCoopNXPairDiagKernel"):       # (Diag: the full-state build with the counters, round 6)
        t.update(ncyc=True, NP=2, full="Full" in name or "Diag" in name, ext="NX" in name)
    else:
        return None
    return t


def disassemble(d):
    """-> {demangled kernel name: [(address, mnemonic, operand text, branch target offset in the kernel or None)]} of every
    product cooperative kernel; d: a scratch directory"""
    shutil.copyfile(LIB, d / "lib.so")
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=d, capture_output=True, timeout=300)
    out = {}
    for f in sorted(os.listdir(d)):
        if "gfx950" not in f:
            continue
        syms = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "-s", "--wide", str(d / f)], capture_output=True, text=True).stdout
        names = sorted({l.split()[-1] for l in syms.splitlines() if " FUNC " in l and "stepCoop" in l.split()[-1] and "bounded" not in l.split()[-1]})
        if not names:
            continue
        dem = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.splitlines()
        r = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--disassemble-symbols=" + ",".join(names), str(d / f)],
                           capture_output=True, text=True, timeout=900)
        cur = None
        for line in r.stdout.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
            if m:
                cur = dem[names.index(m.group(1))] if m.group(1) in names else None
                if cur is not None:
                    out[cur] = []
                continue
            if cur is None:
                continue
            m = re.match(r"^\s+(\S+)\s*(.*?)\s*//\s*([0-9A-Fa-f]+):(?:.*<\S+\+0x([0-9a-fA-F]+)>)?", line)
            if m:
                out[cur].append((int(m.group(3), 16), m.group(1), m.group(2), int(m.group(4), 16) if m.group(4) else None))
    assert len(out) >= 60, len(out)
    return out


@pytest.fixture(scope="module")
def disassembly(tmp_path_factory):
    return disassemble(tmp_path_factory.mktemp("codeobj"))


def pins(ins):
    """-> [(role, index of the marker, address where the pinned code starts)]: marker, alignment padding, phase nops"""
    out = []
    for k, (addr, op, args, _) in enumerate(ins):
        if op == "s_nop" and args.strip().isdigit() and 8 <= int(args) <= 12:
            out.append((int(args) - 8, k, addr))
    return out


def test_every_pinned_point_sits_at_its_measured_offset(disassembly):
    checked = 0
    families = set()
    for dem, ins in disassembly.items():
        t = instantiation(dem)
        assert t is not None, dem
        by_addr = {a: (op, args) for a, op, args, _ in ins}
        roles_seen = set()
        for role, k, marker in pins(ins):
            want = expected_phase(role=role, **t)
            boundary = (marker + 4 + 31) // 32 * 32
            # from behind the marker up to boundary + 4 * phase: nothing but padding
            for a in range(marker + 4, boundary + 4 * want, 4):
                assert by_addr.get(a) == ("s_nop", "0"), (dem, role, hex(a), by_addr.get(a))
            assert boundary + 4 * want in by_addr, (dem, role)
            checked += 1
            roles_seen.add(role)
        # every instantiation pins its prologue and its carbon, water and light waves
        assert {0, 1, 2, 4} <= roles_seen, (dem, roles_seen)
        families.add(t["kernel"])
    assert checked >= 4 * len(disassembly)
    assert {"stepCoopKernel", "stepCoopPairKernel", "stepCoopQuadKernel", "stepCoopNKernel", "stepCoopNPairKernel",
            "stepCoopXKernel"} <= families


# the instantiations the bench workloads launch (profiles/r04_phase_sweep.txt measured these)
MEASURED = [
    "void sipnet::stepCoopKernel<double, true, true, false>(sipnet::FastArgs)",        # c10k, c2, c2x16
    "void sipnet::stepCoopKernel<double, true, true, true>(sipnet::FastArgs)",         # ... with the record
    "void sipnet::stepCoopPairKernel<double, true, false>(sipnet::FastArgs)",          # c4
    "void sipnet::stepCoopQuadKernel<float, true>(sipnet::FastArgs)",                  # c3
    "void sipnet::stepCoopQuadKernel<double, true>(sipnet::FastArgs)",
    "void sipnet::stepCoopPairKernel<float, true, false>(sipnet::FastArgs)",
    "void sipnet::stepCoopKernel<float, true, true, false>(sipnet::FastArgs)",
    "void sipnet::stepCoopNKernel<double, true>(sipnet::FastArgs)",                    # c10kn
    "void sipnet::stepCoopNPairKernel<double, true>(sipnet::FastArgs)",                # c4n
    "void sipnet::stepCoopXKernel<double, true, true, false>(sipnet::FastArgs)",       # c10kr3
    "void sipnet::stepCoopXKernel<double, false, true, false>(sipnet::FastArgs)",
]
FIXTURE = os.path.join(REPO, "tests", "golden", "code_placement.json")


def loop_heads(ins):
    """-> {role: [offset mod 32 of every backward-branch target behind that role's pinned point, in address order]}"""
    start = ins[0][0]
    marks = sorted((k, role) for role, k, _ in pins(ins))
    out = {}
    for n, (k, role) in enumerate(marks):
        end = marks[n + 1][0] if n + 1 < len(marks) else len(ins)
        lo, hi = ins[k][0], ins[end - 1][0]
        heads = set()
        for a, op, args, tgt in ins[k:end]:
            if op.startswith(("s_cbranch", "s_branch")) and tgt is not None and lo <= start + tgt < a:
                heads.add(start + tgt)
        out[str(role)] = [h % 32 for h in sorted(heads)]
    return out


def signature(dis):
    return {dem: loop_heads(dis[dem]) for dem in MEASURED}


def test_loop_heads_lie_where_the_sweeps_measured_them(disassembly):
    for dem in MEASURED:
        assert dem in disassembly, dem
    want = json.load(open(FIXTURE))
    got = signature(disassembly)
    moved = [(dem, role) for dem in MEASURED for role in want[dem] if got[dem].get(role) != want[dem][role]]
    assert not moved, ("loop heads moved (compiler or source changed): re-run tools/gpu_phase_sweep.sh for these kernels / roles, "
                       "update coopCodePhase, then `python tests/test_code_placement.py --write`", moved)
    # (the time loops of the carbon, water and light waves are there: at least three loop heads each)
    for dem in MEASURED:
        for role in ("0", "1", "2"):
            assert len(got[dem][role]) >= 3, (dem, role, got[dem][role])


def test_the_measured_table_is_the_one_committed_in_profiles():
    """the headline instantiations against the sweep files themselves (best phase per kernel / role)"""
    # c10k's kernel: LDS ring, fp64, lean -> prologue 3, carbon 6, water 6, light 3 (profiles/r04_phase_sweep*.txt)
    t = dict(kernel="stepCoopKernel", R="double", plain_exp=True, ring_lds=True, full=False, NP=1, ncyc=False, ext=False)
    assert [expected_phase(role=r, **t) for r in (4, 0, 1, 2, 3)] == [3, 6, 6, 3, 0]
    t = dict(kernel="stepCoopPairKernel", R="double", plain_exp=True, ring_lds=False, full=False, NP=2, ncyc=False, ext=False)
    assert [expected_phase(role=r, **t) for r in (4, 0, 1)] == [4, 2, 4]
    t = dict(kernel="stepCoopQuadKernel", R="float", plain_exp=True, ring_lds=False, full=False, NP=4, ncyc=False, ext=False)
    assert [expected_phase(role=r, **t) for r in (4, 0, 1, 2)] == [3, 0, 0, 0]
    for f in ("r04_phase_sweep.txt", "r04_phase_sweep_roles.txt"):
        assert os.path.exists(os.path.join(REPO, "profiles", f))


# Scratch memory (register spills) of the cooperative kernels.  Which instantiation spills a few values of its general
# step moves with every edit (round 4: four of them, round 5: five others -- the register allocator's dice), so the rule is
# by ROLE, not by name: the LEAN kernels SIPNET_KERNEL_AUTO can pick -- every BASELINE workload's kernel among them --
# run out of registers and LDS alone: NO scratch instruction in their code (a private segment of a few bytes that no
# instruction touches -- the frame of SGPRs parked in VGPR lanes -- is tolerated; the measured kernels have none at all);
# the full-state builds and the instantiations only a forced SIPNET_KERNEL_COOP_HBM reaches (one chunk per workgroup with
# the ring in HBM) may keep up to 64 bytes per lane there.
def auto_reachable_lean(t):
    if t["full"]:
        return False
    if t["kernel"] in ("stepCoopKernel", "stepCoopXKernel"):
        return t["ring_lds"]                      # (one chunk per workgroup with the ring in HBM: forced only)
    return True


def test_scratch_memory_of_the_cooperative_kernels_is_pinned(tmp_path, disassembly):
    shutil.copyfile(LIB, tmp_path / "lib.so")
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=tmp_path, capture_output=True, timeout=300)
    scratch = {}
    for f in sorted(os.listdir(tmp_path)):
        if "gfx950" not in f:
            continue
        notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", str(tmp_path / f)], capture_output=True, text=True).stdout
        name = None
        for line in notes.splitlines():
            m = re.match(r"\s+\.name:\s+(\S+)", line)
            if m:
                name = m.group(1)
            m = re.match(r"\s+\.private_segment_fixed_size:\s+(\d+)", line)
            if m and name and "stepCoop" in name:
                scratch[name] = int(m.group(1))
    dem = subprocess.run(["c++filt"], input="\n".join(scratch), capture_output=True, text=True).stdout.splitlines()
    by_name = dict(zip(dem, scratch.values()))
    product = {k: v for k, v in by_name.items() if "bounded::" not in k}
    assert len(product) >= 80, len(product)
    lean = {k: v for k, v in product.items() if auto_reachable_lean(instantiation(k))}
    assert len(lean) >= 36, len(lean)
    for k, v in lean.items():
        key = k[len("void sipnet::"):k.rindex("(")] if k.startswith("void sipnet::") else k
        ins = disassembly[[d for d in disassembly if key in d][0]]
        touching = [op for _, op, _, _ in ins if op.startswith("scratch_") or op.startswith("buffer_")]
        if "stepCoopSumsAtKernel<double" in k:
            # the fp64 four-chunk sums build (168 registers, three wavefronts per SIMD) keeps ONE dword in scratch: read and written
            # once per group of steps, where the group's sum is stored (profiles/r06_sums_time.txt: 12.74 ms, the planes' build 12.92)
            assert v <= 8 and len(touching) <= 8, (k, v, touching)
            continue
        assert not touching and v <= 64, (k, v, touching[:4])
    assert max(product.values()) <= 64, {k: v for k, v in product.items() if v > 64}
    for k in MEASURED:
        assert product[k] == 0, k
    # the bounded-wait build (test-only kernels: a poll counter per wait) is held to a small budget, not to zero
    assert max(v for k, v in by_name.items() if "bounded::" in k) <= 64


if __name__ == "__main__":
    if "--write" in sys.argv:
        import tempfile
        with tempfile.TemporaryDirectory() as d:
            sig = signature(disassemble(__import__("pathlib").Path(d)))
        json.dump(sig, open(FIXTURE, "w"), indent=1, sort_keys=True)
        print("wrote", FIXTURE, {k.split("sipnet::")[1].split("(")[0]: {r: len(v) for r, v in s_.items()} for k, s_ in sig.items()})
