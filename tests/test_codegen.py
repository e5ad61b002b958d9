"""Guards the headline kernel's code generation: the planar SoftPendulum hot loop must stay
free of scratch (spill) traffic and within its instruction budget.  The register allocation
of this kernel has been knocked over before by unrelated edits to the 3-D fallback path
(profiles/README.md, r1g), which costs 15 % without failing any parity test."""
import re
import shutil
import sys
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
CSRC = ROOT / "gym_softrobot_amd" / "csrc"


def _loops(asm: str, mangled_substr: str):
    lines = asm.split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN7softrod\w*:", l) and mangled_substr in l)
    end = start
    while not lines[end].startswith(".Lfunc_end"):
        end += 1
    labels, ins = {}, []
    for l in lines[start:end + 1]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        t = l.strip()
        if l.startswith("\t") and t and not t.startswith((".", ";")):
            ins.append(t)
    out = []
    for i, t in enumerate(ins):
        m = re.match(r"s_cbranch\w*\s+(\.LBB\d+_\d+)|s_branch\s+(\.LBB\d+_\d+)", t)
        if m:
            tgt = labels.get(m.group(1) or m.group(2))
            if tgt is not None and tgt < i and i - tgt > 100:
                out.append(ins[tgt:i + 1])
    return out


@pytest.fixture(scope="module")
def isa_text(tmp_path_factory):
    """The gfx950 ISA of the whole library (one hipcc -S run for the module)."""
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not Path(hipcc).exists():
        pytest.skip("hipcc not available")
    asm = tmp_path_factory.mktemp("isa") / "capi.s"
    subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                    "-o", str(asm), str(CSRC / "softrod_capi.hip")], check=True, timeout=900,
                   stderr=subprocess.DEVNULL)
    return asm.read_text()


def test_planar_hot_loop_has_no_scratch_traffic(isa_text):
    text = isa_text
    loops = _loops(text, "fast_kernelILj15ELi1ELi1E")   # <SOFTPENDULUM, SOFTPENDULUM, EPL = 1>
    assert loops, "no loop found in the SoftPendulum step kernel"
    hot = loops[0]                                                   # the planar substep loop comes first
    assert not any(x.startswith("s_swappc") for x in hot), "a function call inside the hot loop"
    # what one substep executes when every range check passes (tools/hot_path_isa.py): 102 VALU
    # instructions as of r1j, 98 with the edge vectors carried as state; SQ_INSTS_VALU per rod-substep measures the same number on the GPU
    sys.path.insert(0, str(ROOT / "tools"))
    import hot_path_isa
    ins, labels = hot_path_isa.function_body(text, "fast_kernelILj15ELi1ELi1E")
    path = hot_path_isa.hot_path(ins, labels)
    valu = [x for x in path if x.startswith("v_")]
    assert len(valu) <= 102, f"planar substep grew to {len(valu)} VALU instructions"
    assert not [x for x in path if x.startswith(("scratch", "global", "buffer", "flat"))]
    assert not [x for x in valu if x.startswith("v_mov_b64")], "register copies inside the planar substep"
    # the whole kernel: nearly all of its scratch traffic brackets the out-of-line 3-D fallback call
    # (live values saved around s_swappc in the block a non-planar rod takes).  At 128 registers
    # (four rods resident per SIMD, softrod_fast.hpp ProgressPriority) a planar rod parks a few
    # values in scratch before the loop (the out-of-plane rows, for the paths that store them
    # back: 9 dwordx2) and a few loop constants: bounded here (r1k:
    # 692 B per lane were written on every launch; 70 dwords before the out-of-plane rows were
    # re-read instead of kept), and never inside a loop
    calls = [i for i, x in enumerate(ins) if "s_swappc" in x]
    assert len(calls) == 1
    bounds = sorted(set(labels.values()))
    lo = max(b for b in bounds if b <= calls[0])
    hi = min([b for b in bounds if b > calls[0]] + [len(ins)])
    stray = [i for i, x in enumerate(ins) if x.startswith("scratch") and not lo <= i < hi]
    stores = [i for i in stray if ins[i].startswith("scratch_store")]
    assert len(stores) <= 20, f"{len(stores)} scratch stores outside the fallback-call block"
    # (the hot-path walk above already holds the loop itself to zero memory instructions; the
    # out-of-range tiers inside the loop's address range may spill)


def test_3d_loops_keep_their_instruction_budget(isa_text):
    """The 3-D substep loops (OctoArmSingle, SoftPendulum3D, OctoFlat) are VALU-issue bound too; what
    round 2 took out of them — register copies behind two-address FMAs, selects on operands that
    already vanish, spilled scalar registers reloaded with v_readlane — comes back silently with an
    unrelated edit.  Budgets are the walked in-range paths at the end of round 2 plus a few per cent."""
    text = isa_text
    sys.path.insert(0, str(ROOT / "tools"))
    import hot_path_isa
    budgets = {   # kernel: (VALU, register copies, v_readlane, scratch loads)
        "fast_kernelILj1073742601ELi3ELi1ELb0E": (535, 8, 4, 0),     # OctoArmSingle: 517 / 3 / 2 (round 6: 508 / 3 / 0)
        "fast_kernelILj201ELi2ELi1ELb0E": (550, 10, 4, 0),           # SoftPendulum3D: 533 / 6 (round 6: 528 / 6 / 0)
        # OctoFlat, four envs per workgroup: 662 / 12 / 6; round 6: 653 / 12 / 0 and TWO scratch reloads on the in-range
        # path — the branch-free last tier of theta/sin(theta) moved the allocation; measured with the arms at rest
        # (tools/octo_ab.sh, --actions zero: 7.87 ms against 7.93 for round 5's tier) and curled (8.75 against 9.39)
        "octo_step_kernelILj1073743625ELi2ELi4E": (685, 16, 10, 2),
    }
    for key, (valu_max, copies_max, readlane_max, mem_max) in budgets.items():
        ins, labels = hot_path_isa.function_body(text, key)
        path = hot_path_isa.hot_path(ins, labels)
        valu = [x for x in path if x.startswith("v_")]
        assert len(valu) <= valu_max, f"{key}: {len(valu)} VALU instructions per substep"
        assert sum(x.startswith("v_mov_b64") for x in valu) <= copies_max, key
        assert sum(x.startswith("v_readlane") for x in valu) <= readlane_max, key
        mem = [x for x in path if x.startswith(("scratch", "global", "buffer", "flat"))]
        assert len(mem) <= mem_max and not [x for x in mem if not x.startswith("scratch_load")], f"{key}: memory traffic in the loop: {mem}"


def test_muscle_kernels_keep_their_parameter_struct_out_of_scratch(isa_text):
    """Round 6: ONE run-time index into RodParams (`fl_coef[p]` in the muscle law's Horner loop) kept the rigid-body muscle
    kernels' whole local parameter struct in private memory — 1592 B of scratch per lane with ZERO spilled VGPRs, a scratch
    load for every parameter the loop reads, 17-30 % of every muscle env's step time.  The tell is scratch that register
    spills do not explain; held here for the muscle instantiations."""
    def meta(mangled_substr):
        for m in re.finditer(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", isa_text, re.S):
            blk = m.group(0)
            if mangled_substr in re.search(r"\.name:\s+(\S+)", blk).group(1):
                g = lambda k: int(re.search(rf"\.{k}:\s+(\d+)", blk).group(1))      # noqa: E731
                return g("private_segment_fixed_size"), g("vgpr_spill_count"), g("vgpr_count")
        raise AssertionError(f"kernel {mangled_substr} not found")

    for key in ("octo_step_kernelILj13320ELi4ELi1E", "octo_step_kernelILj13320ELi2ELi1E"):      # the muscle octopus, OctoArmPullWeight
        scratch, spills, vgprs = meta(key)
        assert scratch == 0 and spills == 0 and vgprs > 256, (key, scratch, spills, vgprs)     # one wave per SIMD, nothing parked
    for key in ("fast_kernelILj12296ELi6ELi1ELb1E", "fast_kernelILj8216ELi0ELi1ELb0E"):         # OctoArmPush, the uniform muscle rod
        scratch, spills, _ = meta(key)
        assert scratch <= 4 * spills + 8 and scratch <= 512, (key, scratch, spills)            # spill slots only
