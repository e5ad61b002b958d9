#!/usr/bin/env python3
"""Counts the instructions one substep of a step kernel actually executes when every
range-reduction check passes (the normal case): walks the kernel's ISA from the substep
loop's header round to the header again, taking at each wave-uniform branch the side the
in-range case takes (vcc == 0 after an `any lane out of range` compare, exec != 0, loop
not finished).  This is the number SQ_INSTS_VALU / (rods x substeps) measures
(tools/pmc_valu_per_substep.sh).

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -S --cuda-device-only -o capi.s softrod_capi.hip
  python tools/hot_path_isa.py capi.s fast_kernelILj15ELi1ELi1E [-v]
"""
import re
import sys


def function_body(asm: str, name: str):
    lines = asm.split("\n")
    start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN7softrod\w*:", l) and name in l)
    end = start
    while not lines[end].startswith(".Lfunc_end"):      # (a kernel may hold several s_endpgm)
        end += 1
    labels, ins = {}, []
    for l in lines[start:end + 1]:
        m = re.match(r"^(\.LBB\d+_\d+):", l)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        t = l.strip()
        if l.startswith("\t") and t and not t.startswith((".", ";")):
            ins.append(t.split(";")[0].strip())
    return ins, labels


def hot_path(ins, labels):
    # the substep loop: the first of the innermost backward branches whose span holds the substep's
    # v_rsq_f64 AND its DPP wave shifts (the kernel has other loops: the clock fallback, epilogue reductions, the
    # OctoFlat epilogue's crossing count — which holds a square root of its own since round 6's builds)
    head, best = None, None
    for i, t in enumerate(ins):
        m = re.match(r"s_c?branch\w*\s+(\.LBB\d+_\d+)", t)
        if m and labels.get(m.group(1), i) < i - 60:
            lo = labels[m.group(1)]
            span = ins[lo:i]
            if (any(x.startswith("v_rsq_f64") for x in span) and any("_dpp" in x for x in span)
                    and (best is None or i - lo < best)):
                head, best = lo, i - lo
    if head is None:
        raise SystemExit("no loop found")
    path, pc, seen, first = [], head, 0, {}
    const, vcc, scc = {}, None, None    # s[a:b] pairs holding a known 0 / -1; vcc known zero / non-zero (exec != 0); scc
    while True:
        t = ins[pc]
        m = re.match(r"(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", t)
        mm = re.match(r"s_mov_b64\s+(s\[\d+:\d+\]),\s*(-1|0)$", t)
        ma = re.match(r"(s_and_b64|s_andn2_b64)\s+vcc,\s*exec,\s*(s\[\d+:\d+\])", t)
        if mm:
            const[mm.group(1)] = mm.group(2) == "-1"
        elif re.match(r"s_\w+\s+(s\[\d+:\d+\])", t) and not ma:
            const.pop(re.match(r"s_\w+\s+(s\[\d+:\d+\])", t).group(1), None)
        if ma and ma.group(2) in const:       # a flag the compiler carries instead of branching twice
            vcc = const[ma.group(2)] if ma.group(1) == "s_and_b64" else not const[ma.group(2)]
        elif t.startswith("v_cmp") or (t.startswith("s_") and " vcc" in t.split(",")[0]):
            vcc = None
        if pc in first:          # back where the walk has been: the cycle from there on is the substep
            return path[first[pc]:]
        first[pc] = len(path)
        # a range check ANDed with a validity mask on the scalar side: `s_cmp_eq_u64 x, 0` / `s_cmp_lg_u64
        # x, 0` on the lanes out of range -- none, on this path
        mz = re.match(r"s_cmp_(eq|lg)_u64\s+\S+,\s*0$", t)
        if mz:
            scc = mz.group(1) == "eq"
        elif t.startswith("s_cmp") or t.startswith("s_add") or t.startswith("s_sub"):
            scc = None
        if m:
            kind, tgt = m.group(1), labels[m.group(2)]
            if kind in ("s_cbranch_scc0", "s_cbranch_scc1") and scc is not None:
                taken = (kind == "s_cbranch_scc1") == scc
                path.append(t)
                pc = tgt if taken else pc + 1
                seen += 1
                if seen > 5000:
                    raise SystemExit("did not return to the loop header")
                continue
            if kind in ("s_cbranch_vccz", "s_cbranch_vccnz") and vcc is not None:
                taken = (kind == "s_cbranch_vccnz") == vcc
            elif kind == "s_cbranch_execnz" and pc - 30 < tgt <= pc:
                taken = False      # a spin-wait (the flag rendezvous of the OctoFlat kernel): met at once
            else:
                taken = kind in ("s_branch", "s_cbranch_vccz", "s_cbranch_execnz")
            path.append(t)
            pc = tgt if taken else pc + 1
        else:
            path.append(t)
            pc += 1
        seen += 1
        if seen > 5000:
            raise SystemExit("did not return to the loop header")


def main():
    asm = open(sys.argv[1]).read()
    ins, labels = function_body(asm, sys.argv[2])
    path = hot_path(ins, labels)
    valu = [x for x in path if x.startswith("v_")]
    dpp = [x for x in valu if "dpp" in x]
    print(f"{sys.argv[2]}: {len(path)} instructions per substep, {len(valu)} VALU "
          f"({len(dpp)} DPP moves, {sum(x.startswith('v_mov_b64') for x in valu)} v_mov_b64, "
          f"{sum(x.startswith('v_cndmask') for x in valu)} v_cndmask), "
          f"{sum(x.startswith(('scratch', 'buffer', 'global', 'flat')) for x in path)} memory")
    if "-v" in sys.argv:
        for x in path:
            print("   ", x.replace(" row_mask:0xf bank_mask:0xf bound_ctrl:1", ""))


if __name__ == "__main__":
    main()
