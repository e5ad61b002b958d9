#!/usr/bin/env python3
"""Markdown rows of DESIGN.md section 5 / profiles/README.md from the per-workload summaries of one profile set:
    python tools/profile_table.py r6
kernel ms = the timed windows' launches (kernel trace); frac = SQ_INSTS_VALU per launch / kernel time / 614.4 G wave-instr/s
(every instruction priced at 4 cycles of the 2.4 GHz peak clock); busy = SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x
1024 SIMDs) (= roofline.frac_cycle_weighted, capped at 1 there); HBM = FETCH_SIZE x 2 + WRITE_SIZE per launch."""
import glob
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
PEAK = 1024 * 2.4e9 / 4.0


def main(tag):
    rows = []
    for p in sorted(glob.glob(str(ROOT / "profiles" / f"{tag}_*.json"))):
        d = json.load(open(p))
        if "pmc3" not in d or "workload" not in d:
            continue
        ms = d["step_kernel_timed_avg_ms"]
        pm = d["pmc3"]
        frac = pm["SQ_INSTS_VALU_per_env_step"] / (ms * 1e-3) / PEAK
        lines = d.get("bench_lines") or {}
        val = None
        for k in ("bench_trace.log", "bench_pmc3.log", "bench_pmc_fetch.log", "bench_pmc_write.log"):
            b = lines.get(k) if isinstance(lines, dict) else None
            if isinstance(b, dict) and b.get("value"):
                val = b["value"]
                break
        r = d.get("registers", {})
        rows.append((d["workload"].split(" (")[0], ms, pm["valu_instr_per_rod_substep"], frac, pm["valu_busy_frac"],
                     f'{r.get("Scratch_Size")} B',
                     (d.get("hbm_bytes_per_launch") or 0) / 1e6, val, d.get("library_source_hash")))
    print("| workload | kernel ms | VALU instr / rod-substep | `frac` | busy (raw) | scratch per lane | HBM MB / step | env-steps/s |")
    print("|---|---|---|---|---|---|---|---|")
    for w, ms, v, f, b, regs, hbm, val, h in rows:
        vs = "—" if val is None else (f"{val / 1e6:.2f} M" if val >= 1e6 else f"{val / 1e3:.0f} k")
        print(f"| {w} | {ms:.3f} | {v:.1f} | {f:.2f} | {b:.2f} | {regs} | {hbm:.0f} | {vs} |")
    print("hashes:", sorted({r[-1] for r in rows}))


if __name__ == "__main__":
    main(sys.argv[1] if len(sys.argv) > 1 else "r6")
