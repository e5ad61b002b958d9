#!/usr/bin/env python3
"""Rebuild the tables bench.py prices its kernels against — profiles/valu_counts.json and
profiles/hbm_traffic.json — from the per-workload summaries of ONE tools/profile_all.sh run:

    python tools/update_profile_tables.py profiles/r3a_*.json

Every entry carries `source_hash`, the softrod_source_hash() of the library that was profiled;
bench.py uses an entry only when the library it has loaded reports the same hash, and prints
`frac: null` with the reason otherwise.  Summaries whose passes ran on different builds (hash
None) are refused."""
import json
import re
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
valu = {"_about": "SQ_INSTS_VALU per rod-substep of each workload's step kernel(s): rocprofv3 --pmc SQ_INSTS_VALU "
                  "SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace -- python3 bench.py <workload> (tools/profile_all.sh; "
                  "three counters only - more perturb the kernel).  valu_busy_frac = SQ_ACTIVE_INST_VALU*4 / "
                  "(GRBM_GUI_ACTIVE/8 * 1024 SIMDs), raw: it can exceed 1 by ~1 % (mean over XCDs; nominal 4-cycle granule).  "
                  "source_hash = softrod_source_hash() of the profiled library; bench.py ignores entries of another build."}
hbm = {"_about": "HBM bytes per env.step of each workload: FETCH_SIZE and WRITE_SIZE (KiB) from two separate rocprofv3 "
                 "--pmc passes of python3 bench.py <workload>; read side x2 per MI355X_MICROARCH.md's gfx950 correction.  "
                 "source_hash as in valu_counts.json."}
for f in sys.argv[1:]:
    f = Path(f)
    d = json.loads(f.read_text())
    h = d.get("library_source_hash")
    if not h:
        sys.exit(f"{f}: passes ran on different builds of the library (or no bench line): refused")
    m = re.match(r"(\S+), (\d+) envs x (?:\d+ arms x )?(\d+) elements per GPU (tapered )?(libm kernel )?", d["workload"])
    env, envs, n_elem = m.group(1), int(m.group(2)), int(m.group(3))
    key = f"{env}|n_elem={n_elem}" + ("|taper" if m.group(4) else "") + ("|libm" if m.group(5) else "")      # bench.py profile_key()
    src = f"profiles/{f.name} ({d.get('command', 'tools/profile_all.sh')})"
    if "pmc3" in d:
        p = d["pmc3"]
        valu[key] = {
            "valu_instr_per_rod_substep": p["valu_instr_per_rod_substep"], "valu_busy_frac": p["valu_busy_frac"],
            "valu_issue_frac_measured_cycles": p["valu_issue_frac_measured_cycles"],
            "cycles_per_env_step_per_xcd": p["cycles_per_env_step_per_xcd"],
            "kernel_ms_trace_median_window": d.get("step_kernel_timed_avg_ms"),
            "registers": d.get("registers"), "source_hash": h, "source": src}
    if "hbm_bytes_per_launch" in d:
        per = int(d.get("step_kernels_per_env_step", 1))   # the windowed arm runs two kernels per env.step
        hbm[f"{key}|envs={envs}"] = {
            "FETCH_SIZE_KiB_raw": d["FETCH_SIZE_KiB_per_launch_raw"] * per,
            "WRITE_SIZE_KiB_raw": d["WRITE_SIZE_KiB_per_launch_raw"] * per,
            "hbm_bytes_per_launch": d["hbm_bytes_per_launch"] * per,
            "note": "read side x2 per MI355X_MICROARCH.md gfx950 FETCH_SIZE correction"
                    + ("; window kernel + epilogue kernel of one env.step added up" if per == 2 else ""),
            "source_hash": h, "source": src}
(ROOT / "profiles" / "valu_counts.json").write_text(json.dumps(valu, indent=1) + "\n")
(ROOT / "profiles" / "hbm_traffic.json").write_text(json.dumps(hbm, indent=1) + "\n")
print(json.dumps({"valu": sorted(k for k in valu if k != "_about"), "hbm": sorted(k for k in hbm if k != "_about")}))
