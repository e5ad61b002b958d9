#!/usr/bin/env python3
"""Fold a tracked 3-counter PMC summary (tools/pmc_valu_per_substep.sh -> summary.json, copied to
profiles/<round>_pmc3_<tag>.json) into profiles/valu_counts.json under the workload key bench.py
looks up ("<env>|n_elem=<n>").

    python tools/update_valu_counts.py profiles/r2a_pmc3_SoftPendulum-v0.json 50
"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
src, n_elem = Path(sys.argv[1]), int(sys.argv[2])
d = json.loads(src.read_text())
out = ROOT / "profiles" / "valu_counts.json"
doc = json.loads(out.read_text()) if out.exists() else {}
key = f"{d['env']}|n_elem={n_elem}"
doc[key] = {
    "valu_instr_per_rod_substep": d["valu_instr_per_rod_substep"],
    "valu_busy_frac": d["valu_busy_frac"],
    "valu_issue_frac_measured_cycles": d["valu_issue_frac_measured_cycles"],
    "source": f"profiles/{src.name} ({d['command']}; {d['envs']} envs x {d['substeps_per_step']} substeps x {d['rods_per_env']} rods)",
}
out.write_text(json.dumps(doc, indent=1) + "\n")
print(key, json.dumps(doc[key]))
