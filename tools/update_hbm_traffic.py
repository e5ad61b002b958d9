#!/usr/bin/env python3
"""Fold the FETCH_SIZE/WRITE_SIZE result of a tools/profile_gpu.sh summary into
profiles/hbm_traffic.json under the workload key bench.py looks up.

    python tools/update_hbm_traffic.py profiles/r1e_summary.json "SoftPendulum-v0|n_elem=50|envs=4096"
"""
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
summary, key = Path(sys.argv[1]), sys.argv[2]
d = json.loads(summary.read_text())
out = ROOT / "profiles" / "hbm_traffic.json"
doc = json.loads(out.read_text()) if out.exists() else {}
if "hbm_bytes_per_launch" in doc:      # old single-entry layout
    doc = {}
doc[key] = {
    "source": f"{summary.name} (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, tools/profile_gpu.sh)",
    "FETCH_SIZE_KiB_raw": d.get("FETCH_SIZE_KiB_per_launch_raw"),
    "WRITE_SIZE_KiB_raw": d.get("WRITE_SIZE_KiB_per_launch_raw"),
    "hbm_bytes_per_launch": d.get("hbm_bytes_per_launch"),
    "note": "read side x2 per MI355X_MICROARCH.md gfx950 FETCH_SIZE correction",
}
out.write_text(json.dumps(doc, indent=1) + "\n")
print(json.dumps(doc[key]))
