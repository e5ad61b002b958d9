#!/usr/bin/env python3
"""Which reformulation of the fast planar kernel buys the stabilised pendulum its head start on the oracle?
(VERDICT r5 "next" #4: the paired-divergence band of SoftPendulum-v0 is 16 x the FMA control's, OctoFlat's 4 x.)

Run ON THE GPU BOX.  For the shipped library, the LIBM kernel and every diagnostic build of `make -C
gym_softrobot_amd/csrc diag` (variants/libsoftrod_diag_<X>.so: ONE reformulation of the planar substep undone per
library — IEEE_DIV: correctly rounded divisions / square roots instead of the Newton-refined v_rcp / v_rsq seeds;
LIBM_TRIG: libm sin / cos; RECOMPUTE_EDGES: edges from the node positions every substep instead of integrated;
RECOMPUTE_ANGLE: the bending angle from the directors instead of carried; TWO_HALF_STEPS: two half kinematic steps
instead of one merged; NO_EPS_SIN: without the eps_sin correction (expected WORSE); NO_PLANAR: the general 3-D loop) the
ensemble scenario of tools/ensemble_parity.py `run_pendulum` (closed loop on each side's own observations, the
stabilising PD script, whole 126-step episode) runs in a process of its own (the library is chosen at load time:
SOFTROD_HIP_LIB) and reports, against the oracle (A) with the oracle's FMA build (B) as the control:

    worst / median ratio of HIP's paired-divergence quantiles to the control's, over all steps and statistics
    the same restricted to the steps before the ensemble saturates (step <= 60)
    NaN rods by step 126 (oracle, control, HIP)

    python tools/pendulum_divergence_attribution.py [--envs 384] > gpurun_out/pendulum_attribution.json
"""
from __future__ import annotations

import argparse
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))


def one(envs: int, math_mode):
    import numpy as np

    import ensemble_parity as ep
    from gym_softrobot_amd import _capi

    rec, series = ep.run_pendulum(envs, 126, True, math_mode=math_mode)
    head = ep.headline(rec)
    ratios_all, ratios_early = [], []
    for stat, rows in rec["stats"].items():
        for r in rows:
            if "hip" not in r or "control" not in r:
                continue
            for h, c in zip(r["hip"]["paired_q"], r["control"]["paired_q"]):
                if h > ep.FLOORS[stat] and c > 0:
                    ratios_all.append(h / c)
                    if r["step"] <= 60:
                        ratios_early.append(h / c)
    nan = {k: int(np.isnan(series[k]["x0"][-1]).sum()) for k in series}
    return {"library": os.environ.get("SOFTROD_HIP_LIB", "default"), "library_source_hash": _capi.library_source_hash(),
            "math_mode": "fast" if math_mode is None else "libm", "envs": envs,
            "worst_ratio_all_steps": max(ratios_all, default=0.0), "median_ratio_all_steps": float(np.median(ratios_all)) if ratios_all else 0.0,
            "worst_ratio_steps_le_60": max(ratios_early, default=0.0),
            "median_ratio_steps_le_60": float(np.median(ratios_early)) if ratios_early else 0.0,
            "nan_rods_by_step_126": nan, "violations_of_the_bands": len(ep.check(rec)),
            "worst_ratio_over_libm_control": head.get("worst_paired_ratio_over_libm_control"),
            "theta_q50_hip_over_control_last": (head["theta"]["hip"]["paired_q50_last"] / max(head["theta"]["control"]["paired_q50_last"], 1e-300))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--envs", type=int, default=384)
    ap.add_argument("--child", default=None, help=argparse.SUPPRESS)
    args = ap.parse_args()
    if args.child is not None:
        print("RESULT " + json.dumps(one(args.envs, 0 if args.child == "libm" else None)))
        return
    variants = [("shipped (fast)", None, "fast")]
    for d in ("IEEE_DIV", "LIBM_TRIG", "RECOMPUTE_EDGES", "RECOMPUTE_ANGLE", "TWO_HALF_STEPS", "NO_EPS_SIN", "NO_PLANAR"):
        f = ROOT / "variants" / f"libsoftrod_diag_{d}.so"
        if f.exists():
            variants.append((f"diag {d}", str(f), "fast"))
    variants.append(("shipped, LIBM kernel", None, "libm"))
    out = {"what": __doc__.split("\n\n")[0], "rows": []}
    for name, lib, mode in variants:
        env = dict(os.environ)
        if lib:
            env["SOFTROD_HIP_LIB"] = lib
        p = subprocess.run([sys.executable, __file__, "--envs", str(args.envs), "--child", mode], env=env,
                           capture_output=True, text=True, timeout=1500)
        row = {"variant": name}
        for ln in p.stdout.splitlines():
            if ln.startswith("RESULT "):
                row.update(json.loads(ln[7:]))
        if p.returncode != 0:
            row["error"] = p.stderr[-800:]
        out["rows"].append(row)
        sys.stderr.write(json.dumps(row) + "\n")
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
