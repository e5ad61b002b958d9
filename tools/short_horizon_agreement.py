#!/usr/bin/env python3
"""Short-horizon agreement of the kernels with each other and with the oracle, at ROUNDING level
(run on the MI355X box: python tools/short_horizon_agreement.py > profiles/r3_short_horizon_agreement.json).

The whole-episode tolerances (1e-5) cannot see a systematic term of 1e-10 in a motion that is not
chaotic, and round 3 found exactly such a term by looking closer (sin(theta + eps_sin), DESIGN.md
§3).  This tool looks closer everywhere else: OctoArmSingle-v0 (plane contact, anisotropic
friction, rest-kappa actuation) fast vs libm vs oracle after 3 env.steps of 1 / 10 / 100 / 714
substeps, and OctoFlat-v0 (joints, head, contact) fast vs oracle after 2 env.steps of 1 / 10 / 100 /
400 substeps; every number is max|a - b| / max|b| of a whole field.  Anything much above 1e-12 here
would be a formula difference, not a rounding."""
import json
import sys
import numpy as np, torch
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parents[1]))
import gym_softrobot_amd as gsa
from gym_softrobot_amd import _capi
from gym_softrobot_amd.backend import HipRodBackend
from oracle import oracle_c
oracle_c.build()

OUT = {"metric": "max|a-b| / max|b| over a whole field (head_xy_abs: absolute, the head starts at the origin)",
       "library_source_hash": _capi.library_source_hash(), "OctoArmSingle-v0": {}, "OctoFlat-v0": {}}


def relscale(a, b):
    return float(np.max(np.abs(a - b)) / np.max(np.abs(b)))

# --- OctoArmSingle: fast vs libm vs oracle over short horizons
for n_sub in (1, 10, 100, 714):
    n = 8
    cfgs = {}
    res = {}
    for mode in (0, 1):
        cfg = _capi.arm_single_config(n, math_mode=mode)
        cfg.n_substeps = n_sub
        be = HipRodBackend(cfg, 0)
        be.reset_straight(np.zeros((n, 3)), np.tile([1.0, 0, 0], (n, 1)), np.tile([0, 0, 1.0], (n, 1)))
        be.observe(None)
        acts = np.random.default_rng(7).uniform(-6, 6, (3, n, 7)).astype(np.float32)
        for t in range(3):
            be.step(acts[t])
        torch.cuda.synchronize()
        res[mode] = be.state_numpy()
        be.close()
    rods = []
    cfg = _capi.arm_single_config(n); cfg.n_substeps = n_sub
    for i in range(n):
        r = oracle_c.OracleRod(cfg); r.reset_arm()
        for t in range(3):
            r.env_step_arm(acts[t, i])
        rods.append(r)
    ox = np.stack([r.get("x") for r in rods]); oq = np.stack([r.get("Q") for r in rods]); ow = np.stack([r.get("w") for r in rods])
    OUT["OctoArmSingle-v0"][f"{n_sub} substeps x 3 steps"] = {
        "fast_vs_libm": {"x": relscale(res[1]["x"], res[0]["x"]), "Q": relscale(res[1]["Q"], res[0]["Q"])},
        "fast_vs_oracle": {"x": relscale(res[1]["x"], ox), "Q": relscale(res[1]["Q"], oq), "w": relscale(res[1]["w"], ow)},
        "libm_vs_oracle": {"x": relscale(res[0]["x"], ox), "Q": relscale(res[0]["Q"], oq), "w": relscale(res[0]["w"], ow)}}

# --- OctoFlat: fast vs oracle
for n_sub in (1, 10, 100, 400):
    n = 3
    cfg = _capi.octo_flat_config(n); cfg.n_substeps = n_sub
    be = HipRodBackend(cfg, 0)
    tg = np.random.default_rng(11).uniform(0.5, 2.0, (n, 2))
    be.reset_octo(tg)
    orc = [oracle_c.OracleOcto(cfg) for _ in range(n)]
    for i in range(n):
        orc[i].reset(tg[i])
    acts = np.random.default_rng(3).uniform(-22, 22, (2, n, 24)).astype(np.float32)
    for t in range(2):
        be.step(acts[t])
        for i in range(n):
            orc[i].env_step(acts[t, i])
    torch.cuda.synchronize()
    st = be.octo_state_numpy()
    ox = np.stack([np.stack([o.arm(a).get("x") for a in range(8)]) for o in orc])
    oq = np.stack([np.stack([o.arm(a).get("Q") for a in range(8)]) for o in orc])
    hx = np.stack([o.head()["x"] for o in orc]); hq = np.stack([o.head()["Q"] for o in orc])
    OUT["OctoFlat-v0"][f"{n_sub} substeps x 2 steps"] = {
        "fast_vs_oracle": {"arms_x": relscale(st["x"], ox), "arms_Q": relscale(st["Q"], oq),
                           "head_xy_abs": float(np.max(np.abs(st["head_x"] - hx))), "head_Q": relscale(st["head_Q"], hq)}}
    be.close()
print(json.dumps(OUT, indent=1))
