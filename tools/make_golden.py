"""Generate tests/golden/*.  Two kinds of fixture, kept apart on purpose:

1. softpendulum_reset.json, other_envs_reset.json — REFERENCE-DERIVED golden vectors: the
   reset observations, re-exported from the fixtures that tools/make_env_golden.py records by
   EXECUTING the reference's own reset() / build_* / get_state (tests/golden/ref_*.npz; run that
   tool first).  No expression of the reference is re-typed here.
2. softpendulum_oracle_rollout.npz — REGRESSION pins produced by this repo's own fp64
   C oracle (oracle/softrod_oracle.c).  They are NOT reference outputs ("parity
   unpinned", see the oracle header); they freeze the oracle so that an accidental edit
   is caught, and they travel to the GPU box where the oracle is rebuilt from source.
"""
import json
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def reset_vectors():
    """From tests/golden/ref_softpendulum.npz: the reset observations and rod frames the
    reference's own SoftPendulumEnv.reset / build_soft_pendulum produced (tools/make_env_golden.py
    executes them); u is the RNG draw behind the recorded direction, theta0 its angle."""
    z = np.load(ROOT / "tests" / "golden" / "ref_softpendulum.npz")
    out = []
    for i, seed in enumerate(z["reset_seed"]):
        rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence(int(seed))))
        u = rng.random()
        d = z["reset_direction"][i]
        theta = np.deg2rad(90 + (u - 0.5) * 10)
        assert d[0] == 1.0 * np.cos(theta) and d[1] == 1.0 * np.sin(theta)     # build.py:47-50 as executed
        out.append({"seed": int(seed), "u": float(u), "theta0": float(theta),
                    "obs": [float(v) for v in z["reset_obs"][i]]})
    return out


def oracle_rollout():
    from gym_softrobot_amd._capi import softpendulum_config
    from oracle.oracle_c import OracleRod

    cfg = softpendulum_config(1)
    seeds = [0, 1, 42, 123]
    T = 5
    acts = np.random.default_rng(7).uniform(-22, 22, (T, len(seeds))).astype(np.float32)
    obs = np.zeros((T, len(seeds), 4), np.float32)
    rew = np.zeros((T, len(seeds)))
    xfin = []
    for j, s in enumerate(seeds):
        rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence(s)))
        r = OracleRod(cfg)
        r.reset_pendulum(np.deg2rad(90 + (rng.random() - 0.5) * 10))
        for t in range(T):
            o, rw, term, trunc = r.env_step(acts[t, j])
            obs[t, j], rew[t, j] = o, rw
        xfin.append(r.get("x"))
    return dict(seeds=np.array(seeds), actions=acts, obs=obs, reward=rew, x_final=np.stack(xfin))


def reset_vectors_other_envs():
    """Reset observations of the other envs, from the fixtures recorded by executing the
    reference's own reset() (tests/golden/ref_softpendulum3d.npz, ref_armsingle.npz,
    ref_octoflat.npz; tools/make_env_golden.py)."""
    g = ROOT / "tests" / "golden"
    out = {}
    z = np.load(g / "ref_softpendulum3d.npz")
    v3 = []
    for i, s in enumerate(z["reset_seed"]):
        rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence(int(s))))
        tilt = np.deg2rad(rng.uniform(-1.0, 1.0))          # the draw behind the recorded direction
        d = z["reset_direction"][i]
        assert d[0] == np.sin(tilt) and d[2] == np.cos(tilt)   # soft_pendulum_3d/build.py:51-52 as executed
        v3.append({"seed": int(s), "tilt": float(tilt), "obs": [float(x) for x in z["reset_obs"][i]]})
    out["SoftPendulum3D-v0"] = v3
    z = np.load(g / "ref_armsingle.npz")
    out["OctoArmSingle-v0"] = [{"seed": 0, "obs": [float(x) for x in z["reset_obs"]]}]
    z = np.load(g / "ref_octoflat.npz")
    out["OctoFlat-v0"] = [
        {"seed": int(s), "target": [float(x) for x in z["reset_target"][i]],
         "individual": z["reset_individual"][i].tolist(), "shared": z["reset_shared"][i].tolist()}
        for i, s in enumerate(z["reset_seed"])]
    return out


def oracle_rollouts_other_envs():
    """REGRESSION pins of this repo's oracle for the other envs (not reference outputs)."""
    from gym_softrobot_amd import _capi
    from oracle.oracle_c import OracleOcto, OracleRod

    out = {}
    rng = np.random.default_rng(11)
    cfg = _capi.softpendulum3d_config(1)
    r = OracleRod(cfg)
    r.reset_pendulum3d(np.deg2rad(0.37))
    a = rng.uniform(-1, 1, (3, 2)).astype(np.float32)
    res = [r.env_step3d(a[t]) for t in range(3)]
    out.update(p3d_actions=a, p3d_obs=np.stack([x[0] for x in res]), p3d_reward=np.array([x[1] for x in res]),
               p3d_x=r.get("x"))
    cfg = _capi.arm_single_config(1)
    r = OracleRod(cfg)
    r.reset_arm()
    a = rng.uniform(-6, 6, (3, 7)).astype(np.float32)
    res = [r.env_step_arm(a[t]) for t in range(3)]
    out.update(arm_actions=a, arm_obs=np.stack([x[0] for x in res]), arm_reward=np.array([x[1] for x in res]),
               arm_x=r.get("x"))
    cfg = _capi.octo_flat_config(1)
    cfg.n_substeps = 200          # short windows: whole OctoFlat rollouts are chaotic
    o = OracleOcto(cfg)
    o.reset([1.0, 1.3])
    a = rng.uniform(-22, 22, (2, 24)).astype(np.float32)
    res = [o.env_step(a[t]) for t in range(2)]
    out.update(octo_actions=a, octo_target=np.array([1.0, 1.3]),
               octo_individual=np.stack([x[0]["individual"] for x in res]),
               octo_shared=np.stack([x[0]["shared"] for x in res]), octo_reward=np.array([x[1] for x in res]))
    return out


def oracle_rollout_soft_arm():
    """REGRESSION pin of this repo's oracle for SoftArmTracking-v0 (not reference outputs; the
    actuation itself is pinned against reference code in softarm_vectors.npz)."""
    from gym_softrobot_amd import _capi
    from oracle.oracle_c import OracleRod

    rng = np.random.default_rng(21)
    r = OracleRod(_capi.soft_arm_config(1))
    r.reset_soft_arm()
    a = rng.uniform(-1, 1, (6, 8)).astype(np.float32)
    a[3] = a[2]
    res = [r.env_step_soft_arm(a[t]) for t in range(6)]
    return dict(actions=a, obs=np.stack([x[0] for x in res]), reward=np.array([x[1] for x in res]),
                x=r.get("x"), kappa=r.get("kappa"))


if __name__ == "__main__":
    g = ROOT / "tests" / "golden"
    g.mkdir(parents=True, exist_ok=True)
    (g / "softpendulum_reset.json").write_text(json.dumps(reset_vectors(), indent=1))
    np.savez(g / "softpendulum_oracle_rollout.npz", **oracle_rollout())
    (g / "other_envs_reset.json").write_text(json.dumps(reset_vectors_other_envs(), indent=1))
    np.savez(g / "other_envs_oracle_rollout.npz", **oracle_rollouts_other_envs())
    np.savez(g / "softarm_oracle_rollout.npz", **oracle_rollout_soft_arm())
    print("wrote", sorted(p.name for p in g.iterdir()))
