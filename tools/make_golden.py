"""Generate tests/golden/*.  Two kinds of fixture, kept apart on purpose:

1. softpendulum_reset.json — REFERENCE-DERIVED golden vectors.  The only outputs of
   SoftPendulum-v0 that depend on gym-softrobot's own code + NumPy alone (PyElastica
   is not importable here): the observation returned by `reset(seed=s)`.  The
   expressions below are the reference's, evaluated with NumPy:
     theta0, direction                gym_softrobot/envs/soft_pendulum/build.py:47-51
     rng = Generator(PCG64(SeedSequence(seed)))   gymnasium.utils.seeding.np_random,
                                      reached via soft_pendulum.py:114
     obs = [x0, vx0, prev_action, wrap(arctan(mean tx / mean ty))]
                                      soft_pendulum.py:149-161
   with tangents of a straight rod = direction (CosseratRod.straight_rod).
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
    out = []
    for seed in (0, 1, 2, 3, 42, 123, 2024):
        rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence(seed)))
        u = rng.random()
        theta = np.deg2rad(90 + (u - 0.5) * 10)
        direction = np.array([1.0 * np.cos(theta), 1.0 * np.sin(theta), 0.0])
        tangents = np.repeat(direction[:, None], 50, axis=1)
        tm = np.mean(tangents, axis=1)
        th = np.arctan(tm[0] / tm[1])
        th = ((th + np.pi) % (2 * np.pi)) - np.pi
        obs = np.hstack([0.0, 0.0, np.zeros(1, np.float32), th]).astype(np.float32)
        out.append({"seed": seed, "u": float(u), "theta0": float(theta), "obs": [float(v) for v in obs]})
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


if __name__ == "__main__":
    g = ROOT / "tests" / "golden"
    g.mkdir(parents=True, exist_ok=True)
    (g / "softpendulum_reset.json").write_text(json.dumps(reset_vectors(), indent=1))
    np.savez(g / "softpendulum_oracle_rollout.npz", **oracle_rollout())
    print("wrote", sorted(p.name for p in g.iterdir()))
