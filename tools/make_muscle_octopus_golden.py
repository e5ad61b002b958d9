#!/usr/bin/env python3
"""Golden vectors produced by EXECUTING the reference's own muscle-octopus env code under
tools/refshim.py's stand-ins and the recording COOMM stand-ins of tools/make_muscle_env_golden.py:

    CrawlEnv    gym_softrobot/envs/octopus/crawl_env.py     (OctoCrawl-v0)
    ArmTwoEnv   gym_softrobot/envs/octopus/arm_two_env.py   (OctoArmTwo-v0)
    ReachEnv    gym_softrobot/envs/octopus/reach_env.py     (OctoReach-v0)
    build_octopus_muscles / build_two_arms / build_arm      gym_softrobot/envs/octopus/build_muscle_octopus.py
    create_es_muscle_layers                                  gym_softrobot/envs/octopus/build.py:295-338

NOTHING of the muscle force law or of PyElastica's stepper is exercised (neither is on disk).  What the fixtures pin is
what the reference's files themselves do: the builds (arm frames, radii, material, dampers and their own time_step,
the head Cylinder, the joints' arguments incl. their angles, registration order, suckers, the layers' constructor
arguments), `reset` (targets, first observation), `set_action` (sucker indices and ratios, the activations every layer
received, incl. ArmTwoEnv's cubic interpolation), `get_state` (layout, dtype, np.nan_to_num, ArmTwoEnv's _prev_kappa),
and `step`'s reward / termination / truncation code around a SCRIPTED stepper that installs recorded states.

The states come from this repo's oracle (tests/oracle_mocto.py rollouts) or are synthetic.

Outputs: tests/golden/ref_muscle_octopus.npz, tests/golden/ref_muscle_octopus_build_records.json — data only.
tests/test_muscle_octopus.py replays them through the oracle's env code (CPU), tests/test_gpu_muscle_octopus.py through
the HIP library (state-view injection into a handle of n_substeps = 0).

    python tools/make_muscle_octopus_golden.py
"""
from __future__ import annotations

import json
import sys
import warnings
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))
sys.path.insert(0, str(ROOT / "tests"))

import refshim  # noqa: E402
from make_muscle_env_golden import Stack, install_coomm, jsonable  # noqa: E402

from gym_softrobot_amd import _capi  # noqa: E402
from oracle import oracle_c  # noqa: E402
from oracle_mocto import MuscleOctopusOracleEnv  # noqa: E402

GOLD = ROOT / "tests" / "golden"
warnings.filterwarnings("ignore", category=RuntimeWarning)
ENVS = (("OctoCrawl", "crawl_env", "CrawlEnv", _capi.ENV_CRAWL), ("OctoArmTwo", "arm_two_env", "ArmTwoEnv", _capi.ENV_ARM_TWO),
        ("OctoReach", "reach_env", "ReachEnv", _capi.ENV_REACH))


def install_allocations():
    """straight_rod / Cylinder hand back fakes holding what PyElastica's allocation would (node positions, frames,
    masses, radii, rest lengths: the oracle's allocation for the recorded arguments), so that reset's get_state and
    ReachEnv's `sum(rest_lengths)` run on real numbers."""
    orig = refshim.CosseratRod.straight_rod

    def straight_rod(*a, **k):
        rod = orig(*a, **k)
        r = rod.recorded
        cfg = _capi.muscle_octopus_config(_capi.ENV_CRAWL, 1, n_elems=int(r["n_elements"]))
        cfg.features &= ~_capi.FEAT_OCTO_HEAD
        cfg.env_kind = _capi.ENV_NONE
        o = oracle_c.OracleRod(cfg)
        o.set_radius_profile(np.asarray(r["base_radius"], np.float64))
        o.reset_straight(r["start"], r["direction"], r["normal"])
        rod.position_collection[:] = o.get("x")
        rod.director_collection[:] = o.get("Q")
        rod.mass[:] = o.get("mass")
        rod.radius[:] = np.asarray(r["base_radius"], np.float64)      # allocate() stores the radii as given
        rod.rest_lengths[:] = o.get("rest_lengths")
        rod.lengths[:] = o.get("rest_lengths")
        return rod
    refshim.CosseratRod.straight_rod = staticmethod(straight_rod)
    cyl_init = refshim.Cylinder.__init__

    def cylinder(self, start, direction, normal, base_length, base_radius, density):
        cyl_init(self, start, direction, normal, base_length, base_radius, density)
        d, nrm = np.asarray(direction, np.float64), np.asarray(normal, np.float64)
        self.position_collection[:, 0] = np.asarray(start, np.float64) + d * base_length / 2
        self.director_collection[0, :, 0] = nrm
        self.director_collection[1, :, 0] = np.cross(d, nrm)
        self.director_collection[2, :, 0] = d
    refshim.Cylinder.__init__ = cylinder


def body_state(orc: MuscleOctopusOracleEnv):
    na = orc.n_arm
    h = orc.head()
    return {"x": np.stack([orc.arm(a).get("x") for a in range(na)]), "v": np.stack([orc.arm(a).get("v") for a in range(na)]),
            "kappa": np.stack([orc.arm(a).get("kappa") for a in range(na)]),
            "hx": h["x"].copy(), "hv": h["v"].copy(), "hQ": h["Q"].copy(), "time": np.float64(orc.time)}


def install(env, st):
    for a, rod in enumerate(env.shearable_rods):
        rod.position_collection[:] = st["x"][a]
        rod.velocity_collection[:] = st["v"][a]
        rod.kappa[:] = st["kappa"][a]
    env.rigid_rod.position_collection[:, 0] = st["hx"]
    env.rigid_rod.velocity_collection[:, 0] = st["hv"]
    env.rigid_rod.director_collection[:, :, 0] = st["hQ"]


def main():
    refshim.install()
    install_coomm()
    oracle_c.build()
    install_allocations()
    records = {"_about": "constructor arguments and operator registration order recorded while executing the reference's "
                         "CrawlEnv / ArmTwoEnv / ReachEnv reset() (build_octopus_muscles, build_two_arms, create_es_muscle_layers) "
                         "under recording stand-ins (tools/make_muscle_octopus_golden.py); no muscle force law, no stepper"}
    out = {}
    for name, modname, clsname, kind in ENVS:
        mod = refshim.load(f"gym_softrobot.envs.octopus.{modname}")
        env = getattr(mod, clsname)()
        obs0, info0 = env.reset(seed=0)
        assert info0 == {}
        sim = env.simulator
        ops = []
        for op in sim._ops:
            kw = {k: jsonable(v) for k, v in op["kwargs"].items() if k not in ("controller", "muscles", "callback_params_list")}
            ops.append({"kind": op["kind"], "cls": op["cls"].__name__, "kwargs": kw,
                        "targets": [sim._systems.index(t) for t in op["targets"] if t in sim._systems],
                        "indices": list(getattr(op.get("using", None), "indices", ()))})
        muscles = env.tm_muscle_activations if kind == _capi.ENV_CRAWL else env.muscle_activations
        suckers = []
        if kind == _capi.ENV_CRAWL:
            suckers = [[{"index": c.index, "flag": bool(c.flag), "reduction_ratio": c.reduction_ratio}] for c in env.sucker_controller]
        elif kind == _capi.ENV_ARM_TWO:
            suckers = [[{"index": c.index, "flag": bool(c.flag), "reduction_ratio": c.reduction_ratio} for c in arm] for arm in env.sucker_controller]
        tp = prefix = name[4:].lower() + "_"
        records[name] = {
            "init": {"step_skip": env.step_skip, "final_time": env.final_time, "time_step": env.time_step, "n_arm": env.n_arm,
                     "n_elems": env.n_elems, "n_action": env.n_action, "obs_shape": list(env.observation_space.shape),
                     "action_shape": list(env.action_space.shape), "action_low": float(env.action_space.low.min()),
                     "action_high": float(env.action_space.high.max()), "reward_range": env.reward_range,
                     "env_info": env.get_env_info(), "metadata": env.metadata,
                     **({"sucker_location": env.sucker_location, "control_location": env.control_location} if kind == _capi.ENV_ARM_TWO else {})},
            "arms": [{k: jsonable(v) for k, v in rod.recorded.items()} for rod in env.shearable_rods],
            "cylinder": {k: jsonable(v) for k, v in env.rigid_rod.recorded.items()},
            "order": sim.order(), "ops": ops,
            "connect_indices": list(getattr(sim, "_last_connect_idx", ())),
            "muscle_layers_arm0": [{"kind": m.kind, **{k: jsonable(v) for k, v in m.kwargs.items()}} for m in muscles[0]],
            "suckers": suckers, "target": jsonable(np.asarray(env._target)), "target_dtype": str(np.asarray(env._target).dtype),
        }
        out[prefix + "reset_obs"] = np.asarray(obs0)
        out[prefix + "reset_target"] = np.asarray(env._target, np.float64)

        # ---- step(): the reference's set_action / get_state / reward code around a scripted stepper ------------------
        cfg = _capi.muscle_octopus_config(kind, 1)
        orc = MuscleOctopusOracleEnv(cfg)
        orc.reset(np.asarray(env._target, np.float64) if kind == _capi.ENV_REACH else None)
        S = Stack()
        rng = np.random.default_rng(20 + kind)
        na, nk, n = env.n_arm, env.n_action, env.n_elems

        def ref_step(action, pre, post, label):
            install(env, pre)
            for arm_m in muscles:
                for m in arm_m:
                    m.activations.clear()
            prev_action = np.asarray(env._prev_action, np.float32).ravel().copy()
            prev_kappa = np.asarray(getattr(env, "_prev_kappa", np.zeros((na, n - 1))), np.float32).copy()
            sim._calls = 0

            def script(k, t, dt):
                if k == env.step_skip:
                    install(env, post)
                    return np.float64(post["time"])
                return t
            sim._script = script
            env.time = np.float64(pre["time"])
            obs, rew, term, trunc, info = env.step(action)
            assert sim._calls == env.step_skip
            acts = np.full((na, 3, n), np.nan)        # what each layer's LAST apply_activation of this step received (NaN: none)
            for a, arm_m in enumerate(muscles):
                for j, m in enumerate(arm_m):
                    if m.activations:
                        acts[a, j] = np.broadcast_to(np.asarray(m.activations[-1], np.float64), (n,))
            sidx = np.full((na, 3), -99, np.int64)
            srat = np.full((na, 3), np.nan)
            if kind == _capi.ENV_CRAWL:
                for a, c in enumerate(env.sucker_controller):
                    sidx[a, 0], srat[a, 0] = c.index, c.reduction_ratio
            elif kind == _capi.ENV_ARM_TWO:
                for a, arm_c in enumerate(env.sucker_controller):
                    for j, c in enumerate(arm_c):
                        sidx[a, j], srat[a, j] = c.index, c.reduction_ratio
            S.add(label=label, action=np.asarray(action, np.float32).ravel(), prev_action_before=prev_action, prev_kappa_before=prev_kappa,
                  pre_hx=pre["hx"], pre_time=pre["time"],
                  x=post["x"], v=post["v"], kappa=post["kappa"], hx=post["hx"], hv=post["hv"], hQ=post["hQ"], time=post["time"],
                  obs=obs, reward=np.float64(rew), terminated=bool(term), truncated=bool(trunc), info_time=np.float64(info["time"]),
                  sucker_index=sidx, sucker_ratio=srat, activations=acts,
                  prev_kappa_after=np.asarray(getattr(env, "_prev_kappa", np.zeros((na, n - 1))), np.float32).copy())

        # a rollout of the oracle: pre = before the step, post = after it
        last = None
        for k in range(4):
            a = rng.uniform(0.0, 1.0, na * nk).astype(np.float32) * (0.6 if kind == _capi.ENV_REACH else 1.0)
            if k == 3 and kind == _capi.ENV_CRAWL:      # locations at and beyond both ends of the arm, and just below a whole index
                a[0::3] = np.array([-0.3, 0.0, 0.05, 0.35, 0.95, 0.999, 1.0, 1.8], np.float32)
            pre = body_state(orc)
            orc.step(a)
            ref_step(a, pre, body_state(orc), f"rollout{k}")
            last = a
        base = body_state(orc)
        for label, key, idx in (("nan_x", "x", (1, 1, 7)), ("nan_x_z", "x", (0, 2, 20)), ("nan_v", "v", (na - 1, 2, 0)),
                                ("nan_kappa", "kappa", (0, 0, 3)), ("inf_v", "v", (1, 0, 4)), ("nan_head_v", "hv", (1,))):
            st = {k: np.array(v, copy=True) for k, v in base.items()}
            st[key][idx] = np.inf if label == "inf_v" else np.nan
            ref_step(last, base, st, label)
        ft = float(env.final_time)
        for label, t in (("time_eq_final", ft), ("time_just_past", np.nextafter(ft, 2 * ft))):
            st = {k: np.array(v, copy=True) for k, v in base.items()}
            st["time"] = np.float64(t)
            ref_step(last, base, st, label)
        st = {k: np.array(v, copy=True) for k, v in base.items()}
        if kind == _capi.ENV_REACH:                  # a tip inside 0.1 * 0.25 of the target; and past final_time at once
            st["x"][3, :, -1] = np.asarray(env._target, np.float64) + np.array([0.004, -0.003, 0.002])
            ref_step(last, base, st, "tip_at_target")
            st["time"] = np.float64(ft + 1.0)
            ref_step(last, base, st, "tip_at_target_late")
        else:                                        # the head inside 0.2 of (5, 0); and past final_time at once
            st["hx"][:2] = [4.9, 0.12]
            ref_step(last, base, st, "head_at_target")
            st["time"] = np.float64(ft + 1.0)
            ref_step(last, base, st, "head_at_target_late")
            far = {k: np.array(v, copy=True) for k, v in base.items()}
            far["hx"][:2] = [0.4, -0.3]
            far["time"] = np.float64(ft + 1.0)
            ref_step(last, base, far, "moved_and_late")
        out.update(S.arrays(prefix + "step_"))
        if kind == _capi.ENV_CRAWL:      # config_random_final_time (crawl_env.py:135-136): the episode's own final_time, first draw of reset()
            env2 = getattr(mod, clsname)(config_random_final_time=True)
            fts = []
            for seed in (0, 1, 42):
                env2.reset(seed=seed)
                fts.append(float(env2.final_time))
            env2.reset()                 # no seed: the stream goes on
            fts.append(float(env2.final_time))
            out[prefix + "random_final_time"] = np.array(fts)

    GOLD.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(GOLD / "ref_muscle_octopus.npz", **out)
    (GOLD / "ref_muscle_octopus_build_records.json").write_text(json.dumps(jsonable(records), indent=1) + "\n")
    for f in ("ref_muscle_octopus.npz", "ref_muscle_octopus_build_records.json"):
        print(f, (GOLD / f).stat().st_size)


if __name__ == "__main__":
    main()
