#!/usr/bin/env python3
"""Golden vectors produced by EXECUTING the reference's own muscle-arm env code, the way
tools/make_env_golden.py does for the other envs (tools/refshim.py explains how the reference's files can run
here without gymnasium / pyelastica / numba / coomm):

    ArmPushEnv   gym_softrobot/envs/octopus/arm_push_env.py:52-347   (OctoArmPush-v0 "discrete", -v1 "continuous")
    create_es_muscle_layers   gym_softrobot/envs/octopus/build.py:295-338
    ControllableFixConstraint / SuckerController   octopus/controllable_constraint.py:9-69

COOMM is not on disk, so its classes are RECORDING stand-ins here: `LongitudinalMuscle(...)`, `TransverseMuscle(...)`
remember their constructor arguments and what `apply_activation` was given; `ApplyMuscles(...)` remembers its
muscles.  NOTHING of the muscle force law is exercised or pinned by these fixtures — only what the reference's own
files do: the arm's geometry and material, the operator registration order, the layers' constructor arguments,
`set_action` (the sucker's index, the activations), `prev_cm_pos`, the NaN check over position / velocity /
director / alpha / omega / centre of mass, reward, truncation, `get_state` with `np.nan_to_num`.

The rod states the scripted stepper installs come from this repo's oracle (a short rollout) or are synthetic; the
stepper itself (PyElastica + COOMM) is not on disk.

Outputs: tests/golden/ref_armpush.npz, tests/golden/ref_muscle_build_records.json — data only.
tests/test_muscle_reference_fixtures.py replays them through the oracle (CPU) and
tests/test_gpu_muscle_fixtures.py through the HIP library (state-view injection, n_substeps = 0).

    python tools/make_muscle_env_golden.py
"""
from __future__ import annotations

import json
import sys
import types
import warnings
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))

import refshim  # noqa: E402

from gym_softrobot_amd import _capi  # noqa: E402
from oracle import oracle_c  # noqa: E402

GOLD = ROOT / "tests" / "golden"
warnings.filterwarnings("ignore", category=RuntimeWarning)


# ---- COOMM: recording stand-ins (no arithmetic) ---------------------------------------------------------------
class _Muscle:
    def __init__(self, **kwargs):
        self.kind = type(self).__name__
        self.kwargs = {k: (np.array(v, dtype=np.float64) if isinstance(v, np.ndarray) else v) for k, v in kwargs.items()}
        self.activations = []

    def apply_activation(self, activation):
        self.activations.append(np.array(activation, dtype=np.float64))


class LongitudinalMuscle(_Muscle):
    pass


class TransverseMuscle(_Muscle):
    pass


class ApplyMuscles(refshim.NoForces):
    def __init__(self, muscles, step_skip, callback_params_list):
        super().__init__()
        self.muscles, self.step_skip = muscles, step_skip


def install_coomm():
    mods = {"coomm": {}, "coomm.actuations": {}, "coomm.actuations.muscles": {},
            "coomm.actuations.muscles.longitudinal_muscle": {"LongitudinalMuscle": LongitudinalMuscle},
            "coomm.actuations.muscles.transverse_muscle": {"TransverseMuscle": TransverseMuscle},
            "coomm.actuations.muscles.muscle": {"ApplyMuscles": ApplyMuscles}}
    for name, attrs in mods.items():
        m = types.ModuleType(name)
        for k, v in attrs.items():
            setattr(m, k, v)
        sys.modules[name] = m


def jsonable(v):
    if isinstance(v, np.ndarray):
        return v.tolist()
    if isinstance(v, (np.floating, np.integer)):
        return v.item()
    if isinstance(v, (list, tuple)):
        return [jsonable(x) for x in v]
    if isinstance(v, dict):
        return {k: jsonable(x) for k, x in v.items()}
    return v


class Stack:
    def __init__(self):
        self.rows = []

    def add(self, **kw):
        self.rows.append({k: np.array(v) for k, v in kw.items()})

    def arrays(self, prefix):
        return {prefix + k: np.stack([r[k] for r in self.rows]) for k in self.rows[0].keys()}


def oracle_arm(mode):
    """The oracle's ArmPush rod (only to have non-trivial, physically sensible states to install)."""
    cfg = _capi.arm_push_config(1, mode=mode)
    o = oracle_c.OracleRod(cfg)
    rad = _capi.arm_push_radii()
    o.set_radius_profile(rad)
    o.set_muscle_layers(*_capi.es_muscle_layers(rad, 0.012))
    o.reset_push()
    return o


def fill_rod(rod, orc):
    rod.position_collection[:] = orc.get("x")
    rod.velocity_collection[:] = orc.get("v")
    rod.director_collection[:] = orc.get("Q")
    rod.omega_collection[:] = orc.get("w")
    rod.mass[:] = orc.get("mass")
    rod.radius[:] = orc.get("radius")


def state_of(orc):
    return {"x": orc.get("x").copy(), "v": orc.get("v").copy(), "Q": orc.get("Q").copy(), "w": orc.get("w").copy(),
            "alpha": np.zeros((3, orc.n))}


def main():
    refshim.install()
    install_coomm()
    oracle_c.build()
    mod = refshim.load("gym_softrobot.envs.octopus.arm_push_env")
    # straight_rod hands back a fake rod holding the oracle's ALLOCATION for the recorded arguments (node positions,
    # frames, masses), so that the reset observation is the reference's get_state on a real straight arm
    orig = refshim.CosseratRod.straight_rod

    def straight_rod(*a, **k):
        rod = orig(*a, **k)
        r = rod.recorded
        cfg = _capi.arm_push_config(1)
        o = oracle_c.OracleRod(cfg)
        o.set_radius_profile(np.asarray(r["base_radius"], np.float64))
        o.reset_straight(r["start"], r["direction"], r["normal"])
        fill_rod(rod, o)
        return rod
    refshim.CosseratRod.straight_rod = staticmethod(straight_rod)
    records = {"_about": "constructor arguments and operator registration order recorded while executing the reference's "
                         "ArmPushEnv._build and create_es_muscle_layers under recording stand-ins for COOMM "
                         "(tools/make_muscle_env_golden.py); no muscle force law is exercised"}
    out = {}
    for mode in ("discrete", "continuous"):
        tag = "d_" if mode == "discrete" else "c_"
        env = mod.ArmPushEnv(mode=mode)
        obs0, info0 = env.reset(seed=0)
        assert info0 == {}
        rod = env.shearable_rod
        rec = rod.recorded
        ops = []
        for op in env.simulator._ops:
            kw = {k: jsonable(v) for k, v in op["kwargs"].items() if k not in ("controller", "muscles", "callback_params_list")}
            ops.append({"kind": op["kind"], "cls": op["cls"].__name__, "kwargs": kw})
        layers = [{"kind": m.kind, **{k: jsonable(v) for k, v in m.kwargs.items()}} for m in env.muscle_layers]
        records[f"OctoArmPush ({mode})"] = {
            "init": {"step_skip": env.step_skip, "final_time": env.final_time, "time_step": env.time_step,
                     "n_elem": env.n_elem, "mode": env.mode, "obs_shape": env.observation_space.shape,
                     "action_space": type(env.action_space).__name__ if mode == "continuous" else "Discrete(2)",
                     "prev_action_initial": jsonable(np.asarray(env._prev_action))},
            "straight_rod": {k: jsonable(v) for k, v in rec.items()},
            "order": env.simulator.order(), "ops": ops, "muscle_layers": layers,
            "sucker": {"index": env.BC.index, "flag": bool(env.BC.flag), "reduction_ratio": env.BC.reduction_ratio},
        }
        out[tag + "reset_obs"] = np.asarray(obs0)

        # ---- step(): the reference's set_action / epilogue around a scripted stepper -------------------------
        orc = oracle_arm(mode)
        fill_rod(rod, orc)                      # the allocation's masses and the straight initial state
        S = Stack()
        rng = np.random.default_rng(5 if mode == "discrete" else 6)

        def ref_step(action, pre, post, time, label):
            rod.position_collection[:] = pre["x"]
            rod.velocity_collection[:] = pre["v"]
            rod.director_collection[:] = pre["Q"]
            rod.omega_collection[:] = pre["w"]
            rod.alpha_collection[:] = pre["alpha"]
            for m in env.muscle_layers:
                m.activations.clear()
            prev = np.asarray(env._prev_action, dtype=np.float64).copy()
            env.simulator._calls = 0

            def script(k, t, dt):
                if k == env.step_skip:
                    rod.position_collection[:] = post["x"]
                    rod.velocity_collection[:] = post["v"]
                    rod.director_collection[:] = post["Q"]
                    rod.omega_collection[:] = post["w"]
                    rod.alpha_collection[:] = post["alpha"]
                    return np.float64(time)
                return t
            env.simulator._script = script
            env.time = np.float64(0.0)
            obs, rew, term, trunc, info = env.step(action)
            assert env.simulator._calls == env.step_skip
            acts = np.full(3, np.nan)            # the LAST activation each layer received in this set_action (NaN: none)
            for i, m in enumerate(env.muscle_layers):
                if m.activations:
                    acts[i] = float(np.asarray(m.activations[-1]))
            S.add(label=label, action=np.atleast_1d(np.asarray(action, np.float64)).astype(np.float64)[:2] if mode == "continuous"
                  else np.array([float(action), 0.0]),
                  prev_action_before=np.resize(prev, 2) if prev.size else np.zeros(2),
                  pre_x=pre["x"], pre_v=pre["v"], pre_Q=pre["Q"], pre_w=pre["w"],
                  x=post["x"], v=post["v"], Q=post["Q"], w=post["w"], alpha=post["alpha"], time=np.float64(time),
                  obs=obs, reward=np.float64(rew), terminated=bool(term), truncated=bool(trunc),
                  info_time=np.float64(info["time"]), info_trunc=bool(info["TimeLimit.truncated"]),
                  sucker_index=int(env.BC.index), activations=acts)

        # a rollout of the oracle: pre = before the step, post = after it
        if mode == "discrete":
            script_actions = [0, 0, 1, 1, 0, 1]
        else:
            script_actions = [np.array(a, np.float32) for a in ([0.0, 0.8], [1.0, 0.3], [0.999, 0.5], [0.5, 0.0],
                                                                [0.0125, 1.0], [-0.2, 0.4], [1.7, 0.6])]
        t = 0.0
        for k, a in enumerate(script_actions):
            pre = state_of(orc)
            orc.env_step_push(np.atleast_1d(np.asarray(a, np.float32)))
            ref_step(a, pre, state_of(orc), orc.time, f"rollout{k}")
        base, t_end = state_of(orc), orc.time
        a_last = script_actions[-1]
        # NaN in each of the checked arrays (:298-309) and in none of them
        for label, key, idx in (("nan_x", "x", (1, 7)), ("nan_x0", "x", (0, 3)), ("nan_v", "v", (2, 40)), ("nan_v0", "v", (0, 0)),
                                ("nan_Q", "Q", (1, 2, 5)), ("nan_w", "w", (0, 39)), ("nan_alpha", "alpha", (2, 11))):
            st = {k: v.copy() for k, v in base.items()}
            st[key][idx] = np.nan
            ref_step(a_last, base, st, t_end, label)
        st = {k: v.copy() for k, v in base.items()}
        st["v"][0, 4] = np.inf                                           # nan_to_num is only applied when a NaN is present
        ref_step(a_last, base, st, t_end, "inf_v0")
        st["x"][0, 9] = np.nan
        ref_step(a_last, base, st, t_end, "nan_and_inf")
        # truncation: strict `>` (:321)
        ref_step(a_last, base, base, 2.5, "time_eq_final")
        ref_step(a_last, base, base, np.nextafter(2.5, 5.0), "time_just_past")
        # the reward is the change of |cm_xy|, also when the arm has moved sideways / backwards
        for k in range(3):
            st = {k_: v.copy() for k_, v in base.items()}
            st["x"][:2] += rng.normal(0, 0.05, (2, 1))
            ref_step(a_last, base, st, t_end, f"shifted{k}")
        out.update(S.arrays(tag + "step_"))

    # ---- ArmPullWeightEnv._build (arm_push_env.py:520-618): what it hands to Cylinder, BodyBoundaryCondition,
    #      FixedJoint2Rigid, the damper and the sucker; its step() is ArmPushEnv's (inherited) -------------------------
    env = mod.ArmPullWeightEnv(mode="continuous")
    obs0, _ = env.reset(seed=0)
    ops = []
    for op in env.simulator._ops:
        kw = {k: jsonable(v) for k, v in op["kwargs"].items() if k not in ("controller", "muscles", "callback_params_list")}
        ops.append({"kind": op["kind"], "cls": op["cls"].__name__, "kwargs": kw,
                    "targets": [env.simulator._systems.index(t) for t in op["targets"] if t in env.simulator._systems]})
    records["OctoArmPullWeight"] = {
        "init": {"step_skip": env.step_skip, "final_time": env.final_time, "time_step": env.time_step, "n_elem": env.n_elem,
                 "mode": env.mode, "obs_shape": env.observation_space.shape},
        "straight_rod": {k: jsonable(v) for k, v in env.shearable_rod.recorded.items()},
        "cylinder": {k: jsonable(v) for k, v in env.rigid_rod.recorded.items()},
        "order": env.simulator.order(), "ops": ops,
        "connect_indices": list(getattr(env.simulator, "_last_connect_idx", ())),
        "muscle_layers": [{"kind": m.kind, **{k: jsonable(v) for k, v in m.kwargs.items()}} for m in env.muscle_layers],
        "sucker": {"index": env.BC.index, "flag": bool(env.BC.flag), "reduction_ratio": env.BC.reduction_ratio},
    }
    out["w_reset_obs"] = np.asarray(obs0)
    try:
        mod.ArmPullWeightEnv(time_step=1e-5)
        records["OctoArmPullWeight"]["time_step_kwarg"] = "accepted"
    except TypeError as exc:
        records["OctoArmPullWeight"]["time_step_kwarg"] = "TypeError: " + str(exc)

    # ---- ControllableFixConstraint with the indices set_action produces (0, -1, 39) -----------------------------
    cc = refshim.load("gym_softrobot.envs.octopus.controllable_constraint")
    C = Stack()
    rng = np.random.default_rng(9)
    # `off`: "construction" = the controller is already off when the constraint is built — `controller or
    # SuckerController(index=index, ...)` then REPLACES it (SuckerController.__bool__ is its flag,
    # controllable_constraint.py:15-16,28-30) by a fresh one that is on, with the constraint's own index and ratio 1;
    # "later" = switched off after construction (turn_off, :21-22): constrain_rates does nothing
    for index, ratio, off in ((0, 1.0, ""), (-1, 1.0, ""), (39, 1.0, ""), (17, 0.9, ""), (-1, 0.25, ""),
                              (5, 0.5, "construction"), (5, 0.5, "later")):
        ctrl = cc.SuckerController(index=index, reduction_ratio=ratio)
        if off == "construction":
            ctrl.turn_off()
        bc = cc.ControllableFixConstraint(index=2, controller=ctrl)
        if off == "later":
            ctrl.turn_off()
        eff = bc.get_controller
        sysm = refshim.FakeRod(40)
        sysm.velocity_collection[:] = rng.normal(0, 1.0, (3, 41))
        sysm.omega_collection[:] = rng.normal(0, 2.0, (3, 40))
        v_in, w_in = sysm.velocity_collection.copy(), sysm.omega_collection.copy()
        bc.constrain_values(sysm, 0.0)
        bc.constrain_rates(sysm, 0.0)
        C.add(index=index, ratio=ratio, off=off, effective_index=int(eff.index), effective_ratio=float(eff.reduction_ratio),
              effective_flag=bool(eff.flag), v_in=v_in, w_in=w_in, v_out=sysm.velocity_collection, w_out=sysm.omega_collection)
    out.update(C.arrays("sucker_"))

    GOLD.mkdir(parents=True, exist_ok=True)
    np.savez_compressed(GOLD / "ref_armpush.npz", **out)
    (GOLD / "ref_muscle_build_records.json").write_text(json.dumps(jsonable(records), indent=1) + "\n")
    for f in ("ref_armpush.npz", "ref_muscle_build_records.json"):
        print(f, (GOLD / f).stat().st_size)


if __name__ == "__main__":
    main()
