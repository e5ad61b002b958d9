#!/usr/bin/env python3
"""Golden vectors produced by EXECUTING the reference's own env code (tools/refshim.py explains
how it can run here): for each of the four envs on the graded path

    SoftPendulumEnv   gym_softrobot/envs/soft_pendulum/soft_pendulum.py:59-251 + soft_pendulum/build.py:29-115
    SoftPendulum3DEnv gym_softrobot/envs/soft_pendulum_3d/soft_pendulum_3d.py:28-174 + soft_pendulum_3d/build.py:15-86
    ArmSingleEnv      gym_softrobot/envs/octopus/arm_single_env.py:55-316 + octopus/build.py:220-292
    FlatEnv           gym_softrobot/envs/octopus/flat_env.py:55-408 + octopus/build.py:52-217

the reference's `__init__`, `reset(seed)`, `set_action`, `step` and `get_state`, and the operator
classes its build functions define (PendulumBoundaryConditions, PendulumPointForces,
MovingBaseConstraint), are run on NON-TRIVIAL rod states (bent, moving; some with NaN; some past
`final_time`; clipped base commands; an arm at its target ...).  The states come from this repo's
fp64 oracle (a short rollout) or are synthetic — the stepper itself is PyElastica and is not on
disk — but everything the reference does around the stepper is the reference's code, untouched:

    reset:   the arguments build_* hands to CosseratRod.straight_rod / Cylinder / Plane / the
             contact, joint, damper and gravity operators, their REGISTRATION ORDER, the RNG draws,
             the reset observation (get_state on the freshly allocated rod)
    step:    set_action (point force / rest-kappa spline / base controller), the NaN and
             blow-up checks, reward, termination, truncation, info, the observation
    hooks:   constrain_values / constrain_rates / apply_forces of the env's own operator classes

Outputs: tests/golden/ref_softpendulum.npz, ref_softpendulum3d.npz, ref_armsingle.npz,
ref_octoflat.npz (+ ref_build_records.json with the recorded constructor arguments and operator
orders).  tests/test_reference_fixtures.py replays them through the oracle (CPU) and
tests/test_gpu_reference_fixtures.py through the HIP path (state-view injection).

    python tools/make_env_golden.py
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

import refshim  # noqa: E402

from gym_softrobot_amd import _capi  # noqa: E402
from oracle import oracle_c  # noqa: E402

GOLD = ROOT / "tests" / "golden"
warnings.filterwarnings("ignore", category=RuntimeWarning)     # NaN cases are deliberate


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


def op_records(sim):
    out = []
    for op in sim._ops:
        if op["kind"] == "append":
            rec = getattr(op["targets"][0], "recorded", {})
            out.append({"kind": "append", "cls": op["cls"].__name__, "recorded": jsonable(rec)})
        else:
            kw = {k: jsonable(v) for k, v in op["kwargs"].items()
                  if k not in ("point_force", "controller", "callback_params")}
            tg = [sim._systems.index(t) for t in op["targets"] if t in sim._systems]
            out.append({"kind": op["kind"], "cls": op["cls"].__name__, "targets": tg, "kwargs": kw})
    return out


def fill_rod(rod, orc):
    """Fake CosseratRod <- the oracle rod's arrays (PyElastica shapes)."""
    rod.position_collection[:] = orc.get("x")
    rod.velocity_collection[:] = orc.get("v")
    rod.director_collection[:] = orc.get("Q")
    rod.omega_collection[:] = orc.get("w")
    rod.tangents[:] = orc.get("tangents")
    rod.kappa[:] = orc.get("kappa")
    rod.rest_kappa[:] = orc.get("rest_kappa")
    rod.mass[:] = orc.get("mass")
    rod.lengths[:] = orc.get("lengths")
    rod.rest_lengths[:] = orc.get("rest_lengths")
    rod.radius[:] = orc.get("radius")


def rod_state(orc):
    return {k: orc.get(k).copy() for k in ("x", "v", "Q", "w", "tangents", "kappa")}


class Stack:
    """Collects per-case records and stacks them into arrays."""

    def __init__(self):
        self.rows = []

    def add(self, **kw):
        self.rows.append({k: np.array(v) for k, v in kw.items()})

    def arrays(self, prefix):
        keys = self.rows[0].keys()
        return {prefix + k: np.stack([r[k] for r in self.rows]) for k in keys}


def script_final(sim, n_calls, final_time, on_last):
    """Scripted stepper: nothing happens until the last of `n_calls` calls, which installs the
    post-loop state (on_last) and returns the post-loop time."""
    sim._calls = 0

    def script(k, time, dt):
        if k == n_calls:
            on_last()
            return np.float64(final_time)
        return time
    sim._script = script


# =============================================================================================
# SoftPendulum-v0
# =============================================================================================
def softpendulum(records):
    mod = refshim.load("gym_softrobot.envs.soft_pendulum.soft_pendulum")
    cfg = _capi.softpendulum_config(1)
    out = {}

    def on_rod(rod):                                   # straight_rod -> the oracle's allocation
        r = rod.recorded
        o = oracle_c.OracleRod(cfg)
        o.reset_straight(r["start"], r["direction"], r["normal"])
        fill_rod(rod, o)
        rod._oracle = o
    refshim.CosseratRod._on_create = staticmethod(on_rod)

    # ---- reset: recorded build arguments, RNG draw, reset observation -------------------------
    R = Stack()
    for seed in (0, 1, 2, 3, 42, 123, 2024):
        env = mod.SoftPendulumEnv()
        obs, info = env.reset(seed=seed)
        rec = env.shearable_rod.recorded
        assert info == {} and "shear_modulus" not in rec
        bc = [op["instance"] for op in env.simulator._ops if op["cls"].__name__ == "PendulumBoundaryConditions"][0]
        R.add(seed=seed, start=rec["start"], direction=rec["direction"], normal=rec["normal"], obs=obs,
              bc_fixed_position=bc.fixed_position, bc_fixed_directors=bc.fixed_directors)
        if seed == 0:
            records["SoftPendulum-v0"] = {
                "init": {"step_skip": env.step_skip, "final_time": env.final_time, "time_step": env.time_step,
                         "n_elems": env.n_elems, "action_low": env.action_space.low, "action_high": env.action_space.high,
                         "obs_shape": env.observation_space.shape},
                "order": env.simulator.order(), "ops": op_records(env.simulator)}
    out.update(R.arrays("reset_"))

    # ---- step: the reference's step() over states of an oracle rollout ------------------------
    S = Stack()
    env = mod.SoftPendulumEnv()
    env.reset(seed=5)
    rod = env.shearable_rod
    orc = rod._oracle
    rng = np.random.default_rng(17)
    acts = rng.uniform(-22, 22, 8).astype(np.float32)
    acts[3] = 0.0

    def ref_step(action, state, time, label):
        prev = env._prev_action.copy()

        def install():
            rod.position_collection[:] = state["x"]
            rod.velocity_collection[:] = state["v"]
            rod.director_collection[:] = state["Q"]
            rod.omega_collection[:] = state["w"]
            rod.tangents[:] = state["tangents"]
        script_final(env.simulator, env.step_skip, time, install)
        obs, rew, term, trunc, info = env.step(np.array([action], np.float32))
        assert env.simulator._calls == env.step_skip
        S.add(label=label, action=np.float32(action), prev_action_before=prev, x=state["x"], v=state["v"],
              Q=state["Q"], w=state["w"], tangents=state["tangents"], time=np.float64(time), obs=obs,
              reward=np.float64(rew), terminated=bool(term), truncated=bool(trunc), info_time=np.float64(info["time"]),
              info_trunc=bool(info["TimeLimit.truncated"]), point_force=np.float64(env.point_force[0]),
              prev_action_after=env._prev_action.copy())

    for t, a in enumerate(acts):
        orc.env_step(a)
        ref_step(a, rod_state(orc), orc.time, f"rollout{t}")
    base = rod_state(orc)
    t_end = orc.time
    # NaN in a position / in a velocity (terminated, -50); NaN only in omega (not checked -> valid)
    for label, key, idx in (("nan_x", "x", (1, 7)), ("nan_v", "v", (0, 3)), ("nan_w_only", "w", (1, 2))):
        st = {k: v.copy() for k, v in base.items()}
        st[key][idx] = np.nan
        ref_step(acts[1], st, t_end, label)
    # truncation is strict: time == final_time is not truncated, the next float is
    ref_step(acts[2], base, 5.0, "time_eq_final")
    ref_step(acts[2], base, np.nextafter(5.0, 10.0), "time_just_past")
    ref_step(acts[2], base, 4.999999999995016, "time_t125_two_half_adds")
    # theta = wrap(arctan(mean tx / mean ty)): every quadrant, ty -> 0, tx = ty = 0
    for label, ang in (("q1", 0.3), ("q2", 2.0), ("q3", -2.5), ("q4", -0.4), ("ty_tiny", np.pi / 2 - 1e-9)):
        st = {k: v.copy() for k, v in base.items()}
        jit = rng.normal(0, 0.05, 50)
        st["tangents"] = np.stack([np.sin(ang + jit), np.cos(ang + jit), np.zeros(50)])
        st["x"][0, 0] = rng.normal(0, 0.4)
        ref_step(acts[4], st, t_end, "theta_" + label)
    st = {k: v.copy() for k, v in base.items()}
    st["tangents"] = np.zeros((3, 50))
    ref_step(acts[4], st, t_end, "theta_zero_over_zero")
    out.update(S.arrays("step_"))

    # ---- the env's operator classes on arbitrary states ---------------------------------------
    env = mod.SoftPendulumEnv()
    env.reset(seed=9)
    ops = {op["cls"].__name__: op["instance"] for op in env.simulator._ops if op.get("instance") is not None}
    bc, pf = ops["PendulumBoundaryConditions"], ops["PendulumPointForces"]
    B = Stack()
    for case in range(10):
        sysm = refshim.FakeRod(50)
        sysm.position_collection[:] = rng.normal(0, 0.3, (3, 51))
        sysm.velocity_collection[:] = rng.normal(0, 1.0, (3, 51))
        sysm.director_collection[:] = rng.normal(0, 1.0, (3, 3, 50))
        sysm.omega_collection[:] = rng.normal(0, 2.0, (3, 50))
        sysm.external_forces[:] = rng.normal(0, 3.0, (3, 51))          # what gravity has added so far
        pin = {k: getattr(sysm, k).copy() for k in ("position_collection", "velocity_collection",
                                                   "director_collection", "omega_collection", "external_forces")}
        bc.constrain_values(sysm, 0.0)
        bc.constrain_rates(sysm, 0.0)
        force = np.float32(rng.uniform(-22, 22))
        env.point_force[:] = force
        pf.apply_forces(sysm, 0.0)
        B.add(x_in=pin["position_collection"], v_in=pin["velocity_collection"], Q_in=pin["director_collection"],
              w_in=pin["omega_collection"], f_in=pin["external_forces"], force=np.float64(force),
              fixed_position=bc.fixed_position, fixed_directors=bc.fixed_directors,
              x_out=sysm.position_collection, v_out=sysm.velocity_collection, Q_out=sysm.director_collection,
              w_out=sysm.omega_collection, f_out=sysm.external_forces)
    out.update(B.arrays("op_"))
    np.savez_compressed(GOLD / "ref_softpendulum.npz", **out)


# =============================================================================================
# SoftPendulum3D-v0
# =============================================================================================
def softpendulum3d(records):
    mod = refshim.load("gym_softrobot.envs.soft_pendulum_3d.soft_pendulum_3d")
    cfg = _capi.softpendulum3d_config(1)
    out = {}

    def on_rod(rod):
        r = rod.recorded
        o = oracle_c.OracleRod(cfg)
        o.reset_straight(r["start"], r["direction"], r["normal"])
        fill_rod(rod, o)
        rod._oracle = o
    refshim.CosseratRod._on_create = staticmethod(on_rod)

    R = Stack()
    for seed in (0, 1, 42, 123):
        env = mod.SoftPendulum3DEnv()
        env._prev_action[:] = 0.7                      # reset must clear it (soft_pendulum_3d.py:68)
        obs, info = env.reset(seed=seed)
        rec = env.shearable_rod.recorded
        assert "shear_modulus" not in rec
        R.add(seed=seed, start=rec["start"], direction=rec["direction"], normal=rec["normal"], obs=obs)
        if seed == 0:
            records["SoftPendulum3D-v0"] = {
                "init": {"step_skip": env.step_skip, "final_time": env.final_time, "time_step": env.time_step,
                         "n_elems": env.n_elems, "base_step": env.base_step, "base_limit": env.base_limit,
                         "action_low": env.action_space.low, "action_high": env.action_space.high},
                "order": env.simulator.order(), "ops": op_records(env.simulator)}
    out.update(R.arrays("reset_"))

    S = Stack()
    env = mod.SoftPendulum3DEnv()
    env.reset(seed=5)
    rod = env.shearable_rod
    orc = rod._oracle
    rng = np.random.default_rng(23)
    acts = rng.uniform(-1, 1, (8, 2)).astype(np.float32)
    acts[2] = [1.0, -1.0]

    def ref_step(action, state, time, label, ctrl=None):
        if ctrl is not None:                            # put the base controller somewhere first
            env.base_controller.position[:2] = ctrl[:2]
            env.base_controller.velocity[:2] = ctrl[2:]
        ctrl_before = np.concatenate([env.base_controller.position[:2], env.base_controller.velocity[:2]])
        prev = env._prev_action.copy()

        def install():
            rod.position_collection[:] = state["x"]
            rod.velocity_collection[:] = state["v"]
            rod.director_collection[:] = state["Q"]
            rod.omega_collection[:] = state["w"]
            rod.tangents[:] = state["tangents"]
        script_final(env.simulator, env.step_skip, time, install)
        obs, rew, term, trunc, info = env.step(np.asarray(action, np.float32))
        S.add(label=label, action=np.asarray(action, np.float32), prev_action_before=prev, ctrl_before=ctrl_before,
              x=state["x"], v=state["v"], Q=state["Q"], w=state["w"], tangents=state["tangents"],
              time=np.float64(time), obs=obs, reward=np.float64(rew), terminated=bool(term), truncated=bool(trunc),
              info_time=np.float64(info["time"]), info_tilt=np.float64(info["tilt"]),
              ctrl_after=np.concatenate([env.base_controller.position[:2], env.base_controller.velocity[:2]]),
              ctrl_pos_z=np.float64(env.base_controller.position[2]))

    for t, a in enumerate(acts):
        orc.env_step3d(a)
        ref_step(a, rod_state(orc), orc.time, f"rollout{t}")
    base = rod_state(orc)
    t_end = orc.time
    # clipping of the commanded base position at +-base_limit (np.clip), both axes / one axis
    ref_step(np.array([1.0, 1.0], np.float32), base, t_end, "clip_hi", ctrl=[0.4995, 0.5, 0.0, 0.0])
    ref_step(np.array([-1.0, 0.5], np.float32), base, t_end, "clip_lo", ctrl=[-0.49999, -0.2, 0.01, 0.0])
    for label, key, idx in (("nan_x", "x", (2, 30)), ("nan_v", "v", (1, 0))):
        st = {k: v.copy() for k, v in base.items()}
        st[key][idx] = np.nan
        ref_step(acts[1], st, t_end, label, ctrl=[0.1, -0.05, 0.0, 0.0])
    # truncation is `>=` here
    ref_step(acts[3], base, np.nextafter(5.0, 0.0), "time_just_before", ctrl=[0.1, -0.05, 0.0, 0.0])
    ref_step(acts[3], base, 5.0, "time_eq_final", ctrl=[0.1, -0.05, 0.0, 0.0])
    # tilt: large, beyond pi/2, clip of tangent_z at +-1
    for label, tz in (("tilt_mid", 0.6), ("tilt_down", -0.8), ("tilt_up_exact", 1.0)):
        st = {k: v.copy() for k, v in base.items()}
        ph = rng.uniform(0, 2 * np.pi, 50) if tz != 1.0 else np.zeros(50)
        s_ = np.sqrt(max(0.0, 1 - tz * tz))
        st["tangents"] = np.stack([s_ * np.cos(ph) * 0.2 + 0.1 * s_, s_ * np.sin(ph) * 0.2, np.full(50, tz)])
        ref_step(acts[5], st, t_end, label, ctrl=[0.3, 0.2, 0.0, 0.0])
    out.update(S.arrays("step_"))

    # action outside the Box raises ValueError (soft_pendulum_3d.py:116-117)
    raised = []
    for bad in (np.array([1.5, 0.0], np.float32), np.array([0.0, 0.0, 0.0], np.float32)):
        try:
            env.step(bad)
            raised.append(False)
        except ValueError:
            raised.append(True)
    records["SoftPendulum3D-v0"]["bad_action_raises_ValueError"] = raised

    # MovingBaseConstraint on arbitrary states
    env = mod.SoftPendulum3DEnv()
    env.reset(seed=9)
    bc = [op["instance"] for op in env.simulator._ops if op["cls"].__name__ == "MovingBaseConstraint"][0]
    B = Stack()
    for case in range(8):
        sysm = refshim.FakeRod(50)
        sysm.position_collection[:] = rng.normal(0, 0.3, (3, 51))
        sysm.velocity_collection[:] = rng.normal(0, 1.0, (3, 51))
        sysm.director_collection[:] = rng.normal(0, 1.0, (3, 3, 50))
        sysm.omega_collection[:] = rng.normal(0, 2.0, (3, 50))
        env.base_controller.position[:] = [rng.uniform(-0.5, 0.5), rng.uniform(-0.5, 0.5), 0.0]
        env.base_controller.velocity[:] = [rng.normal(0, 0.1), rng.normal(0, 0.1), 0.0]
        pin = {k: getattr(sysm, k).copy() for k in ("position_collection", "velocity_collection",
                                                   "director_collection", "omega_collection")}
        bc.constrain_values(sysm, 0.0)
        bc.constrain_rates(sysm, 0.0)
        B.add(x_in=pin["position_collection"], v_in=pin["velocity_collection"], Q_in=pin["director_collection"],
              w_in=pin["omega_collection"], ctrl=np.concatenate([env.base_controller.position[:2],
                                                                  env.base_controller.velocity[:2]]),
              fixed_height=np.float64(bc.fixed_height), fixed_director=bc.fixed_director,
              x_out=sysm.position_collection, v_out=sysm.velocity_collection, Q_out=sysm.director_collection,
              w_out=sysm.omega_collection)
    out.update(B.arrays("op_"))
    np.savez_compressed(GOLD / "ref_softpendulum3d.npz", **out)


# =============================================================================================
# OctoArmSingle-v0
# =============================================================================================
def armsingle(records):
    mod = refshim.load("gym_softrobot.envs.octopus.arm_single_env")
    cfg = _capi.arm_single_config(1)
    out = {}

    def on_rod(rod):
        r = rod.recorded
        o = oracle_c.OracleRod(cfg)
        o.reset_straight(r["start"], r["direction"], r["normal"])
        fill_rod(rod, o)
        rod._oracle = o
    refshim.CosseratRod._on_create = staticmethod(on_rod)

    env = mod.ArmSingleEnv()
    obs0, info = env.reset(seed=0)
    rec = env.shearable_rod.recorded
    records["OctoArmSingle-v0"] = {
        "init": {"step_skip": env.step_skip, "final_time": env.final_time, "time_step": env.time_step,
                 "n_elems": env.n_elems, "n_action": env.n_action, "control_penalty_coeff": env.control_penalty_coeff,
                 "kappa_range": env.kappa_range, "kappa_rate_range": env.kappa_rate_range,
                 "target": env._target, "action_low": env.action_space.low, "action_high": env.action_space.high},
        "order": env.simulator.order(), "ops": op_records(env.simulator)}
    out.update(reset_obs=obs0, reset_start=rec["start"], reset_direction=rec["direction"], reset_normal=rec["normal"],
               reset_prev_kappa=env.prev_kappa_state.copy(), reset_prev_com=env.prev_com_state.copy(),
               mass=env.shearable_rod.mass.copy())

    S = Stack()
    rod = env.shearable_rod
    orc = rod._oracle
    orc.reset_arm()
    rng = np.random.default_rng(29)
    acts = rng.uniform(-8, 8, (6, 7)).astype(np.float32)

    def ref_step(action, state, time, label, memory=None):
        if memory is not None:                           # put the env's own memory somewhere first
            env.prev_kappa_state[...] = memory[0]
            env.prev_com_state[...] = memory[1]
        mem_before = (env.prev_kappa_state.copy(), np.array(env.prev_com_state).copy())
        prev = env._prev_action.copy()

        def install():
            rod.position_collection[:] = state["x"]
            rod.velocity_collection[:] = state["v"]
            rod.director_collection[:] = state["Q"]
            rod.omega_collection[:] = state["w"]
            rod.tangents[:] = state["tangents"]
            rod.kappa[:] = state["kappa"]
        script_final(env.simulator, env.step_skip, time, install)
        obs, rew, term, trunc, info = env.step(np.asarray(action, np.float32))
        S.add(label=label, action=np.asarray(action, np.float32), prev_action_before=prev,
              prev_kappa_before=mem_before[0], prev_com_before=mem_before[1],
              x=state["x"], v=state["v"], Q=state["Q"], w=state["w"], tangents=state["tangents"], kappa=state["kappa"],
              time=np.float64(time), obs=obs, reward=np.float64(rew), terminated=bool(term), truncated=bool(trunc),
              info_time=np.float64(info["time"]), rest_kappa=rod.rest_kappa.copy(),
              prev_kappa_after=env.prev_kappa_state.copy(), prev_com_after=np.array(env.prev_com_state).copy())

    for t, a in enumerate(acts):
        orc.env_step_arm(a)
        ref_step(a, rod_state(orc), orc.time, f"rollout{t}")
    base = rod_state(orc)
    mem = (env.prev_kappa_state.copy(), np.array(env.prev_com_state).copy())
    t_end = orc.time
    for label, key, idx in (("nan_x", "x", (0, 11)), ("nan_v", "v", (2, 50))):
        st = {k: v.copy() for k, v in base.items()}
        st[key][idx] = np.nan
        ref_step(acts[1], st, t_end, label, mem)
    # |omega|_F > 250 terminates with -1; exactly at the threshold it does not
    st = {k: v.copy() for k, v in base.items()}
    st["w"] = rng.normal(0, 1, (3, 50))
    st["w"] *= 251.0 / np.linalg.norm(st["w"])
    ref_step(acts[2], st, t_end, "omega_blown", mem)
    st = {k: v.copy() for k, v in base.items()}
    st["w"] = np.zeros((3, 50))
    st["w"][1, 4] = 250.0
    ref_step(acts[2], st, t_end, "omega_at_threshold", mem)
    # centre of mass within 0.1 of the target (1, 0): +5 and terminated
    st = {k: v.copy() for k, v in base.items()}
    com = (st["x"] * env.shearable_rod.mass).sum(axis=1) / env.shearable_rod.mass.sum()
    st["x"][0] += 1.0 - com[0] - 0.06
    st["x"][1] += 0.0 - com[1] + 0.05
    ref_step(acts[3], st, t_end, "at_target", mem)
    ref_step(acts[3], base, 10.0, "time_eq_final", mem)
    ref_step(acts[3], base, np.nextafter(10.0, 20.0), "time_just_past", mem)
    ref_step(np.zeros(7, np.float32), base, t_end, "zero_action", mem)
    ref_step(np.full(7, 22.0, np.float32), base, t_end, "max_action", mem)
    out.update(S.arrays("step_"))
    np.savez_compressed(GOLD / "ref_armsingle.npz", **out)


# =============================================================================================
# OctoFlat-v0
# =============================================================================================
FLAT_FPS = 357          # int(1 / (357 * 7e-5)) = 40 substeps per env.step: short, not yet chaotic


def octoflat(records, n_arm=8, n_action=3, name="OctoFlat-v0", fname="ref_octoflat.npz", specials=True):
    mod = refshim.load("gym_softrobot.envs.octopus.flat_env")
    cfg = _capi.octo_flat_config(1, recording_fps=FLAT_FPS, n_arm=n_arm, n_action=n_action)
    assert int(cfg.n_substeps) == 40
    out = {}
    refshim.CosseratRod._on_create = None

    def fill_all(env, orc):
        for a, rod in enumerate(env.shearable_rods):
            fill_rod(rod, orc.arm(a))
        h = orc.head()
        hd = env.rigid_rod
        hd.position_collection[:, 0] = h["x"]
        hd.velocity_collection[:, 0] = h["v"]
        hd.director_collection[:, :, 0] = h["Q"]
        hd.omega_collection[:, 0] = h["w"]

    def snapshot(orc):
        arms = {k: np.stack([orc.arm(a).get(k) for a in range(n_arm)]) for k in ("x", "v", "Q", "w", "kappa", "rest_kappa")}
        h = orc.head()
        arms.update(head_x=h["x"].copy(), head_v=h["v"].copy(), head_Q=h["Q"].copy(), head_w=h["w"].copy())
        return arms

    # reset: get_state runs inside reset(), so the fake bodies are filled by a hook on finalize()
    orc = oracle_c.OracleOcto(cfg)
    state_holder = {}
    orig_finalize = refshim.BaseSystemCollection.finalize

    def finalize_and_fill(sim):
        orig_finalize(sim)
        env = state_holder["env"]
        env.shearable_rods = sim._systems[:n_arm]
        env.rigid_rod = sim._systems[n_arm]
        orc.reset([1.0, 1.0])                           # geometry only; the target is the env's draw
        fill_all(env, orc)
    refshim.BaseSystemCollection.finalize = finalize_and_fill
    R = Stack()
    for seed in (0, 1, 42):
        env = mod.FlatEnv(recording_fps=FLAT_FPS, n_arm=n_arm, n_action=n_action)
        state_holder["env"] = env
        obs, info = env.reset(seed=seed)
        R.add(seed=seed, target=env._target.copy(), individual=obs["individual"], shared=obs["shared"])
        if seed == 0:
            records[name] = {
                "init": {"step_skip": env.step_skip, "final_time": env.final_time, "time_step": env.time_step,
                         "n_elems": env.n_elems, "n_arm": env.n_arm, "n_action": env.n_action,
                         "default_step_skip": mod.FlatEnv(n_arm=n_arm, n_action=n_action).step_skip,
                         "action_low": env.action_space.low, "action_high": env.action_space.high},
                "order": env.simulator.order(), "ops": op_records(env.simulator)}
            out["arm_start"] = np.stack([r.recorded["start"] for r in env.shearable_rods])
            out["arm_direction"] = np.stack([r.recorded["direction"] for r in env.shearable_rods])
            out["arm_normal"] = np.stack([r.recorded["normal"] for r in env.shearable_rods])
    out.update(R.arrays("reset_"))

    # step: the reference's FlatEnv.step over an oracle rollout of 40-substep steps
    S = Stack()
    env = mod.FlatEnv(recording_fps=FLAT_FPS, n_arm=n_arm, n_action=n_action)
    state_holder["env"] = env
    env.reset(seed=7)
    orc.reset(env._target)
    fill_all(env, orc)
    rng = np.random.default_rng(31)
    acts = rng.uniform(-22, 22, (8, n_arm * n_action)).astype(np.float32)

    def ref_step(action, pre, post, time, label, target=None):
        if target is not None:
            env._target = np.array(target, dtype=np.float64)
        # the fake bodies hold the pre-loop state when step() is entered (xposbefore, :321)
        for a, rod in enumerate(env.shearable_rods):
            rod.position_collection[:] = pre["x"][a]
            rod.velocity_collection[:] = pre["v"][a]
            rod.kappa[:] = pre["kappa"][a]
        env.rigid_rod.position_collection[:, 0] = pre["head_x"]
        env.rigid_rod.velocity_collection[:, 0] = pre["head_v"]
        env.rigid_rod.director_collection[:, :, 0] = pre["head_Q"]
        prev = env._prev_action.copy()

        def install():
            for a, rod in enumerate(env.shearable_rods):
                rod.position_collection[:] = post["x"][a]
                rod.velocity_collection[:] = post["v"][a]
                rod.director_collection[:] = post["Q"][a]
                rod.omega_collection[:] = post["w"][a]
                rod.kappa[:] = post["kappa"][a]
            env.rigid_rod.position_collection[:, 0] = post["head_x"]
            env.rigid_rod.velocity_collection[:, 0] = post["head_v"]
            env.rigid_rod.director_collection[:, :, 0] = post["head_Q"]
            env.rigid_rod.omega_collection[:, 0] = post["head_w"]
        script_final(env.simulator, env.step_skip, time, install)
        obs, rew, term, trunc, info = env.step(np.asarray(action, np.float32))
        S.add(label=label, action=np.asarray(action, np.float32), prev_action_before=prev, target=env._target.copy(),
              **{"pre_" + k: v for k, v in pre.items()}, **{"post_" + k: v for k, v in post.items()},
              time=np.float64(time), individual=obs["individual"], shared=obs["shared"], reward=np.float64(rew),
              terminated=bool(term), truncated=bool(trunc), info_time=np.float64(info["time"]),
              rest_kappa=np.stack([r.rest_kappa.copy() for r in env.shearable_rods]))

    pre = snapshot(orc)
    for t, a in enumerate(acts):
        orc.env_step(a)
        post = snapshot(orc)
        ref_step(a, pre, post, orc.time, f"rollout{t}")
        pre = post
    if specials:
        base_pre, base_post, t_end = S.rows[-1], post, orc.time
        pre_last = {k[4:]: v for k, v in base_pre.items() if k.startswith("pre_")}
        for label, key, idx in (("nan_x", "x", (3, 0, 5)), ("nan_v", "v", (7, 1, 0))):
            st = {k: v.copy() for k, v in base_post.items()}
            st[key][idx] = np.nan
            ref_step(acts[1], pre_last, st, t_end, label)
        # head within 0.1 of the target: +100, terminated, reward -= dist - 0.1
        tgt = base_post["head_x"][:2] + np.array([0.05, -0.03])
        ref_step(acts[2], pre_last, base_post, t_end, "at_target", target=tgt)
        ref_step(acts[2], pre_last, base_post, 5.0, "time_eq_final", target=[1.2, 0.9])
        ref_step(acts[2], pre_last, base_post, np.nextafter(5.0, 10.0), "time_just_past", target=[1.2, 0.9])
        # crossing arms (radial arms never cross, so three arms are laid across others as straight
        # polylines): arm 1 across arm 0 and arm 7 across arm 0 are counted — pairs (0, 1) and (7, 0) —
        # arm 6 across the displaced arm 7 is not: the loop never tests the pair (6, 7) (flat_env.py:347-357)
        st = {k: v.copy() for k, v in base_post.items()}
        x0 = st["x"][0]

        def lay(arm, p, q):
            for c in range(2):
                st["x"][arm, c] = np.linspace(p[c], q[c], 11)
        up = np.array([0.0, 1.0])
        lay(1, 0.5 * (x0[:2, 2] + x0[:2, 3]) + 0.04 * up, 0.5 * (x0[:2, 6] + x0[:2, 7]) - 0.04 * up)
        lay(7, 0.5 * (x0[:2, 8] + x0[:2, 9]) + 0.03 * up, 0.5 * (x0[:2, 9] + x0[:2, 10]) - 0.03 * up)
        x7 = st["x"][7]
        lay(6, x7[:2, 2] + np.array([0.02, 0.001]), x7[:2, 3] - np.array([0.02, 0.0]))
        ref_step(acts[3], pre_last, st, t_end, "crossing", target=[1.2, 0.9])
    out.update(S.arrays("step_"))
    refshim.BaseSystemCollection.finalize = orig_finalize
    np.savez_compressed(GOLD / fname, **out)


# =============================================================================================
# ControllableFixConstraint ("sucker"), octopus/controllable_constraint.py:24-69
# =============================================================================================
def sucker(records):
    mod = refshim.load("gym_softrobot.envs.octopus.controllable_constraint")
    rng = np.random.default_rng(41)
    B = Stack()
    n = 20
    for case in range(12):
        index = int(rng.integers(0, n))
        ratio = [1.0, 0.9, 0.3, 0.0][case % 4]
        ctrl = mod.SuckerController(index=index, reduction_ratio=ratio)
        bc = mod.ControllableFixConstraint(index=index, controller=ctrl)
        if case % 5 == 4:
            ctrl.turn_off()
        sysm = refshim.FakeRod(n)
        sysm.position_collection[:] = rng.normal(0, 0.3, (3, n + 1))
        sysm.velocity_collection[:] = rng.normal(0, 1.0, (3, n + 1))
        sysm.director_collection[:] = rng.normal(0, 1.0, (3, 3, n))
        sysm.omega_collection[:] = rng.normal(0, 2.0, (3, n))
        pin = {k: getattr(sysm, k).copy() for k in ("position_collection", "velocity_collection",
                                                   "director_collection", "omega_collection")}
        bc.constrain_values(sysm, 0.0)
        bc.constrain_rates(sysm, 0.0)
        B.add(index=index, ratio=np.float64(ratio), flag=bool(ctrl), x_in=pin["position_collection"],
              v_in=pin["velocity_collection"], Q_in=pin["director_collection"], w_in=pin["omega_collection"],
              x_out=sysm.position_collection, v_out=sysm.velocity_collection, Q_out=sysm.director_collection,
              w_out=sysm.omega_collection)
    default = mod.SuckerController(index=0)
    records["ControllableFixConstraint"] = {"default_reduction_ratio": default.reduction_ratio,
                                            "default_flag": bool(default)}
    np.savez_compressed(GOLD / "ref_sucker.npz", **B.arrays("op_"))


# =============================================================================================
# diagnostic callbacks, utils/custom_elastica/callback_func.py:4-41
# =============================================================================================
def callbacks(records):
    mod = refshim.load("gym_softrobot.utils.custom_elastica.callback_func")
    from collections import defaultdict
    out = {}
    rod = refshim.FakeRod(6)
    rod.dilatation = np.ones(6)
    rod.voronoi_dilatation = np.ones(5)
    body = refshim.FakeRigidBody()
    for cls, system in ((mod.RodCallBack, rod), (mod.RigidCylinderCallBack, body)):
        d = defaultdict(list)
        cb = cls(step_skip=4, callback_params=d)
        fired = []
        for step in range(0, 13):
            before = len(d["time"])
            cb.make_callback(system, 0.1 * step, step)
            if len(d["time"]) > before:
                fired.append(step)
        out[cls.__name__] = {"fields": list(d.keys()), "fires_at_steps_with_step_skip_4": fired,
                             "shapes": {k: list(np.shape(v[0])) for k, v in d.items()}}
    records["callbacks"] = out


def main():
    refshim.install()
    # straight_rod calls back into the generator so that the fake rod holds a real allocation
    orig = refshim.CosseratRod.straight_rod
    refshim.CosseratRod._on_create = None

    def straight_rod(*a, **k):
        rod = orig(*a, **k)
        if refshim.CosseratRod._on_create is not None:
            refshim.CosseratRod._on_create(rod)
        return rod
    refshim.CosseratRod.straight_rod = staticmethod(straight_rod)
    oracle_c.build()
    GOLD.mkdir(parents=True, exist_ok=True)
    records = {"_about": "constructor arguments and operator registration order recorded while executing the "
                         "reference's build_* functions (tools/make_env_golden.py)"}
    softpendulum(records)
    softpendulum3d(records)
    armsingle(records)
    octoflat(records)
    # OctoFlatLite-v0: the same class registered with n_arm = 1, n_action = 8 (gym_softrobot/__init__.py:11-15)
    octoflat(records, n_arm=1, n_action=8, name="OctoFlatLite-v0", fname="ref_octoflatlite.npz", specials=False)
    sucker(records)
    callbacks(records)
    (GOLD / "ref_build_records.json").write_text(json.dumps(jsonable(records), indent=1) + "\n")
    for f in sorted(GOLD.glob("ref_*")):
        print(f.name, f.stat().st_size)


if __name__ == "__main__":
    main()
