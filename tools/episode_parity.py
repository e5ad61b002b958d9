#!/usr/bin/env python3
"""Episode-length parity, measured (1, 3 and 126 steps, "report achieved";
§7.3-1: a tolerance per horizon).  Run on the MI355X box:

    python tools/episode_parity.py > profiles/parity_episode.json

For every scenario the same action script drives (a) the HIP path, (b) the fp64 CPU oracle
(oracle/softrod_oracle.c, -ffp-contract=off) and (c) the CONTROL: the same oracle source built
with FMA contraction (-ffp-contract=fast -mfma) — two correct evaluations of the same algorithm
that differ only in rounding.  Per env.step it records the largest error over envs of the
observation, the reward and the node positions, for GPU-vs-oracle and control-vs-oracle.  The
pendulum of SoftPendulum-v0 starts INVERTED (soft_pendulum/build.py:47-51,88-91), so rounding
differences grow exponentially along an episode; the control curve is what ANY second
implementation — a GPU kernel, PyElastica on another CPU or another NumPy — can be asked to hold.

Scenarios: SoftPendulum-v0 for 126 steps (truncation fires on #126) under zero action, random
+-22 N, and a stabilising action script (a PD law evaluated on the ORACLE's observations, the
same numbers fed to all three); SoftPendulum3D-v0 for 125 steps; OctoArmSingle-v0 to its
truncation (201 steps).  A second pass re-synchronises the GPU state with the oracle's every
`window` steps (state-view injection) and records the error at the end of each window: parity
of the step map itself along the whole oracle trajectory, free of the accumulated divergence.
"""
import json
import sys
from pathlib import Path

import numpy as np
import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

import gym_softrobot_amd as gsa  # noqa: E402
from gym_softrobot_amd.envs.soft_pendulum_3d import initial_tilt  # noqa: E402
from gym_softrobot_amd.seeding import initial_angle, np_random  # noqa: E402
from oracle import oracle_c  # noqa: E402

FLOOR = 1e-3


def rel_scale(a, b):
    """max|a - b| / max|b|: relative to the field's own scale, no absolute floor (north_star's
    tolerance is relative; `rel` below carries a 1e-3 floor for entries that pass through zero)."""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if not np.isfinite(b).all() or not np.isfinite(a).all():
        return 0.0 if (np.isnan(a) == np.isnan(b)).all() else float("inf")
    m = float(np.max(np.abs(b)))
    return float(np.max(np.abs(a - b)) / m) if m > 0 else 0.0


def rel(a, b):
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    with np.errstate(invalid="ignore"):
        e = np.abs(a - b) / (np.abs(b) + FLOOR)
    e = np.where(np.isnan(a) & np.isnan(b), 0.0, e)
    return float(np.nanmax(e)) if e.size else 0.0


class Kind:
    """How one env kind is reset / stepped on the oracle and injected into the HIP state."""

    def __init__(self, env_id, step, reset, extra_inject=None):
        self.env_id, self.step, self.reset, self.extra_inject = env_id, step, reset, extra_inject


def pend_reset(r, env, i):
    r.reset_pendulum(initial_angle(np_random(i)[0]))


def pend3_reset(r, env, i):
    r.reset_pendulum3d(initial_tilt(np_random(i)[0]))


KINDS = {
    "SoftPendulum-v0": Kind("SoftPendulum-v0", lambda r, a: r.env_step(float(a[0])), pend_reset),
    "SoftPendulum3D-v0": Kind("SoftPendulum3D-v0", lambda r, a: r.env_step3d(a)[:4], pend3_reset),
    "OctoArmSingle-v0": Kind("OctoArmSingle-v0", lambda r, a: r.env_step_arm(a), lambda r, env, i: r.reset_arm()),
}


def inject(env, rods):
    """GPU state <- the oracle rods' (every array the step reads), through softrod_state_view."""
    be = env.backend
    st = be.state()
    dev = st["position"].device
    n = int(env.cfg.n_elem)

    def put(name, key, width, comps):
        a = np.stack([r.get(key).reshape(comps, width) for r in rods], axis=1)      # [comps][N][width]
        st[name][:, :, :width] = torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    put("position", "x", n + 1, 3)
    put("velocity", "v", n + 1, 3)
    put("omega", "w", n, 3)
    put("tangents", "tangents", n, 3)
    put("director", "Q", n, 9)
    put("kappa", "kappa", n - 1, 3)
    put("rest_kappa", "rest_kappa", n - 1, 3)
    st["time"][:] = torch.tensor([r.time for r in rods], dtype=torch.float64, device=dev)
    if env.cfg.env_kind == gsa._capi.ENV_SOFTPENDULUM3D:
        st["control"][:] = torch.from_numpy(np.stack([r.get("control") for r in rods], axis=1)).to(dev)
    if env.cfg.env_kind == gsa._capi.ENV_ARM_SINGLE:
        st["env_memory"][:, : n - 1] = torch.from_numpy(np.stack([r.get("prev_kappa") for r in rods])).to(dev)
        st["control"][0:2] = torch.from_numpy(np.stack([r.get("prev_com") for r in rods], axis=1)).to(dev)


def run(kind, n, steps, script, window=0, with_control=True, **kw):
    """script(t, oracle_obs[n, od]) -> actions[n, adim] float32.  window > 0: re-synchronise the GPU
    state with the oracle's before every window-th step.  with_control=False skips the FMA build
    (tests/test_gpu_episode_parity.py: its curves are zeros then)."""
    K = KINDS[kind]
    env = gsa.make_vec(K.env_id, n, device=0, **kw)
    obs0, _ = env.reset(seed=0)
    rods = [oracle_c.OracleRod(env.cfg) for _ in range(n)]
    ctrl = [oracle_c.OracleRod(env.cfg, "fma") for _ in range(n)] if with_control else rods
    for i in range(n):
        K.reset(rods[i], env, i)
        if with_control:
            K.reset(ctrl[i], env, i)
    o_obs = obs0.cpu().numpy().copy()
    curves = {k: [] for k in ("gpu_obs", "gpu_reward", "gpu_x", "ctl_obs", "ctl_reward", "ctl_x", "flags_equal",
                              "gpu_x_rel_scale", "ctl_x_rel_scale")}
    for t in range(steps):
        acts = np.asarray(script(t, o_obs), np.float32).reshape(n, env.action_dim)
        if window and t % window == 0 and t > 0:
            inject(env, rods)
        g_obs, g_rew, g_te, g_tr, _ = env.step(acts)
        g_obs, g_rew = g_obs.cpu().numpy(), g_rew.cpu().numpy()
        g_x = env.backend.state_numpy()["x"]
        eo = er = ex = co = cr = cx = xs = cs = 0.0
        same = True
        for i in range(n):
            o, rw, te, tr = K.step(rods[i], acts[i])
            c_o, c_rw = (o, rw) if not with_control else K.step(ctrl[i], acts[i])[:2]
            o_obs[i] = o
            eo, er, ex = max(eo, rel(g_obs[i], o)), max(er, rel(g_rew[i], rw)), max(ex, rel(g_x[i], rods[i].get("x")))
            co, cr, cx = max(co, rel(c_o, o)), max(cr, rel(c_rw, rw)), max(cx, rel(ctrl[i].get("x"), rods[i].get("x")))
            xs, cs = max(xs, rel_scale(g_x[i], rods[i].get("x"))), max(cs, rel_scale(ctrl[i].get("x"), rods[i].get("x")))
            same = same and bool(g_te[i]) == te and bool(g_tr[i]) == tr
        for k, v in (("gpu_obs", eo), ("gpu_reward", er), ("gpu_x", ex), ("ctl_obs", co), ("ctl_reward", cr),
                     ("ctl_x", cx), ("flags_equal", same), ("gpu_x_rel_scale", xs), ("ctl_x_rel_scale", cs)):
            curves[k].append(v)
    env.close()

    def horizon(c, tol=1e-5):
        bad = [t for t, v in enumerate(c) if not v <= tol]
        return len(c) if not bad else bad[0]       # env.steps that stayed within tol
    out = {"envs": n, "steps": steps, "window": window,
           "steps_within_1e-5": {k: horizon(np.maximum(curves[k + "_obs"], curves[k + "_reward"])) for k in ("gpu", "ctl")},
           "max": {k: float(np.max(curves[k])) for k in curves if k != "flags_equal"},
           "at_steps": {str(t): {k: curves[k][t - 1] for k in curves} for t in (1, 3, 10, 30, 60, 100, steps) if t <= steps},
           "flags_equal_all_steps": bool(all(curves["flags_equal"])),
           "curves": {k: [float(f"{v:.3e}") for v in curves[k]] for k in curves if k != "flags_equal"}}
    return out


def inject_octo(env, oracles):
    """GPU state <- the oracle envs' (arms, head, clock), through softrod_state_view."""
    st = env.backend.state()
    seg = st["arm_stride"]
    dev = st["position"].device
    T = torch.from_numpy
    for i, o in enumerate(oracles):
        for a in range(o.n_arm):
            arm = o.arm(a)
            lo = a * seg
            for name, key, comps in (("position", "x", 3), ("velocity", "v", 3), ("omega", "w", 3), ("director", "Q", 9),
                                     ("kappa", "kappa", 3), ("rest_kappa", "rest_kappa", 3)):
                v = np.ascontiguousarray(arm.get(key).reshape(comps, -1))
                st[name][:, i, lo : lo + v.shape[1]] = T(v).to(dev)
        h = o.head()
        st["head"][0:18, i] = T(np.concatenate([h["x"], h["v"], h["Q"].ravel(), h["w"]])).to(dev)
        st["time"][i] = o.time


def run_octo(n, steps, amax, window):
    """OctoFlat-v0 over a whole 5 s episode (35 env.steps of 2857 substeps).  Whole rollouts are
    chaotic at rounding level (the friction's stick-slip switches), so the curve of interest is the
    re-synchronised one: before every `window`-th step the HIP state and the control's are
    overwritten with the oracle's, i.e. the error of ONE env.step along the oracle's trajectory."""
    env = gsa.make_vec("OctoFlat-v0", n, device=0, numpy_output=True)
    env.reset(seed=0)
    orc = [oracle_c.OracleOcto(env.cfg) for _ in range(n)]
    ctl = [oracle_c.OracleOcto(env.cfg, "fma") for _ in range(n)]
    for i in range(n):
        orc[i].reset(env.targets[i])
        ctl[i].reset(env.targets[i])
    acts = np.random.default_rng(7).uniform(-amax, amax, (steps, n, 24)).astype(np.float32)
    flat = lambda ob: np.concatenate([ob["individual"].ravel(), ob["shared"]])   # noqa: E731
    curves = {k: [] for k in ("gpu_obs", "gpu_reward", "ctl_obs", "ctl_reward", "flags_equal", "crossings_equal")}
    for t in range(steps):
        if window and t % window == 0 and t > 0:
            inject_octo(env, orc)
            for i in range(n):
                ctl[i].copy_state_from(orc[i])
        g_obs, g_rew, g_te, g_tr, _ = env.step(acts[t])
        eo = er = co = cr = 0.0
        same = True
        for i in range(n):
            ob, rw, te, tr = orc[i].env_step(acts[t, i])
            ob2, rw2, _, _ = ctl[i].env_step(acts[t, i])
            eo, er = max(eo, rel(g_obs[i], flat(ob))), max(er, rel(g_rew[i], rw))
            co, cr = max(co, rel(flat(ob2), flat(ob))), max(cr, rel(rw2, rw))
            same = same and bool(g_te[i]) == te and bool(g_tr[i]) == tr
        for k, v in (("gpu_obs", eo), ("gpu_reward", er), ("ctl_obs", co), ("ctl_reward", cr), ("flags_equal", same)):
            curves[k].append(v)
    env.close()
    within = lambda c: int(np.sum(np.asarray(c) <= 1e-5))   # noqa: E731
    return {"envs": n, "steps": steps, "window": window, "action_amplitude": amax,
            "steps_within_1e-5_of": steps,
            "steps_within_1e-5": {"gpu": within(np.maximum(curves["gpu_obs"], curves["gpu_reward"])),
                                  "ctl": within(np.maximum(curves["ctl_obs"], curves["ctl_reward"]))},
            "max": {k: float(np.max(curves[k])) for k in ("gpu_obs", "gpu_reward", "ctl_obs", "ctl_reward")},
            "median": {k: float(np.median(curves[k])) for k in ("gpu_obs", "gpu_reward", "ctl_obs", "ctl_reward")},
            "flags_equal_all_steps": bool(all(curves["flags_equal"])),
            "curves": {k: [float(f"{v:.3e}") for v in curves[k]] for k in ("gpu_obs", "gpu_reward", "ctl_obs", "ctl_reward")}}


def pd_script(n):
    """The stabilising PD law on the ORACLE's observation (all paths get its numbers)."""
    st = {"prev": None}

    def pd(t, obs):
        x, v, th = obs[:, 0].astype(np.float64), obs[:, 1].astype(np.float64), obs[:, 3].astype(np.float64)
        dth = np.zeros_like(th) if st["prev"] is None or t == 0 else (th - st["prev"]) / 0.04
        st["prev"] = th.copy()
        return np.clip(100.0 * th + 20.0 * dth + 10.0 * x + 8.0 * v, -22, 22).astype(np.float32)[:, None]
    return pd


def pd_horizons(n=8):
    """`--pd-horizons`: the stabilised inverted pendulum (126 steps) on the LOADED library
    (SOFTROD_HIP_LIB selects a diagnostic build of tools/fastmath_cost.sh) in both math modes:
    env.steps within 1e-5 of the oracle, next to the control's.  One JSON object on stdout."""
    from gym_softrobot_amd import _capi

    out = {"library": str(_capi.library_path()), "library_source_hash": _capi.library_source_hash(), "envs": n}
    for name, mode in (("fast", _capi.MATH_FAST), ("libm", _capi.MATH_LIBM)):
        r = run("SoftPendulum-v0", n, 126, pd_script(n), with_control=(name == "fast"), math_mode=mode)
        out[name] = {"steps_within_1e-5": r["steps_within_1e-5"]["gpu"], "max": r["max"]["gpu_obs"],
                     "at_steps": {k: max(v["gpu_obs"], v["gpu_reward"]) for k, v in r["at_steps"].items()}}
        if name == "fast":
            out["control"] = {"steps_within_1e-5": r["steps_within_1e-5"]["ctl"], "max": r["max"]["ctl_obs"],
                              "at_steps": {k: max(v["ctl_obs"], v["ctl_reward"]) for k, v in r["at_steps"].items()}}
    print(json.dumps(out))


def main():
    if "--pd-horizons" in sys.argv:
        return pd_horizons()
    from gym_softrobot_amd import _capi

    LIBM = dict(math_mode=_capi.MATH_LIBM)
    doc = {"metric": "max over envs and entries of |a - oracle| / (|oracle| + 1e-3); gpu = HIP path, ctl = the same "
                     "oracle source built with FMA contraction (rounding control); tests assert 1e-5 (north_star)",
           "library_source_hash": _capi.library_source_hash(),
           "scenarios": {}}
    S = doc["scenarios"]
    n = 8
    rng = np.random.default_rng(1)
    rnd22 = rng.uniform(-22, 22, (126, n, 1)).astype(np.float32)
    st = {"prev": None}

    def pd(t, obs):                       # a PD law on the ORACLE's observation; all three paths get its numbers
        x, v, th = obs[:, 0].astype(np.float64), obs[:, 1].astype(np.float64), obs[:, 3].astype(np.float64)
        dth = np.zeros_like(th) if st["prev"] is None or t == 0 else (th - st["prev"]) / 0.04
        st["prev"] = th.copy()
        return np.clip(100.0 * th + 20.0 * dth + 10.0 * x + 8.0 * v, -22, 22).astype(np.float32)[:, None]

    for name, script in (("zero action", lambda t, o: np.zeros((n, 1), np.float32)),
                         ("random +-22 N", lambda t, o: rnd22[t]),
                         ("stabilising PD script", pd)):
        st["prev"] = None
        S[f"SoftPendulum-v0, 126 steps, {name}"] = run("SoftPendulum-v0", n, 126, script)
        st["prev"] = None
        S[f"SoftPendulum-v0, 126 steps, {name}, re-synchronised every 5 steps"] = run("SoftPendulum-v0", n, 126, script, window=5)
        # the libm kernel (SOFTROD_MATH_LIBM: the substep as PyElastica writes it) over the same episode
        st["prev"] = None
        S[f"SoftPendulum-v0, 126 steps, {name}, libm kernel"] = run("SoftPendulum-v0", n, 126, script, with_control=False, **LIBM)
    rnd1 = rng.uniform(-1, 1, (125, n, 2)).astype(np.float32)
    S["SoftPendulum3D-v0, 125 steps, random +-1"] = run("SoftPendulum3D-v0", n, 125, lambda t, o: rnd1[t])
    S["SoftPendulum3D-v0, 125 steps, random +-1, re-synchronised every 5 steps"] = run(
        "SoftPendulum3D-v0", n, 125, lambda t, o: rnd1[t], window=5)
    S["SoftPendulum3D-v0, 125 steps, random +-1, libm kernel"] = run(
        "SoftPendulum3D-v0", n, 125, lambda t, o: rnd1[t], with_control=False, **LIBM)
    m = 4
    rnd6 = rng.uniform(-6, 6, (201, m, 7)).astype(np.float32)
    S["OctoArmSingle-v0, 201 steps (to truncation), random +-6"] = run("OctoArmSingle-v0", m, 201, lambda t, o: rnd6[t])
    S["OctoArmSingle-v0, 201 steps, random +-6, re-synchronised every 5 steps"] = run(
        "OctoArmSingle-v0", m, 201, lambda t, o: rnd6[t], window=5)
    S["OctoArmSingle-v0, 201 steps (to truncation), random +-6, libm kernel"] = run(
        "OctoArmSingle-v0", m, 201, lambda t, o: rnd6[t], with_control=False, **LIBM)
    S["OctoFlat-v0, 36 steps of 2857 substeps (to truncation), random +-22, re-synchronised before every step"] = run_octo(4, 36, 22.0, 1)
    S["OctoFlat-v0, 36 steps, random +-5 (gentle), re-synchronised before every step"] = run_octo(4, 36, 5.0, 1)
    S["OctoFlat-v0, 8 steps, random +-22, free-running"] = run_octo(4, 8, 22.0, 0)
    print(json.dumps(doc, indent=1))


if __name__ == "__main__":
    main()
