"""Shared pieces of the one-command PyElastica pin (TEST-FIXTURE TOOLING; nothing here is on the
product path):

    tools/make_pyelastica_golden.py   where `import elastica` works: run the REFERENCE
                                      (/root/reference/gym_softrobot on pyelastica 1.0.0) and dump
                                      tests/golden/pyelastica_<env>_seed<k>.npz
    tests/test_pyelastica_fixtures.py oracle (CPU) and HIP (-m gpu) against those files at 1e-5;
                                      skipped while the files are absent
    tools/sweep_switches.py           which combination of the RECALLED PyElastica details
                                      (SURVEY.md App. A "(?)" items = fields of softrod_config)
                                      reproduces the fixtures

One record layout, three drivers that fill / replay it:
    PyElasticaDriver   the reference's own env object (gym.make), stepped by PyElastica
    OracleDriver       this repo's env classes over the C oracle (tests/oracle_backend.py)
    HipDriver          this repo's env classes over libsoftrod_hip.so
`record_case(driver, ...)` produces the dict a fixture file holds; `compare_case(driver, fixture)`
replays the fixture's stored actions through a driver and returns the worst relative deviation per
record.  Deviation metric: max |a - b| / max(max |b|, floor) over an array, i.e. relative to the array's own
scale (a node coordinate that passes through zero does not blow the ratio up) with a per-field floor
under the scale (FLOOR below).

Record schedule (VERDICT r3 "next" #2): from a reset, rod state after 1 / 10 / 100 RAW substeps under
zero action; then, from a fresh reset of the same seed, env.steps under the stored action script with
obs / reward / flags / time after EVERY step and the full rod state after steps 1, 3, 10 and 126
(where the episode is that long: SoftPendulum truncates on step 126 or 125 depending on how the clock
accumulates — the `time_two_half_adds` switch).
"""
from __future__ import annotations

import sys
from pathlib import Path
from typing import Dict, List, Optional

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

SEEDS = (0, 1, 42, 123)
RAW_SUBSTEPS = (1, 10, 100)
# env id -> action amplitude of the script, env.steps recorded with full state, steps run in total, and
# `strict_steps`: the env.steps up to which 1e-5 is demanded of ANY second implementation.  Beyond it
# the records are kept and reported, not asserted: OctoFlat (8 arms + head, stiff joints, 2857 substeps
# per env.step) amplifies a rounding-level difference to 1e-2 by step 3 — the oracle built with FMA
# contraction against itself without, tools/episode_parity.py and DESIGN.md §3 — while the one-rod envs
# hold 1e-7 or better over the whole episode under the same control.
ENVS = {
    "SoftPendulum-v0": dict(amax=22.0, state_steps=(1, 3, 10, 126), n_steps=126, strict_steps=10),
    "SoftPendulum3D-v0": dict(amax=1.0, state_steps=(1, 3, 10, 126), n_steps=126, strict_steps=10),
    "OctoArmSingle-v0": dict(amax=6.0, state_steps=(1, 3, 10, 126), n_steps=126, strict_steps=10),
    "OctoFlat-v0": dict(amax=22.0, state_steps=(1, 3, 10), n_steps=10, strict_steps=1),
}

# the COOMM muscle arm.  Needs `import coomm` next to `import elastica` on the generating side
# (uv.lock:173-175); kept apart from ENVS so that the four graded envs' tooling is unchanged.  `raw`: no raw-substep
# records (an unactuated straight arm without gravity does not move); `script`: how the stored actions are drawn —
# "strokes01": 0 / 1 alternating every step from a random phase (alternations that hold an activation for two steps
# and release for one drive single elements of THIS repo's restatement to stretches of 1e3, see DESIGN.md section 3;
# the reference may or may not: the fixtures will tell), "unit": uniform in [0, 1].  strict_steps 3: the horizon of the
# reference's own determinism test (tests/envs/test_determinism.py:46-54).
MUSCLE_ENVS = {
    "OctoArmPush-v0": dict(amax=1.0, state_steps=(1, 3, 10, 101), n_steps=102, strict_steps=3, raw=False, script="strokes01",
                           mode="discrete"),
    "OctoArmPush-v1": dict(amax=1.0, state_steps=(1, 3, 10, 101), n_steps=102, strict_steps=3, raw=False, script="unit",
                           mode="continuous"),
    # the arm with a rigid weight, and the muscle octopus (build_muscle_octopus.py): ArmTwo / Reach drive the
    # longitudinal layers, so their fixtures decide the two longitudinal-geometry details the push arm cannot see
    "OctoArmPullWeight-v0": dict(amax=1.0, state_steps=(1, 3, 10), n_steps=10, strict_steps=3, raw=False, script="unit"),
    "OctoCrawl-v0": dict(amax=1.0, state_steps=(1, 3, 10), n_steps=10, strict_steps=3, raw=False, script="unit06"),
    "OctoArmTwo-v0": dict(amax=1.0, state_steps=(1, 3, 10), n_steps=10, strict_steps=3, raw=False, script="unit06"),
    "OctoReach-v0": dict(amax=1.0, state_steps=(1, 3, 10), n_steps=10, strict_steps=3, raw=False, script="unit06"),
}
MUSCLE_OCTOPUS_IDS = {"OctoCrawl-v0": "ENV_CRAWL", "OctoArmTwo-v0": "ENV_ARM_TWO", "OctoReach-v0": "ENV_REACH"}
ENVS_ALL = dict(ENVS, **MUSCLE_ENVS)

# The recalled COOMM details a muscle-env fixture can decide (fields of softrod_config honoured by the oracle, the
# NumPy twin and both HIP kernels, and the two constructor behaviours of _capi.es_muscle_layers); first = shipped.
MUSCLE_SWITCHES = {
    "muscle_equiv_load_form": (0, 1),
    "muscle_position_current_radius": (1, 0),
    "muscle_tm_length_law": (0, 1),
    "muscle_init_angle_rotates": (True, False),
    "muscle_tm_sign": (-1.0, 1.0),
}

# The recalled details a fixture can decide, as (name, candidates).  The first candidate of each is
# what the repo ships (gym_softrobot_amd/_capi.py _common / *_config).  Every one is a field of
# softrod_config — honoured by the oracle, the NumPy twin AND the HIP library, so whatever combination a
# PyElastica fixture selects needs no kernel work — except `shear_modulus_over_E`, which scales
# cfg.shear_modulus.
SWITCHES = {
    "alpha_c": (27.0 / 28.0, 4.0 / 3.0, 5.0 / 6.0, 1.0),
    "shear_modulus_over_E": (1.0 / 3.0, 1.0 / 1.5),
    "damp_before_constrain": (0, 1),
    "contact_before_forcing": (0, 1),
    "damper_protocol": ("per_unit_mass", "uniform"),
    "time_two_half_adds": (1, 0),
    "eps_length": (1e-14, 0.0),
    "eps_rot_axis": (1e-14, 0.0),
    "acos_shift": (1e-10, 0.0),
    "eps_sin": (1e-14, 0.0),
}
COARSE = ("alpha_c", "shear_modulus_over_E", "damp_before_constrain", "contact_before_forcing", "damper_protocol")
FINE = ("time_two_half_adds", "eps_length", "eps_rot_axis", "acos_shift", "eps_sin")


def default_switches() -> Dict[str, object]:
    return {k: v[0] for k, v in SWITCHES.items()}


def action_script(env_id: str, seed: int, adim: int) -> np.ndarray:
    """The fixture's actions: uniform in the env's box, float32, drawn from NumPy alone so that the
    script does not depend on gymnasium's Box.sample (the fixture stores them anyway)."""
    spec = ENVS_ALL[env_id]
    rng = np.random.default_rng(100_000 + seed)
    if spec.get("script") == "strokes01":
        return ((np.arange(spec["n_steps"]) + int(rng.integers(0, 2))) % 2).astype(np.float32).reshape(-1, 1)
    if spec.get("script") == "unit":
        return rng.uniform(0.0, 1.0, (spec["n_steps"], adim)).astype(np.float32)
    if spec.get("script") == "unit06":      # Box(0, 1) actions kept inside the restated force-length law's range
        return rng.uniform(0.0, 0.6, (spec["n_steps"], adim)).astype(np.float32)
    return rng.uniform(-spec["amax"], spec["amax"], (spec["n_steps"], adim)).astype(np.float32)


# A field that is physically zero (the out-of-plane velocity of an arm at rest on the plane) holds
# rounding noise of ~1e-15 in both implementations: its own scale says nothing.  Each field therefore
# has a floor under its scale; 1e-5 x floor is the absolute tolerance the 1e-5 bar amounts to there
# (x: 1e-8 m, v: 1e-8 m/s, omega: 1e-7 rad/s, directors: 1e-5, observations / rewards: 1e-7).
FLOOR = {"x": 1e-3, "v": 1e-3, "w": 1e-2, "Q": 1.0, "obs": 1e-2, "reward": 1e-2}


def floor_of(record_name: str) -> float:
    return FLOOR[record_name.rsplit("_", 1)[-1]]


def deviation(a, b, atol_scale: float = 1e-12) -> float:
    """max |a - b| / max(max |b|, atol_scale)"""
    a, b = np.asarray(a, np.float64), np.asarray(b, np.float64)
    if a.shape != b.shape:
        return float("inf")
    if a.size == 0:
        return 0.0
    nan_a, nan_b = np.isnan(a), np.isnan(b)
    if (nan_a != nan_b).any():
        return float("inf")
    d = np.abs(np.where(nan_a, 0.0, a) - np.where(nan_b, 0.0, b)).max()
    return float(d / max(atol_scale, np.abs(np.where(nan_b, 0.0, b)).max()))


# ---------------------------------------------------------------------------------------------
# drivers
# ---------------------------------------------------------------------------------------------
class _RepoDriver:
    """This repo's batched env (N = 1) over a given backend kind, with the switches applied to the
    softrod_config the backend is built from."""

    kind = "repo"

    def __init__(self, env_id: str, switches: Optional[Dict[str, object]] = None):
        import gym_softrobot_amd as gsa
        from gym_softrobot_amd import _capi
        from gym_softrobot_amd.envs.base import time_table

        self.env_id = env_id
        self.sw = {**default_switches(), **{k: v[0] for k, v in MUSCLE_SWITCHES.items()}, **(switches or {})}
        self.muscle = env_id in MUSCLE_ENVS
        maker = {"SoftPendulum-v0": _capi.softpendulum_config, "SoftPendulum3D-v0": _capi.softpendulum3d_config,
                 "OctoArmSingle-v0": _capi.arm_single_config, "OctoFlat-v0": _capi.octo_flat_config,
                 "OctoArmPush-v0": lambda n: _capi.arm_push_config(n, mode="discrete"),
                 "OctoArmPush-v1": lambda n: _capi.arm_push_config(n, mode="continuous"),
                 "OctoArmPullWeight-v0": _capi.arm_pull_weight_config,
                 **{k: (lambda n, kind=getattr(_capi, v): _capi.muscle_octopus_config(kind, n)) for k, v in MUSCLE_OCTOPUS_IDS.items()}}[env_id]
        cfg = maker(1)
        self._apply(cfg)
        extra = {}
        if self.muscle:
            extra["muscle_kwargs"] = dict(init_angle_rotates=bool(self.sw["muscle_init_angle_rotates"]),
                                          tm_sign=float(self.sw["muscle_tm_sign"]))
        self.env = gsa.make_vec(env_id, 1, backend=self._backend(cfg), numpy_output=True, **extra)
        self._apply(self.env.cfg)                      # the env's own copy: the host clock table reads it
        self.env._time_tab = time_table(self.env.cfg, 128)
        self.octo = env_id == "OctoFlat-v0" or env_id in MUSCLE_OCTOPUS_IDS      # several arms + a rigid head
        self.pull = env_id == "OctoArmPullWeight-v0"                               # one arm + a rigid weight
        self.adim = self.env.action_dim

    def _apply(self, cfg) -> None:
        for k in ("alpha_c", "damp_before_constrain", "contact_before_forcing", "time_two_half_adds",
                  "eps_length", "eps_rot_axis", "acos_shift", "eps_sin"):
            setattr(cfg, k, type(getattr(cfg, k))(self.sw[k]))
        if not getattr(self, "muscle", False):          # the muscle arm passes shear_modulus explicitly (arm_push_env.py:176)
            cfg.shear_modulus = float(cfg.youngs_modulus) * float(self.sw["shear_modulus_over_E"])
        cfg.damper_protocol = {"per_unit_mass": 0, "uniform": 1}[self.sw["damper_protocol"]]
        if getattr(self, "muscle", False):
            for k in ("muscle_equiv_load_form", "muscle_position_current_radius", "muscle_tm_length_law"):
                setattr(cfg, k, int(self.sw[k]))

    def _after_reset(self) -> None:
        pass

    def reset(self, seed: int):
        obs, _ = self.env.reset(seed=int(seed))
        self._after_reset()
        return np.asarray(obs[0], np.float32).copy()

    def step(self, action):
        o, r, te, tr, info = self.env.step(np.asarray(action, np.float32).reshape(1, self.adim))
        return (np.asarray(o[0], np.float32).copy(), float(np.asarray(r)[0]), bool(np.asarray(te)[0]),
                bool(np.asarray(tr)[0]), float(np.asarray(info["time"])[0]))

    def close(self) -> None:
        self.env.close()


class OracleDriver(_RepoDriver):
    kind = "oracle"

    def _backend(self, cfg):
        from tests.oracle_backend import OracleBackend

        return OracleBackend(cfg)

    def _rods(self):
        r = self.env.backend.rods[0]
        return [r.arm(a) for a in range(r.n_arm)] if self.octo else [r]

    def substeps(self, n: int) -> None:
        r = self.env.backend.rods[0]
        if self.octo:
            r.substeps(int(n))
        else:
            r.substeps(0.0, int(n))

    def state(self) -> Dict[str, np.ndarray]:
        r = self.env.backend.rods[0]
        if self.pull:
            a, h = r.arm(0), r.head()
            return {"x": a.get("x"), "v": a.get("v"), "Q": a.get("Q"), "w": a.get("w"), "head_x": h["x"].copy(),
                    "head_v": h["v"].copy(), "head_Q": h["Q"].copy(), "head_w": h["w"].copy(), "time": np.float64(r.time)}
        if self.octo:
            arms = [r.arm(a) for a in range(r.n_arm)]
            h = r.head()
            return {"x": np.stack([a.get("x") for a in arms]), "v": np.stack([a.get("v") for a in arms]),
                    "Q": np.stack([a.get("Q") for a in arms]), "w": np.stack([a.get("w") for a in arms]),
                    "head_x": h["x"].copy(), "head_v": h["v"].copy(), "head_Q": h["Q"].copy(), "head_w": h["w"].copy(),
                    "time": np.float64(r.time)}
        return {"x": r.get("x"), "v": r.get("v"), "Q": r.get("Q"), "w": r.get("w"), "time": np.float64(r.time)}


class HipDriver(_RepoDriver):
    kind = "hip"

    def __init__(self, env_id, switches=None, math_mode=None):
        self._math_mode = math_mode
        super().__init__(env_id, switches)

    def _backend(self, cfg):
        from gym_softrobot_amd.backend import HipRodBackend

        if self._math_mode is not None:
            cfg.math_mode = int(self._math_mode)
        return HipRodBackend(cfg, 0)

    def substeps(self, n: int) -> None:
        self.env.backend.substeps(None, int(n))

    def state(self) -> Dict[str, np.ndarray]:
        be = self.env.backend
        if self.octo:
            s = be.octo_state_numpy()
            return {"x": s["x"][0], "v": s["v"][0], "Q": s["Q"][0], "w": s["w"][0], "head_x": s["head_x"][0],
                    "head_v": s["head_v"][0], "head_Q": s["head_Q"][0], "head_w": s["head_w"][0],
                    "time": np.float64(s["time"][0])}
        s = be.state_numpy()
        out = {"x": s["x"][0], "v": s["v"][0], "Q": s["Q"][0], "w": s["w"][0], "time": np.float64(s["time"][0])}
        if self.pull:
            hd = be.state()["head"].cpu().numpy()[:, 0]
            out.update(head_x=hd[0:3].copy(), head_v=hd[3:6].copy(), head_Q=hd[6:15].reshape(3, 3).copy(), head_w=hd[15:18].copy())
        return out


class PyElasticaDriver:
    """The reference itself: `gym.make(env_id)` from /root/reference/gym_softrobot, stepped by
    pyelastica 1.0.0.  Only constructible where `import elastica`, `import gymnasium` and the
    reference's other imports succeed (NOT in the build container, NOT on the GPU box).  Reads the
    attributes the reference's env classes keep: `simulator`, `do_step`, `time`, `time_step`,
    `shearable_rod` (soft_pendulum.py:115-139, soft_pendulum_3d.py:65-92, arm_single_env.py:142-163) or
    `shearable_rods` + `rigid_rod` (flat_env.py:179-218)."""

    kind = "pyelastica"
    # gym_softrobot/__init__.py:6-9,27-30,74-80: the entry points of the four ids (no kwargs)
    ENTRY = {"SoftPendulum-v0": ("gym_softrobot.envs.soft_pendulum.soft_pendulum", "SoftPendulumEnv"),
             "SoftPendulum3D-v0": ("gym_softrobot.envs.soft_pendulum_3d.soft_pendulum_3d", "SoftPendulum3DEnv"),
             "OctoArmSingle-v0": ("gym_softrobot.envs.octopus.arm_single_env", "ArmSingleEnv"),
             "OctoFlat-v0": ("gym_softrobot.envs.octopus.flat_env", "FlatEnv"),
             # gym_softrobot/__init__.py:37-46
             "OctoArmPush-v0": ("gym_softrobot.envs.octopus.arm_push_env", "ArmPushEnv"),
             "OctoArmPush-v1": ("gym_softrobot.envs.octopus.arm_push_env", "ArmPushEnv"),
             # gym_softrobot/__init__.py:17-25,32-35,48-52
             "OctoArmPullWeight-v0": ("gym_softrobot.envs.octopus.arm_push_env", "ArmPullWeightEnv"),
             "OctoCrawl-v0": ("gym_softrobot.envs.octopus.crawl_env", "CrawlEnv"),
             "OctoArmTwo-v0": ("gym_softrobot.envs.octopus.arm_two_env", "ArmTwoEnv"),
             "OctoReach-v0": ("gym_softrobot.envs.octopus.reach_env", "ReachEnv")}
    KWARGS = {"OctoArmPush-v1": dict(mode="continuous"), "OctoArmPullWeight-v0": dict(mode="continuous")}

    def __init__(self, env_id: str, reference: str = "/root/reference"):
        if reference not in sys.path:
            sys.path.insert(0, reference)
        import elastica  # noqa: F401  (fails here -> this driver cannot be used in this container)

        self.env_id = env_id
        self.env = self._make(env_id)
        self.octo = env_id == "OctoFlat-v0" or env_id in MUSCLE_OCTOPUS_IDS
        self.pull = env_id == "OctoArmPullWeight-v0"
        self.adim = max(1, int(np.prod(self.env.action_space.shape)))      # Discrete(2): shape () -> one number
        self.discrete = type(self.env.action_space).__name__ == "Discrete"

    def _make(self, env_id):
        """`gym.make(id).unwrapped` as a user of the reference gets it; where gymnasium's registry is not
        available (tests/test_pyelastica_fixtures.py runs this driver over tools/refshim.py's stand-ins)
        the registered entry point itself — the same class, gymnasium's wrappers do not touch the physics."""
        import importlib

        try:
            import gymnasium as gym

            import gym_softrobot  # noqa: F401  (registers the env ids, gym_softrobot/__init__.py)

            return gym.make(env_id).unwrapped
        except (AttributeError, ImportError):
            module, cls = self.ENTRY[env_id]
            return getattr(importlib.import_module(module), cls)(**self.KWARGS.get(env_id, {}))

    def _post_reset(self) -> None:
        pass

    def _obs(self, obs):
        if isinstance(obs, dict):                       # FlatEnv: {"individual": (n_arm, w), "shared": (13,)}
            return np.concatenate([np.asarray(obs["individual"], np.float32).ravel(),
                                   np.asarray(obs["shared"], np.float32).ravel()])
        return np.asarray(obs, np.float32).copy()

    def reset(self, seed: int):
        obs, _ = self.env.reset(seed=int(seed))
        self._post_reset()
        return self._obs(obs)

    def step(self, action):
        a = int(np.asarray(action).ravel()[0]) if self.discrete else np.asarray(action, np.float32).reshape(self.env.action_space.shape)
        o, r, te, tr, info = self.env.step(a)
        return self._obs(o), float(r), bool(te), bool(tr), float(info["time"])

    def substeps(self, n: int) -> None:
        e = self.env
        e.set_action(np.zeros(e.action_space.shape, np.float32))
        for _ in range(int(n)):
            e.time = e.do_step(e.simulator, e.time, e.time_step)

    def state(self) -> Dict[str, np.ndarray]:
        e = self.env

        def rod(r):
            return (np.array(r.position_collection), np.array(r.velocity_collection),
                    np.array(r.director_collection), np.array(r.omega_collection))

        if self.octo:
            parts = [rod(r) for r in e.shearable_rods]
            hx, hv, hq, hw = rod(e.rigid_rod)
            return {"x": np.stack([p[0] for p in parts]), "v": np.stack([p[1] for p in parts]),
                    "Q": np.stack([p[2] for p in parts]), "w": np.stack([p[3] for p in parts]),
                    "head_x": hx[:, 0], "head_v": hv[:, 0], "head_Q": hq[:, :, 0], "head_w": hw[:, 0],
                    "time": np.float64(e.time)}
        x, v, q, w = rod(e.shearable_rod)
        out = {"x": x, "v": v, "Q": q, "w": w, "time": np.float64(e.time)}
        if self.pull:
            hx, hv, hq, hw = rod(e.rigid_rod)
            out.update(head_x=hx[:, 0], head_v=hv[:, 0], head_Q=hq[:, :, 0], head_w=hw[:, 0])
        return out

    def close(self) -> None:
        self.env.close()


# ---------------------------------------------------------------------------------------------
# record / replay
# ---------------------------------------------------------------------------------------------
def record_case(driver, seed: int, n_steps: Optional[int] = None) -> Dict[str, np.ndarray]:
    """Everything one fixture file holds, produced by `driver`."""
    spec = ENVS_ALL[driver.env_id]
    T = int(n_steps or spec["n_steps"])
    out: Dict[str, np.ndarray] = {"env_id": np.array(driver.env_id), "seed": np.int64(seed),
                                  "source": np.array(driver.kind)}
    out["reset_obs"] = driver.reset(seed)
    for k, v in driver.state().items():
        out[f"reset_{k}"] = v
    done = 0
    for n in (RAW_SUBSTEPS if spec.get("raw", True) else ()):
        driver.substeps(n - done)
        done = n
        for k, v in driver.state().items():
            out[f"sub{n}_{k}"] = v
    driver.reset(seed)                                 # same seed, same first draw: the same start
    acts = action_script(driver.env_id, seed, driver.adim)[:T]
    out["actions"] = acts
    obs, rew, term, trunc, tim = [], [], [], [], []
    for t in range(T):
        o, r, te, tr, tm = driver.step(acts[t])
        obs.append(o), rew.append(r), term.append(te), trunc.append(tr), tim.append(tm)
        if (t + 1) in spec["state_steps"]:
            for k, v in driver.state().items():
                out[f"step{t + 1}_{k}"] = v
        if te:                                         # the reference stops integrating sensibly after a NaN / goal
            break
    out["obs"] = np.stack(obs)
    out["reward"] = np.asarray(rew, np.float64)
    out["terminated"] = np.asarray(term, bool)
    out["truncated"] = np.asarray(trunc, bool)
    out["time"] = np.asarray(tim, np.float64)
    return out


def compare_case(driver, fx, max_step: Optional[int] = None, raw: bool = True) -> Dict[str, float]:
    """Replay fixture `fx` (a dict / NpzFile) through `driver`; -> {record name: deviation}."""
    seed = int(fx["seed"])
    dev: Dict[str, float] = {}
    dev["reset_obs"] = deviation(driver.reset(seed), fx["reset_obs"], FLOOR["obs"])
    if raw:
        done = 0
        for n in RAW_SUBSTEPS:
            if f"sub{n}_x" not in fx:
                continue
            driver.substeps(n - done)
            done = n
            st = driver.state()
            for k in ("x", "v", "Q", "w", "head_x", "head_v", "head_Q", "head_w"):
                if f"sub{n}_{k}" in fx:
                    dev[f"sub{n}_{k}"] = deviation(st[k], fx[f"sub{n}_{k}"], floor_of(k))
        driver.reset(seed)
    acts = np.asarray(fx["actions"])
    T = len(fx["obs"]) if max_step is None else min(int(max_step), len(fx["obs"]))
    for t in range(T):
        o, r, te, tr, tm = driver.step(acts[t])
        dev[f"step{t + 1}_obs"] = deviation(o, fx["obs"][t], FLOOR["obs"])
        dev[f"step{t + 1}_reward"] = deviation(r, fx["reward"][t], FLOOR["reward"])
        dev[f"step{t + 1}_flags"] = 0.0 if (te == bool(fx["terminated"][t]) and tr == bool(fx["truncated"][t])) else float("inf")
        dev[f"step{t + 1}_time"] = 0.0 if tm == float(fx["time"][t]) else abs(tm - float(fx["time"][t])) / float(fx["time"][t])
        if f"step{t + 1}_x" in fx:
            st = driver.state()
            for k in ("x", "v", "Q", "w", "head_x", "head_v", "head_Q", "head_w"):
                if f"step{t + 1}_{k}" in fx:
                    dev[f"step{t + 1}_{k}"] = deviation(st[k], fx[f"step{t + 1}_{k}"], floor_of(k))
        if te:
            break
    return dev


def worst(dev: Dict[str, float], upto_step: Optional[int] = None, skip_time: bool = False) -> float:
    """Largest deviation over the records of the raw substeps and of env.steps <= upto_step."""
    w = 0.0
    for k, v in dev.items():
        if skip_time and k.endswith("_time"):
            continue
        if k.startswith("step") and upto_step is not None:
            if int(k[4:].split("_")[0]) > upto_step:
                continue
        w = max(w, v)
    return w


def strict_worst(dev: Dict[str, float], env_id: str, limit: Optional[int] = None) -> float:
    """The figure held against 1e-5: raw substeps and env.steps <= the env's strict horizon (or
    `limit`, if lower); the clock records of ALL steps are exact or not, and are included."""
    n = ENVS_ALL[env_id]["strict_steps"] if limit is None else min(limit, ENVS_ALL[env_id]["strict_steps"])
    return worst(dev, n)


def horizon(dev: Dict[str, float], tol: float = 1e-5) -> int:
    """Number of leading env.steps whose obs / reward / state records all stay within `tol`."""
    t = 0
    while any(k.startswith(f"step{t + 1}_") for k in dev):
        if any(v > tol for k, v in dev.items() if k.startswith(f"step{t + 1}_")):
            break
        t += 1
    return t


def load_switches(directory, prefix: str = "pyelastica") -> Dict[str, object]:
    """What tools/sweep_switches.py --write decided for these fixtures, else the shipped defaults."""
    import json

    f = Path(directory) / f"{prefix}_switches.json"
    return dict(default_switches(), **(json.loads(f.read_text())["switches"] if f.exists() else {}))


def fixture_files(directory, prefix: str = "pyelastica") -> List[Path]:
    return sorted(Path(directory).glob(f"{prefix}_*_seed*.npz"))


def fixture_name(env_id: str, seed: int, prefix: str = "pyelastica") -> str:
    return f"{prefix}_{env_id}_seed{seed}.npz"
