"""OctoArmPush-v0 / OctoArmPush-v1 on the MI355X batched Cosserat-rod stepper — PARITY UNPINNED.

Mirrors gym_softrobot/envs/octopus/arm_push_env.py:52-347 (`ArmPushEnv`): a 40-element arm tapered
12:1 (:160-179), `AnalyticalLinearDamper` (:180-185), one `ControllableFixConstraint` "sucker" whose
index the action moves (:187-195, :257,262,270) and COOMM's `ApplyMuscles` over the three layers of
`create_es_muscle_layers` (:197-212; octopus/build.py:295-338); no gravity, no plane.  "discrete" mode
(v0): action 0 holds the base and drives the transverse muscle at 0.5, action 1 holds the tip and
relaxes (:254-267) — an inchworm; "continuous" mode (v1): (sucker location, transverse activation)
(:268-271).  Observation: x-positions, x-velocities, the previous action (one-hot in v0) (:225-245);
reward: the change of the centre of mass's distance from the origin, -20 on NaN (:288-333).

WHAT IS AND IS NOT PINNED.  Everything in arm_push_env.py (geometry, wiring, set_action, get_state,
reward) is pinned against the EXECUTED reference file (tools/make_env_golden.py,
tests/golden/ref_arm_push.npz).  The muscle force model lives in COOMM (git pin uv.lock:173-175), which
is not on disk: the kernels restate the published law (include/softrod.h, SOFTROD_FEAT_COOMM_MUSCLES)
with every recalled detail a softrod_config switch.  `spec_label` / the registry entry say
"parity-unpinned (COOMM)" and bench.py repeats it.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from .. import _capi
from ..spaces import Box, Discrete
from .base import GymEnv as _GymEnv
from .base import VecRodEnvBase

PARITY_LABEL = "parity-unpinned (COOMM muscle law restated from the published model; not on disk)"


class VecArmPushEnv(VecRodEnvBase):
    """N parallel OctoArmPush envs resident on one GPU (see VecRodEnvBase).  Discrete mode takes
    actions of shape (N,) or (N, 1) holding 0 / 1."""

    metadata = {"render_modes": ["rgb_array", "human"], "render_fps": 40}
    action_low, action_high = 0.0, 1.0                 # arm_push_env.py:101,113-115
    parity_label = PARITY_LABEL

    def __init__(
        self,
        num_envs: int,
        final_time: float = 2.5,
        time_step: float = 5.0e-5,
        recording_fps: int = 40,
        mode: str = "discrete",
        config_generate_video: bool = False,
        config_early_termination: bool = False,
        render_mode: Optional[str] = None,
        *,
        device: int = 0,
        math_mode: int = _capi.MATH_FAST,
        numpy_output: bool = False,
        autoreset: bool = False,
        backend=None,
        muscle_kwargs: Optional[dict] = None,
    ):
        """`muscle_kwargs`: the recalled COOMM constructor behaviour of `_capi.es_muscle_layers`
        (init_angle_rotates, tm_sign) — switches for the day the muscle fixtures exist."""
        if config_early_termination:
            raise NotImplementedError("config_early_termination (the Hamiltonian cut-off of arm_push_env.py:311-314, "
                                      "389-404) is not built: the default (False) only")
        cfg = self._config(num_envs, final_time=final_time, time_step=time_step,
                           recording_fps=recording_fps, mode=mode, math_mode=math_mode)
        super().__init__(num_envs, cfg, render_mode=render_mode, config_generate_video=config_generate_video,
                         device=device, numpy_output=numpy_output, autoreset=autoreset, backend=backend)
        self.final_time = final_time
        self.time_step = time_step
        self.total_steps = int(self.final_time / self.time_step)
        self.recording_fps = recording_fps
        self.step_skip = int(1.0 / (recording_fps * time_step))
        self.n_elem = 40                                # :88
        self.mode = int(cfg.arm_push_mode)
        self.config_early_termination = False
        if self.mode == 0:
            self.single_action_space = Discrete(2)      # :101
        radius_mean = _capi.arm_push_radii(self.n_elem)
        self.backend.set_radius_profile(radius_mean)
        ratio, strength = _capi.es_muscle_layers(radius_mean, 0.012, **(muscle_kwargs or {}))
        self.backend.set_muscle_layers(ratio, strength)

    @staticmethod
    def _config(num_envs, **kw):
        return _capi.arm_push_config(num_envs, **kw)

    def _validate_actions(self, actions) -> None:
        if self.mode == 0:
            if hasattr(actions, "device") and getattr(actions.device, "type", "cpu") != "cpu":
                return                                  # resident actions are not read back (no host sync in the loop)
            a = np.asarray(actions)
            if not np.isin(a, (0, 1)).all():
                raise NotImplementedError("Action must be 1 or 0")     # arm_push_env.py:267

    def _draw_reset(self, i):
        return None                                     # _build draws nothing from the RNG

    def _frames(self, shape):
        start = np.zeros(shape + (3,))
        direction = np.broadcast_to(np.array([1.0, 0.0, 0.0]), shape + (3,)).copy()    # :170
        normal = np.broadcast_to(np.array([0.0, 1.0, -0.0]), shape + (3,)).copy()      # :171
        return start, direction, normal

    def _queue_from_draws(self, draws, counts):
        self.backend.queue_push_straight(*self._frames((self.num_envs, max(1, int(counts.max())))), counts)

    def _reset_backend(self, mask, use_mask, draws=None):
        self.backend.reset_straight(*self._frames((self.num_envs,)), mask.astype(np.uint8) if use_mask else None)


class VecArmPullWeightEnv(VecArmPushEnv):
    """N parallel OctoArmPullWeight-v0 envs: `ArmPullWeightEnv(ArmPushEnv)` (octopus/arm_push_env.py:516-618) — the same
    arm and step(), time_step 2.5e-5, joined at node 0 to a rigid Cylinder "weight" (FixedJoint2Rigid, held upright by
    BodyBoundaryCondition), the sucker at reduction_ratio 0.9.  PARITY UNPINNED like OctoArmPush."""

    def __init__(self, num_envs: int, **kwargs):
        if "time_step" in kwargs:        # `super().__init__(time_step=2.5e-5, **kwargs)` (:518) would raise the same
            raise TypeError("__init__() got multiple values for keyword argument 'time_step'")
        super().__init__(num_envs, time_step=2.5e-5, **kwargs)

    @staticmethod
    def _config(num_envs, *, time_step, **kw):
        assert time_step == 2.5e-5
        return _capi.arm_pull_weight_config(num_envs, **kw)


class ArmPushEnv(_GymEnv):
    """Drop-in for gym_softrobot's ArmPushEnv (octopus/arm_push_env.py:52-347), N = 1.  PARITY UNPINNED
    (module docstring)."""

    metadata = {"render_modes": ["rgb_array", "human"], "render_fps": 40}
    parity_label = PARITY_LABEL

    def __init__(
        self,
        final_time: float = 2.5,
        time_step: float = 5.0e-5,
        recording_fps: int = 40,
        mode: str = "discrete",
        config_generate_video: bool = False,
        config_early_termination: bool = False,
        render_mode: Optional[str] = None,
        *,
        device: int = 0,
        math_mode: int = _capi.MATH_FAST,
        backend=None,
    ):
        super().__init__()
        if render_mode not in {None, *self.metadata["render_modes"]}:
            raise ValueError(f"Unsupported render mode: {render_mode}")
        self.render_mode = render_mode
        self._vec = self._make_vec(final_time, time_step, recording_fps, mode, config_generate_video,
                                   config_early_termination, device, math_mode, backend)
        self.final_time = final_time
        self.time_step = time_step
        self.total_steps = self._vec.total_steps
        self.recording_fps = recording_fps
        self.step_skip = self._vec.step_skip
        self.n_elem = 40
        self.mode = self._vec.mode
        if self.mode == 0:
            self.action_space = Discrete(2)
            self._observation_size = ((self.n_elem + 1) * 2 + 2,)
        else:
            self.action_space = Box(0.0, 1.0, shape=(2,), dtype=np.float32)
            self._observation_size = ((self.n_elem + 1) * 2 + 2,)
        self.observation_space = Box(-np.inf, np.inf, shape=self._observation_size, dtype=np.float32)
        self._prev_action = np.zeros(list(self.action_space.shape), dtype=self.action_space.dtype)
        self.config_generate_video = config_generate_video
        self.config_early_termination = config_early_termination
        self.time = np.float64(0.0)

    @staticmethod
    def _make_vec(final_time, time_step, recording_fps, mode, config_generate_video, config_early_termination,
                  device, math_mode, backend):
        return VecArmPushEnv(1, final_time, time_step, recording_fps, mode, config_generate_video,
                             config_early_termination, None, device=device, math_mode=math_mode,
                             numpy_output=True, backend=backend)

    def reset(self, *, seed: Optional[int] = None, options: Optional[dict] = None):
        super().reset(seed=seed)
        obs, _ = self._vec.reset()
        self.time = np.float64(0.0)
        return np.asarray(obs[0], dtype=np.float32).copy(), {}

    def get_state(self):
        obs = self._vec.backend.observe(None)
        return np.asarray(obs[0].cpu().numpy() if hasattr(obs, "cpu") else obs[0], dtype=np.float32).copy()

    def step(self, action):
        if self.mode == 0:
            if action not in (0, 1):
                raise NotImplementedError("Action must be 1 or 0")     # arm_push_env.py:267
            a = np.array([[float(action)]], np.float32)
        else:
            a = np.asarray(action, dtype=np.float32).reshape(1, 2)
        obs, reward, term, trunc, infos = self._vec.step(a)
        self._prev_action = action
        self.time = np.float64(infos["time"][0])
        return (
            np.asarray(obs[0], dtype=np.float32).copy(),
            float(reward[0]),
            bool(term[0]),
            bool(trunc[0]),
            {"time": self.time, "TimeLimit.truncated": bool(infos["TimeLimit.truncated"][0])},
        )

    def render(self):
        from ..render import render_env

        return render_env(self)

    def close(self):
        from ..render import close_env

        close_env(self)
        self._vec.close()


class ArmPullWeightEnv(ArmPushEnv):
    """Drop-in for gym_softrobot's ArmPullWeightEnv (octopus/arm_push_env.py:516-618), N = 1.  PARITY UNPINNED."""

    def __init__(self, **kwargs):
        if "time_step" in kwargs:
            raise TypeError("__init__() got multiple values for keyword argument 'time_step'")
        super().__init__(time_step=2.5e-5, **kwargs)        # :518

    @staticmethod
    def _make_vec(final_time, time_step, recording_fps, mode, config_generate_video, config_early_termination,
                  device, math_mode, backend):
        return VecArmPullWeightEnv(1, final_time=final_time, recording_fps=recording_fps, mode=mode,
                                   config_generate_video=config_generate_video,
                                   config_early_termination=config_early_termination, render_mode=None, device=device,
                                   math_mode=math_mode, numpy_output=True, backend=backend)
