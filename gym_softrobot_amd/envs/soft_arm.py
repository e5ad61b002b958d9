"""SoftArmTracking-v0 on the MI355X batched Cosserat-rod stepper.

Mirrors gym_softrobot/envs/soft_arm/soft_arm_tracking.py:104-483 (game_mode 1, the registered
SoftArmTracking-v0): a 40-element arm clamped at its base (OneEndFixedBC), damped
(AnalyticalLinearDamper), bent by two MuscleTorquesWithVaryingBetaSplines (normal and
binormal, 4 control points each = the 8-dim action), reaching for a fixed target with its tip;
50 PositionVerlet substeps per env.step, 500 steps per episode.  The target Sphere is appended
to the reference's simulator without any connection and its state is overwritten every substep
(:223-224, :428-436), so it does not act on the rod: here it is just the three numbers that
enter the reward and the observation.

The reference's observation and action spaces are float64; the C-ABI hands observations over
as float32 (include/softrod.h), which this wrapper widens again — values agree to float32
rounding (6e-8 relative, against the 1e-5 the parity tests allow).
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from .. import _capi
from ..spaces import Box
from .base import GymEnv as _GymEnv
from .base import VecRodEnvBase


def target_trajectory(final_time: float, sim_dt: float, target_v_scale: float, rng, every: int = 1):
    """generate_trajectory (soft_arm_tracking.py:46-101) at every `every`-th sample: a product of
    three sines per axis with frequencies, sign and a start time drawn from the env's RNG, in
    millimetres, inside [-800, 800] x [-400 + 0.4, 400 + 0.4] x [-800, 800] (the reference adds
    its 0.4 offset to y AFTER the scaling to millimetres, :85-86; kept).  Draw order as there:
    the start time, then per axis three frequencies and the sign."""
    end_time = final_time * 1.1
    numpoints = int(np.rint(1 / sim_dt * end_time))
    idx = np.arange(0, numpoints, every, dtype=np.float64)
    t = end_time * idx / (numpoints - 1)
    t += rng.random() * 3600
    out = np.zeros((len(idx), 3))
    for axis, amp in ((0, 0.8), (1, 0.4), (2, 0.8)):
        f1 = rng.uniform(2, 5) * 0.025 * target_v_scale
        f2 = rng.uniform(2, 5) * 0.025 * target_v_scale
        f3 = rng.uniform(2, 5) * 0.025 * target_v_scale
        direction = rng.integers(0, 2) * 2 - 1
        out[:, axis] = (direction * amp * np.sin(2 * np.pi * f1 * t) * np.sin(2 * np.pi * f2 * t)
                        * np.sin(2 * np.pi * f3 * t) * 1000)
        if axis == 1:
            out[:, 1] += 0.4
    return out


class VecSoftArmTrackingEnv(VecRodEnvBase):
    """N parallel SoftArmTracking-v0 envs resident on one GPU (see VecRodEnvBase)."""

    metadata = {"render_modes": ["rgb_array", "human"], "render_fps": 30}
    action_low, action_high = -1.0, 1.0               # soft_arm_tracking.py:152-157

    def __init__(self, num_envs: int, game_mode: int = 1, render_mode: Optional[str] = None, *,
                 n_elems: int = 40, device: int = 0, math_mode: int = _capi.MATH_FAST,
                 numpy_output: bool = False, autoreset: bool = False, backend=None):
        if game_mode not in (1, 2):
            raise ValueError("game_mode is 1 (fixed target) or 2 (moving target), soft_arm_tracking.py:386-412")
        if game_mode == 2 and autoreset == "device":
            raise NotImplementedError("game_mode 2 draws a trajectory per episode on the host: use autoreset=True")
        cfg = _capi.soft_arm_config(num_envs, n_elems=n_elems, math_mode=math_mode)
        super().__init__(num_envs, cfg, render_mode=render_mode, config_generate_video=False, device=device,
                         numpy_output=numpy_output, autoreset=autoreset, backend=backend)
        self.n_elem = n_elems
        self.sim_dt = float(cfg.dt)
        self.num_steps_per_update = int(cfg.n_substeps)
        self.max_episode_final_time = float(cfg.final_time)
        self.mode = game_mode
        self.target_v_scale = 0.1                     # :144
        self._traj = None                             # game_mode 2: wsol at the env.step boundaries, (N, K, 3)

    def _draw_reset(self, i):
        if self.mode == 1:
            return None                               # game_mode 1 draws nothing from the RNG
        return target_trajectory(self.max_episode_final_time, self.sim_dt, self.target_v_scale, self._rngs[i],
                                 every=self.num_steps_per_update)

    def _queue_from_draws(self, draws, counts):
        n, m = self.num_envs, max(1, int(counts.max()))
        start = np.zeros((n, m, 3))
        direction = np.tile(np.array([0.0, 1.0, 0.0]), (n, m, 1))
        normal = np.tile(np.array([0.0, 0.0, 1.0]), (n, m, 1))
        self.backend.queue_push_straight(start, direction, normal, counts)

    def _reset_backend(self, mask, use_mask, draws=None):
        n = self.num_envs
        start = np.zeros((n, 3))
        direction = np.tile(np.array([0.0, 1.0, 0.0]), (n, 1))   # rod pointing upwards, :266-268
        normal = np.tile(np.array([0.0, 0.0, 1.0]), (n, 1))
        self.backend.reset_straight(start, direction, normal, mask.astype(np.uint8) if use_mask else None)
        if self.mode == 2:
            import torch

            if self._traj is None:
                self._traj = np.zeros((n, int(np.rint(self.max_episode_final_time * 1.1 / self.sim_dt))
                                       // self.num_steps_per_update + 1, 3))
            for i in np.nonzero(mask)[0]:
                tr = self._draw(int(i), draws)
                self._traj[i, : len(tr)] = tr
            # wsol[0] is what the reset observation shows (:476-477)
            ctrl = self.backend.state()["control"]
            tgt = torch.from_numpy(np.ascontiguousarray(self._traj[:, 0, :].T)).to(ctrl.device)
            m = torch.from_numpy(np.asarray(mask, bool)).to(ctrl.device)
            ctrl[1:4] = torch.where(m[None, :], tgt, ctrl[1:4])

    def step(self, actions):
        if self.mode == 2:
            # the sphere follows wsol[tick] (:223): after this env.step tick = 50 (steps + 1)
            import torch

            k = np.minimum(self._steps + 1, self._traj.shape[1] - 1)
            tgt = self._traj[np.arange(self.num_envs), k, :]
            ctrl = self.backend.state()["control"]
            ctrl[1:4] = torch.from_numpy(np.ascontiguousarray(tgt.T)).to(ctrl.device)
        return super().step(actions)

    def _infos(self, times):
        # truncation is `tick * sim_dt >= max_episode_final_time` (:253): an integer count, not
        # the accumulated float time
        ticks = self._steps * self.num_steps_per_update
        return {"time": times, "ctime": times, "TimeLimit.truncated": ticks * self.sim_dt >= self.max_episode_final_time}


class SoftArmTrackingEnv(_GymEnv):
    """Drop-in for gym_softrobot's SoftArmTrackingEnv (soft_arm/soft_arm_tracking.py:104-548), N = 1."""

    metadata = {"render_modes": ["rgb_array", "human"], "render_fps": 30}

    def __init__(self, game_mode: int = 1, render_mode: Optional[str] = None, *, device: int = 0,
                 math_mode: int = _capi.MATH_FAST, backend=None):
        super().__init__()
        if render_mode not in {None, *self.metadata["render_modes"]}:
            raise ValueError(f"Unsupported render mode: {render_mode}")
        self.render_mode = render_mode
        self._vec = VecSoftArmTrackingEnv(1, game_mode, None, device=device, math_mode=math_mode,
                                          numpy_output=True, backend=backend)
        self.n_elem = 40
        self.sim_dt = 2.0e-4
        self.RL_update_interval = 0.01
        self.num_steps_per_update = self._vec.num_steps_per_update
        self.youngs_modulus = 2e6
        self.torque_scale = 10
        self.max_episode_final_time = 5
        self.base_length = 1000
        self.radius = 50
        self.number_of_control_points = 4
        self.number_of_observation_segments = 4
        self.mode = game_mode
        self.target_location = np.array([500, 500.0, 500])
        self.target_v_scale = 0.1
        self.action_space = Box(-1.0, 1.0, shape=(8,), dtype=np.float64)
        self.observation_space = Box(-np.inf, np.inf, shape=(14,), dtype=np.float64)
        self.tick = 0
        self.time_tracker = np.float64(0.0)

    def reset(self, *, seed: Optional[int] = None, options: Optional[dict] = None):
        super().reset(seed=seed)
        self._vec._rngs[0] = self.np_random       # env-owned stream, as soft_arm_tracking.py:402-407
        obs, _ = self._vec.reset(seed=None)
        self.tick = 0
        self.time_tracker = np.float64(0.0)
        return np.asarray(obs[0], dtype=np.float64), {}

    def step(self, action):
        a = np.asarray(action, dtype=np.float32).reshape(1, 8)
        obs, reward, term, trunc, infos = self._vec.step(a)
        self.tick += self.num_steps_per_update
        self.time_tracker = np.float64(infos["time"][0])
        return (np.asarray(obs[0], dtype=np.float64), float(reward[0]), bool(term[0]), bool(trunc[0]),
                {"ctime": self.time_tracker})

    def get_state(self):
        """Current observation (soft_arm_tracking.py:160-207)."""
        obs = self._vec.backend.observe(None)
        return np.asarray(obs[0].cpu().numpy() if hasattr(obs, "cpu") else obs[0], dtype=np.float64)

    def render(self):
        """None without a render mode; an (H, W, 3) uint8 frame for "rgb_array" (render.py)."""
        from ..render import render_env

        return render_env(self)

    def close(self):
        from ..render import close_env

        close_env(self)
        self._vec.close()
