"""SoftPendulum3D-v0 on the MI355X batched Cosserat-rod stepper.

Mirrors gym_softrobot/envs/soft_pendulum_3d/soft_pendulum_3d.py:20-174 and
soft_pendulum_3d/build.py:15-86: a vertical rod (tilted by up to +-1 degree) whose base
is moved in x-y by a 2-D action; MovingBaseConstraint, GravityForces,
AnalyticalLinearDamper(1.0) and LaplaceDissipationFilter(7) are compiled-in features of
the kernel (SOFTROD_FEATURES_SOFTPENDULUM3D).
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from .. import _capi
from ..spaces import Box
from .base import GymEnv as _GymEnv
from .base import VecRodEnvBase


def initial_tilt(rng: np.random.Generator) -> float:
    """soft_pendulum_3d/build.py:51: one `uniform(-1, 1)` draw per reset."""
    return float(np.deg2rad(rng.uniform(-1.0, 1.0)))


class VecSoftPendulum3DEnv(VecRodEnvBase):
    """N parallel SoftPendulum3D-v0 envs resident on one GPU (see VecRodEnvBase)."""

    metadata = {"render_modes": ["rgb_array"], "render_fps": 25}
    action_low, action_high = -1.0, 1.0               # soft_pendulum_3d.py:41-46

    def __init__(
        self,
        num_envs: int,
        final_time: float = 5.0,
        time_step: float = 1.0e-4,
        recording_fps: int = 25,
        n_elems: int = 50,
        config_generate_video: bool = False,
        render_mode: Optional[str] = None,
        *,
        device: int = 0,
        math_mode: int = _capi.MATH_FAST,
        numpy_output: bool = False,
        autoreset: bool = False,
        backend=None,
    ):
        cfg = _capi.softpendulum3d_config(
            num_envs, final_time=final_time, time_step=time_step,
            recording_fps=recording_fps, n_elems=n_elems, math_mode=math_mode,
        )
        super().__init__(num_envs, cfg, render_mode=render_mode,
                         config_generate_video=config_generate_video, device=device,
                         numpy_output=numpy_output, autoreset=autoreset, backend=backend)
        self.final_time = final_time
        self.time_step = time_step
        self.total_steps = int(self.final_time / self.time_step)
        self.recording_fps = recording_fps
        self.step_skip = int(1.0 / (recording_fps * time_step))
        self.n_elems = n_elems
        self.base_step = 1e-3
        self.base_limit = 0.5

    def _draw_reset(self, i):
        tilt = initial_tilt(self._rngs[i])
        return [np.sin(tilt), 0.0, np.cos(tilt)]              # soft_pendulum_3d/build.py:52

    def _queue_from_draws(self, draws, counts):
        n, m = self.num_envs, max(1, int(counts.max()))
        direction = np.zeros((n, m, 3))
        direction[:, :, 2] = 1.0
        for i, d in enumerate(draws):
            for j, v in enumerate(d):
                direction[i, j] = v
        start = np.zeros((n, m, 3))
        normal = np.tile(np.array([0.0, 1.0, 0.0]), (n, m, 1))
        self.backend.queue_push_straight(start, direction, normal, counts)

    def _reset_backend(self, mask, use_mask, draws=None):
        n = self.num_envs
        direction = np.zeros((n, 3))
        direction[:, 2] = 1.0
        for i in np.nonzero(mask)[0]:
            direction[i] = self._draw(i, draws)
        start = np.zeros((n, 3))
        normal = np.tile(np.array([0.0, 1.0, 0.0]), (n, 1))   # :53
        self.backend.reset_straight(start, direction, normal, mask.astype(np.uint8) if use_mask else None)

    def _validate_actions(self, actions):
        import torch

        if isinstance(actions, np.ndarray) or not torch.is_tensor(actions):
            a_np = np.asarray(actions, dtype=np.float32).reshape(self.num_envs, 2)
            if not (np.all(a_np >= -1.0) and np.all(a_np <= 1.0)):
                # soft_pendulum_3d.py:116-117 (device tensors are the caller's responsibility)
                raise ValueError(f"Action {actions!r} is outside {self.single_action_space}")

    def _infos(self, times):
        return {"time": times, "tilt": self._out(self.backend.aux[:, 0])}


class SoftPendulum3DEnv(_GymEnv):
    """Drop-in for gym_softrobot's SoftPendulum3DEnv (soft_pendulum_3d.py:20-174), N = 1."""

    metadata = {"render_modes": ["rgb_array"], "render_fps": 25}

    def __init__(
        self,
        final_time: float = 5.0,
        time_step: float = 1.0e-4,
        recording_fps: int = 25,
        n_elems: int = 50,
        config_generate_video: bool = False,
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
        self._vec = VecSoftPendulum3DEnv(
            1, final_time, time_step, recording_fps, n_elems, config_generate_video, None,
            device=device, math_mode=math_mode, numpy_output=True, backend=backend,
        )
        self.final_time = final_time
        self.time_step = time_step
        self.total_steps = self._vec.total_steps
        self.recording_fps = recording_fps
        self.step_skip = self._vec.step_skip
        self.n_elems = n_elems
        self.n_action = 2
        self.action_space = Box(-1.0, 1.0, shape=(2,), dtype=np.float32)
        self.observation_space = Box(-np.inf, np.inf, shape=(9,), dtype=np.float32)
        self.base_step = 1e-3
        self.base_limit = 0.5
        self.time = np.float64(0.0)
        self.counter = 0

    def reset(self, *, seed: Optional[int] = None, options: Optional[dict] = None):
        super().reset(seed=seed)
        self._vec._rngs[0] = self.np_random
        obs, _ = self._vec.reset(seed=None)
        self.time = np.float64(0.0)
        self.counter = 0
        return np.asarray(obs[0], dtype=np.float32).copy(), {}

    def step(self, action):
        if not self.action_space.contains(action):  # soft_pendulum_3d.py:116-117
            raise ValueError(f"Action {action!r} is outside {self.action_space}")
        a = np.asarray(action, dtype=np.float32).reshape(1, 2)
        obs, reward, term, trunc, infos = self._vec.step(a)
        self.time = np.float64(infos["time"][0])
        self.counter += 1
        return (
            np.asarray(obs[0], dtype=np.float32).copy(),
            float(reward[0]),
            bool(term[0]),
            bool(trunc[0]),
            {"time": self.time, "tilt": float(infos["tilt"][0])},
        )

    def get_state(self):
        """Current observation (soft_pendulum_3d.py:93-98)."""
        obs = self._vec.backend.observe(None)
        return np.asarray(obs[0].cpu().numpy() if hasattr(obs, "cpu") else obs[0], dtype=np.float32).copy()

    def render(self):
        """None without a render mode; an (H, W, 3) uint8 frame for "rgb_array" (render.py)."""
        from ..render import render_env

        return render_env(self)

    def close(self):
        from ..render import close_env

        close_env(self)
        self._vec.close()
