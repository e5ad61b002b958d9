"""SoftPendulum-v0 on the MI355X batched Cosserat-rod stepper.

Mirrors the reference's env surface for this path
(gym_softrobot/envs/soft_pendulum/soft_pendulum.py:45-322):

* `SoftPendulumEnv`     — the single-env Gymnasium API (`reset(seed=, options=)`,
                          `step(action)`), same constructor keywords, spaces, obs/reward/
                          flag/info types; a batch of one rod on the GPU.
* `VecSoftPendulumEnv`  — N parallel envs (Gymnasium VectorEnv-shaped): the form the hot
                          path is built for.  Env i is seeded `seed + i`.

The per-substep Python hooks of the reference (PendulumBoundaryConditions,
PendulumPointForces; build.py:65-79,94-101) are compiled-in features of the kernel,
selected by `softrod_config.features`.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from .. import _capi
from ..seeding import initial_angle
from ..spaces import Box
from .base import GymEnv as _GymEnv
from .base import VecRodEnvBase
from .base import time_table as _time_table  # noqa: F401  (re-exported for tests)


class VecSoftPendulumEnv(VecRodEnvBase):
    """N parallel SoftPendulum-v0 envs resident on one GPU (see VecRodEnvBase).
    No auto-reset unless `autoreset=True` (the reference has none)."""

    metadata = {"render_modes": ["rgb_array"], "render_fps": 25}
    action_low, action_high = -22.0, 22.0            # soft_pendulum.py:84-90

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
        cfg = _capi.softpendulum_config(
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
        self.n_seg = n_elems - 1

    def _draw_reset(self, i):
        return initial_angle(self._rngs[i])           # build.py:47-49

    def _reset_backend(self, mask, use_mask, draws=None):
        theta0 = np.zeros(self.num_envs, np.float64)
        for i in np.nonzero(mask)[0]:
            theta0[i] = self._draw(i, draws)
        self.backend.reset(theta0, mask.astype(np.uint8) if use_mask else None)

    def _queue_from_draws(self, draws, counts):
        th = np.zeros((self.num_envs, max(1, int(counts.max()))))
        for i in np.nonzero(counts > 0)[0]:
            th[i, : counts[i]] = draws[i]
        self.backend.queue_push(th, counts)


class SoftPendulumEnv(_GymEnv):
    """Drop-in for gym_softrobot's SoftPendulumEnv (soft_pendulum.py:45-322), N = 1.

    Same constructor keywords (soft_pendulum.py:59-67), spaces (:84-94), return
    types (:241-251).  `render()` and video generation are outside the hot path.
    """

    metadata = {"render_modes": ["rgb_array"], "render_fps": 25}

    def __init__(
        self,
        final_time=5.0,
        time_step=1.0e-4,
        recording_fps=25,
        n_elems=50,
        config_generate_video=False,
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
        self._vec = VecSoftPendulumEnv(
            1, final_time, time_step, recording_fps, n_elems, config_generate_video,
            None, device=device, math_mode=math_mode, numpy_output=True, backend=backend,
        )
        self.final_time = final_time
        self.time_step = time_step
        self.total_steps = self._vec.total_steps
        self.recording_fps = recording_fps
        self.step_skip = self._vec.step_skip
        self.n_elems = n_elems
        self.n_seg = n_elems - 1
        self.n_action = 1
        self.action_space = Box(-22.0, 22.0, shape=(1,), dtype=np.float32)
        self.observation_space = Box(-np.inf, np.inf, shape=(4,), dtype=np.float32)
        self.reward_range = 100.0
        self.config_generate_video = config_generate_video
        self.time = np.float64(0.0)
        self.counter = 0

    def reset(self, *, seed: Optional[int] = None, options: Optional[dict] = None):
        super().reset(seed=seed)
        self._vec._rngs[0] = self.np_random  # env-owned stream, as soft_pendulum.py:114,123
        obs, _ = self._vec.reset(seed=None)
        self.time = np.float64(0.0)
        self.counter = 0
        return np.asarray(obs[0], dtype=np.float32).copy(), {}

    def step(self, action):
        a = np.asarray(action, dtype=np.float32).reshape(1)
        obs, reward, term, trunc, infos = self._vec.step(a)
        self.time = np.float64(infos["time"][0])
        self.counter += 1
        info = {"time": self.time, "TimeLimit.truncated": bool(infos["TimeLimit.truncated"][0])}
        if bool(term[0]):
            print(f" Nan detected in, exiting simulation now. {self.time=}")  # soft_pendulum.py:206
        return (
            np.asarray(obs[0], dtype=np.float32).copy(),
            float(reward[0]),
            bool(term[0]),
            bool(trunc[0]),
            info,
        )

    @property
    def rod_parameters_dict(self):
        """RodCallBack's samples (soft_pendulum.py:117-126) when config_generate_video=True."""
        return self._vec.rod_parameters_dict

    def get_state(self):
        """Current observation (soft_pendulum.py:149-161)."""
        obs = self._vec.backend.observe(None)
        return np.asarray(obs[0].cpu().numpy() if hasattr(obs, "cpu") else obs[0], dtype=np.float32).copy()

    def save_data(self, filename_video, fps):
        """The reference renders `rod_parameters_dict` to a video here (soft_pendulum.py:253-256); drawing is out of
        scope: the data is in `rod_parameters_dict`, nothing is written."""
        if getattr(self._vec, "config_generate_video", False):
            raise NotImplementedError("video generation is outside the hot path; use rod_parameters_dict")

    def render(self):
        """None without a render mode; an (H, W, 3) uint8 frame for "rgb_array" (render.py)."""
        from ..render import render_env

        return render_env(self)

    def close(self):
        from ..render import close_env

        close_env(self)
        self._vec.close()
