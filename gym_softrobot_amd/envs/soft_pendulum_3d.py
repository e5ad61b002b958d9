"""SoftPendulum3D-v0 on the MI355X batched Cosserat-rod stepper.

Mirrors gym_softrobot/envs/soft_pendulum_3d/soft_pendulum_3d.py:20-174 and
soft_pendulum_3d/build.py:15-86: a vertical rod (tilted by up to +-1 degree) whose base
is moved in x-y by a 2-D action; MovingBaseConstraint, GravityForces,
AnalyticalLinearDamper(1.0) and LaplaceDissipationFilter(7) are compiled-in features of
the kernel (SOFTROD_FEATURES_SOFTPENDULUM3D).
"""
from __future__ import annotations

from typing import List, Optional, Sequence, Union

import numpy as np

from .. import _capi
from ..seeding import np_random
from ..spaces import Box
from .soft_pendulum import _GymEnv, _time_table


def initial_tilt(rng: np.random.Generator) -> float:
    """soft_pendulum_3d/build.py:51: one `uniform(-1, 1)` draw per reset."""
    return float(np.deg2rad(rng.uniform(-1.0, 1.0)))


class VecSoftPendulum3DEnv:
    """N parallel SoftPendulum3D-v0 envs resident on one GPU (see VecSoftPendulumEnv)."""

    metadata = {"render_modes": ["rgb_array"], "render_fps": 25}

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
        backend=None,
    ):
        if render_mode not in {None, *self.metadata["render_modes"]}:
            raise ValueError(f"Unsupported render mode: {render_mode}")
        if config_generate_video:
            raise NotImplementedError("diagnostic callbacks/video are outside the hot path (DESIGN.md)")
        self.render_mode = render_mode
        self.num_envs = int(num_envs)
        self.final_time = final_time
        self.time_step = time_step
        self.total_steps = int(self.final_time / self.time_step)
        self.recording_fps = recording_fps
        self.step_skip = int(1.0 / (recording_fps * time_step))
        self.n_elems = n_elems
        self.numpy_output = numpy_output
        self.n_action = 2
        self.base_step = 1e-3
        self.base_limit = 0.5
        self.single_action_space = Box(-1.0, 1.0, shape=(2,), dtype=np.float32)
        self.single_observation_space = Box(-np.inf, np.inf, shape=(9,), dtype=np.float32)
        self.action_space = Box(-1.0, 1.0, shape=(self.num_envs, 2), dtype=np.float32)
        self.observation_space = Box(-np.inf, np.inf, shape=(self.num_envs, 9), dtype=np.float32)

        self.cfg = _capi.softpendulum3d_config(
            self.num_envs, final_time=final_time, time_step=time_step,
            recording_fps=recording_fps, n_elems=n_elems, math_mode=math_mode,
        )
        if backend is None:
            from ..backend import HipRodBackend

            backend = HipRodBackend(self.cfg, device=device)
        self.backend = backend
        self._rngs: List[Optional[np.random.Generator]] = [None] * self.num_envs
        import torch

        self._prev_action = torch.zeros((self.num_envs, 2), dtype=torch.float32, device=self.backend.device)
        self._steps = np.zeros(self.num_envs, np.int64)
        self._time_tab = _time_table(self.cfg, 8)

    def _times(self) -> np.ndarray:
        kmax = int(self._steps.max()) if self.num_envs else 0
        if kmax >= len(self._time_tab):
            self._time_tab = _time_table(self.cfg, max(2 * kmax, 16))
        return self._time_tab[self._steps]

    def _out(self, t):
        return t.cpu().numpy() if self.numpy_output else t

    def reset(
        self,
        *,
        seed: Optional[Union[int, Sequence[Optional[int]]]] = None,
        options: Optional[dict] = None,
        mask: Optional[np.ndarray] = None,
    ):
        import torch

        n = self.num_envs
        if seed is None or isinstance(seed, (int, np.integer)):
            seeds = [None if seed is None else int(seed) + i for i in range(n)]
        else:
            seeds = list(seed)
            if len(seeds) != n:
                raise ValueError("need one seed per env")
        m = np.ones(n, bool) if mask is None else np.asarray(mask, bool).reshape(n)
        direction = np.zeros((n, 3))
        direction[:, 2] = 1.0
        for i in range(n):
            if not m[i]:
                continue
            if seeds[i] is not None or self._rngs[i] is None:
                self._rngs[i], _ = np_random(seeds[i])
            tilt = initial_tilt(self._rngs[i])
            direction[i] = [np.sin(tilt), 0.0, np.cos(tilt)]  # soft_pendulum_3d/build.py:52
        start = np.zeros((n, 3))
        normal = np.tile(np.array([0.0, 1.0, 0.0]), (n, 1))   # :53
        self.backend.reset_straight(start, direction, normal, None if mask is None else m.astype(np.uint8))
        self._steps[m] = 0
        # SoftPendulum3DEnv.reset clears _prev_action (soft_pendulum_3d.py:68)
        self._prev_action[torch.from_numpy(m).to(self._prev_action.device)] = 0.0
        obs = self.backend.observe(self._prev_action)
        return self._out(obs), {}

    def step(self, actions):
        import torch

        if isinstance(actions, np.ndarray) or not torch.is_tensor(actions):
            a_np = np.asarray(actions, dtype=np.float32).reshape(self.num_envs, 2)
            if not (np.all(a_np >= -1.0) and np.all(a_np <= 1.0)):
                # soft_pendulum_3d.py:116-117 (device tensors are the caller's responsibility)
                raise ValueError(f"Action {actions!r} is outside {self.single_action_space}")
        a = torch.as_tensor(actions, dtype=torch.float32, device=self.backend.device)
        a = a.reshape(self.num_envs, 2)
        obs, reward, term, trunc = self.backend.step(a)
        self._prev_action = a.detach().clone()
        self._steps += 1
        times = self._times()
        infos = {"time": times, "tilt": self._out(self.backend.aux[:, 0])}
        return (
            self._out(obs),
            self._out(reward),
            self._out(term.bool()),
            self._out(trunc.bool()),
            infos,
        )

    def close(self):
        if self.backend is not None and hasattr(self.backend, "close"):
            self.backend.close()


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

    def render(self):
        if self.render_mode is None:
            return None
        raise NotImplementedError("rendering is outside the hot path (DESIGN.md, out of scope)")

    def close(self):
        self._vec.close()
