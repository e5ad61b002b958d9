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

from typing import Any, Dict, List, Optional, Sequence, Tuple, Union

import numpy as np

from .. import _capi
from ..seeding import initial_angle, np_random
from ..spaces import Box

try:  # pragma: no cover
    from gymnasium import Env as _GymEnv  # type: ignore
except Exception:  # noqa: BLE001
    class _GymEnv:  # minimal stand-in for gymnasium.Env
        metadata: Dict[str, Any] = {}
        render_mode = None
        _np_random = None

        def reset(self, *, seed=None, options=None):
            if seed is not None:
                self._np_random, self._np_random_seed = np_random(seed)

        @property
        def np_random(self):
            if self._np_random is None:
                self._np_random, self._np_random_seed = np_random()
            return self._np_random

        @property
        def unwrapped(self):
            return self

        def close(self):
            pass


def _time_table(cfg: _capi.SoftrodConfig, n_steps: int) -> np.ndarray:
    """float64 simulated time after k env.steps, accumulated exactly as
    `self.time = self.do_step(self.simulator, self.time, self.time_step)` does
    (soft_pendulum.py:183-184): PositionVerlet adds dt/2 twice per substep."""
    t = np.float64(0.0)
    half = np.float64(0.5) * np.float64(cfg.dt)
    dt = np.float64(cfg.dt)
    out = np.empty(n_steps + 1, np.float64)
    out[0] = t
    for k in range(1, n_steps + 1):
        for _ in range(int(cfg.n_substeps)):
            if cfg.time_two_half_adds:
                t = t + half
                t = t + half
            else:
                t = t + dt
        out[k] = t
    return out


class VecSoftPendulumEnv:
    """N parallel SoftPendulum-v0 envs resident on one GPU.

    reset(seed=None|int|sequence, options=None) -> (obs[N,4] float32, infos)
    step(actions[N] or [N,1])                   -> (obs, reward[N] float64,
                                                    terminated[N] bool, truncated[N] bool, infos)
    Outputs are torch tensors on the device (zero-copy views of the backend's
    buffers, overwritten by the next call) unless `numpy_output=True`.
    No auto-reset (the reference has none): call `reset(mask=...)`.
    """

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
            raise ValueError(f"Unsupported render mode: {render_mode}")  # soft_pendulum.py:69-70
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
        self.n_seg = n_elems - 1
        self.numpy_output = numpy_output

        self.n_action = 1
        self.single_action_space = Box(-22.0, 22.0, shape=(1,), dtype=np.float32)
        self.single_observation_space = Box(-np.inf, np.inf, shape=(4,), dtype=np.float32)
        self.action_space = Box(-22.0, 22.0, shape=(self.num_envs, 1), dtype=np.float32)
        self.observation_space = Box(-np.inf, np.inf, shape=(self.num_envs, 4), dtype=np.float32)

        self.cfg = _capi.softpendulum_config(
            self.num_envs, final_time=final_time, time_step=time_step,
            recording_fps=recording_fps, n_elems=n_elems, math_mode=math_mode,
        )
        if backend is None:
            from ..backend import HipRodBackend

            backend = HipRodBackend(self.cfg, device=device)
        self.backend = backend
        self._rngs: List[Optional[np.random.Generator]] = [None] * self.num_envs
        # _prev_action is NOT cleared by reset in the reference (soft_pendulum.py:97-99)
        import torch

        self._prev_action = torch.zeros(self.num_envs, dtype=torch.float32, device=self.backend.device)
        self._steps = np.zeros(self.num_envs, np.int64)  # env.steps since each env's reset
        self._time_tab = _time_table(self.cfg, 8)

    # -- helpers -------------------------------------------------------------------
    def _times(self) -> np.ndarray:
        kmax = int(self._steps.max()) if self.num_envs else 0
        if kmax >= len(self._time_tab):
            self._time_tab = _time_table(self.cfg, max(2 * kmax, 16))
        return self._time_tab[self._steps]

    def _out(self, t):
        return t.cpu().numpy() if self.numpy_output else t

    # -- API -----------------------------------------------------------------------
    def reset(
        self,
        *,
        seed: Optional[Union[int, Sequence[Optional[int]]]] = None,
        options: Optional[dict] = None,
        mask: Optional[np.ndarray] = None,
    ):
        n = self.num_envs
        if seed is None or isinstance(seed, (int, np.integer)):
            seeds = [None if seed is None else int(seed) + i for i in range(n)]
        else:
            seeds = list(seed)
            if len(seeds) != n:
                raise ValueError("need one seed per env")
        m = np.ones(n, bool) if mask is None else np.asarray(mask, bool).reshape(n)
        theta0 = np.zeros(n, np.float64)
        for i in range(n):
            if not m[i]:
                continue
            if seeds[i] is not None or self._rngs[i] is None:
                self._rngs[i], _ = np_random(seeds[i])
            theta0[i] = initial_angle(self._rngs[i])  # build.py:47-49
        self.backend.reset(theta0, None if mask is None else m.astype(np.uint8))
        self._steps[m] = 0
        obs = self.backend.observe(self._prev_action)
        return self._out(obs), {}

    def step(self, actions):
        import torch

        a = torch.as_tensor(actions, dtype=torch.float32, device=self.backend.device)
        a = a.reshape(self.num_envs)
        obs, reward, term, trunc = self.backend.step(a)
        self._prev_action = a.detach().clone()  # set_action: _prev_action[:] = action (:165)
        self._steps += 1
        times = self._times()
        infos = {"time": times, "TimeLimit.truncated": times > self.final_time}
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

    def render(self):
        if self.render_mode is None:
            return None
        raise NotImplementedError("rendering is outside the hot path (DESIGN.md, out of scope)")

    def close(self):
        self._vec.close()
