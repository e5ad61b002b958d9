"""Shared host logic of the batched envs: seeding, masked/auto reset, step bookkeeping.

The reference has a single-env API only (no vector env, no auto-reset: a finished env is
reset by the caller, gym_softrobot/debug/make.py:16-23).  The batched form adds, per
SURVEY.md §8(f) N2:
  * `reset(mask=...)`  partial reset of a subset of the resident envs,
  * `autoreset=True`   Gymnasium-1.0 VectorEnv NEXT_STEP semantics: an env that returned
                       terminated/truncated at step t is reset (instead of stepped) by the
                       call at t+1, which returns its reset observation, reward 0 and both
                       flags False.  Costs one small device->host read of the flags per step.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional, Sequence, Union

import numpy as np

from .. import _capi
from ..seeding import np_random
from ..spaces import Box

try:  # pragma: no cover
    from gymnasium import Env as GymEnv  # type: ignore
except Exception:  # noqa: BLE001
    class GymEnv:  # minimal stand-in for gymnasium.Env
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


def time_table(cfg: _capi.SoftrodConfig, n_steps: int) -> np.ndarray:
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


class VecRodEnvBase:
    """N parallel envs resident on one GPU.

    reset(seed=None|int|sequence, options=None, mask=None) -> (obs[N,obs_dim] float32, infos)
    step(actions[N,action_dim])  -> (obs, reward[N] float64, terminated[N] bool,
                                     truncated[N] bool, infos)
    Outputs are torch tensors on the device (zero-copy views of the backend's buffers,
    overwritten by the next call) unless `numpy_output=True`.
    """

    metadata: Dict[str, Any] = {"render_modes": ["rgb_array"], "render_fps": 25}
    action_low: float = -1.0
    action_high: float = 1.0

    def __init__(self, num_envs: int, cfg: _capi.SoftrodConfig, *, render_mode, config_generate_video,
                 device: int, numpy_output: bool, autoreset: bool, backend):
        if render_mode not in {None, *self.metadata["render_modes"]}:
            raise ValueError(f"Unsupported render mode: {render_mode}")  # soft_pendulum.py:69-70
        if config_generate_video:
            raise NotImplementedError("diagnostic callbacks/video are outside the hot path (DESIGN.md)")
        self.render_mode = render_mode
        self.num_envs = int(num_envs)
        self.cfg = cfg
        self.numpy_output = numpy_output
        self.autoreset = bool(autoreset)
        self.action_dim = _capi.config_action_dim(cfg)
        self.obs_dim = _capi.config_obs_dim(cfg)
        self.n_action = self.action_dim
        lo, hi = self.action_low, self.action_high
        self.single_action_space = Box(lo, hi, shape=(self.action_dim,), dtype=np.float32)
        self.single_observation_space = Box(-np.inf, np.inf, shape=(self.obs_dim,), dtype=np.float32)
        self.action_space = Box(lo, hi, shape=(self.num_envs, self.action_dim), dtype=np.float32)
        self.observation_space = Box(-np.inf, np.inf, shape=(self.num_envs, self.obs_dim), dtype=np.float32)
        if backend is None:
            from ..backend import HipRodBackend

            backend = HipRodBackend(cfg, device=device)
        self.backend = backend
        self._rngs: List[Optional[np.random.Generator]] = [None] * self.num_envs
        self._steps = np.zeros(self.num_envs, np.int64)  # env.steps since each env's reset
        self._time_tab = time_table(cfg, 8)
        self._needs_reset = np.zeros(self.num_envs, bool)

    # -- hooks ---------------------------------------------------------------------
    def _reset_backend(self, mask: np.ndarray, use_mask: bool) -> None:
        """Draw what the env's build function draws from self._rngs[i] for masked envs and
        reset those rods on the backend."""
        raise NotImplementedError

    def _infos(self, times: np.ndarray) -> Dict[str, Any]:
        return {"time": times, "TimeLimit.truncated": times > self.cfg.final_time}

    def _validate_actions(self, actions) -> None:
        pass

    # -- helpers -------------------------------------------------------------------
    def _times(self) -> np.ndarray:
        kmax = int(self._steps.max()) if self.num_envs else 0
        if kmax >= len(self._time_tab):
            self._time_tab = time_table(self.cfg, max(2 * kmax, 16))
        return self._time_tab[self._steps]

    def _out(self, t):
        return t.cpu().numpy() if self.numpy_output else t

    def _seed_rngs(self, seed, mask: np.ndarray) -> None:
        n = self.num_envs
        if seed is None or isinstance(seed, (int, np.integer)):
            seeds = [None if seed is None else int(seed) + i for i in range(n)]
        else:
            seeds = list(seed)
            if len(seeds) != n:
                raise ValueError("need one seed per env")
        for i in range(n):
            if mask[i] and (seeds[i] is not None or self._rngs[i] is None):
                self._rngs[i], _ = np_random(seeds[i])

    # -- API -----------------------------------------------------------------------
    def reset(
        self,
        *,
        seed: Optional[Union[int, Sequence[Optional[int]]]] = None,
        options: Optional[dict] = None,
        mask: Optional[np.ndarray] = None,
    ):
        n = self.num_envs
        m = np.ones(n, bool) if mask is None else np.asarray(mask, bool).reshape(n)
        self._seed_rngs(seed, m)
        self._reset_backend(m, mask is not None)
        self._steps[m] = 0
        self._needs_reset[m] = False
        # _prev_action lives with the resident state (softrod_state_view.prev_action): it
        # survives reset except where the reference clears it (soft_pendulum_3d.py:68)
        obs = self.backend.observe(None)
        return self._out(obs), {}

    def step(self, actions):
        import torch

        self._validate_actions(actions)
        a = torch.as_tensor(actions, dtype=torch.float32, device=self.backend.device)
        a = a.reshape(self.num_envs, self.action_dim)
        pending = self._needs_reset.copy() if self.autoreset else None
        obs, reward, term, trunc = self.backend.step(a)   # also records _prev_action[:] = action
        self._steps += 1
        if pending is not None and pending.any():
            # NEXT_STEP auto-reset: these envs finished on the previous call; their step above
            # is discarded, they restart and report their reset observation
            keep = obs.clone()
            self._seed_rngs(None, pending)
            self._reset_backend(pending, True)
            self._steps[pending] = 0
            robs = self.backend.observe(None)
            pm = torch.from_numpy(pending).to(robs.device)
            obs = torch.where(pm[:, None], robs, keep)
            reward = torch.where(pm, torch.zeros_like(reward), reward)
            term = torch.where(pm, torch.zeros_like(term), term)
            trunc = torch.where(pm, torch.zeros_like(trunc), trunc)
        if self.autoreset:
            self._needs_reset = (term | trunc).cpu().numpy().astype(bool)
        times = self._times()
        infos = self._infos(times)
        return (
            self._out(obs),
            self._out(reward),
            self._out(term.view(torch.bool)),     # uint8 0/1 -> bool, zero-copy
            self._out(trunc.view(torch.bool)),
            infos,
        )

    def step_packed(self, actions):
        """step() with every per-env output in one (N, packed_width) float32 buffer written
        by the kernel itself (distributed.unpack_outputs gives views); no auto-reset."""
        import torch

        if self.autoreset:
            raise NotImplementedError("step_packed does not auto-reset; use step()")
        self._validate_actions(actions)
        a = torch.as_tensor(actions, dtype=torch.float32, device=self.backend.device)
        packed = self.backend.step_packed(a.reshape(self.num_envs, self.action_dim))
        self._steps += 1
        return packed, self._infos(self._times())

    def close(self):
        if self.backend is not None and hasattr(self.backend, "close"):
            self.backend.close()
