"""OctoCrawl-v0, OctoArmTwo-v0, OctoReach-v0 on the MI355X batched Cosserat-rod stepper: the muscle octopus.

Mirrors gym_softrobot/envs/octopus/crawl_env.py (CrawlEnv), arm_two_env.py (ArmTwoEnv) and reach_env.py (ReachEnv)
over build_octopus_muscles / build_two_arms (octopus/build_muscle_octopus.py): 8 (or 2) tapered 20-element arms with
COOMM muscle layers, joined to a rigid head by FixedJoint2Rigid, held by ControllableFixConstraint "suckers"; no
gravity, no plane.  One env = one workgroup of n_arm * 32 lanes; set_action / get_state / reward are two small kernels
around the body's substep kernel (csrc/softrod_mocto.hpp).

PARITY UNPINNED.  The muscle force law lives in COOMM (`coomm`, a git pin of uv.lock:173-175), which is not on disk; the
kernels restate the published model exactly as for OctoArmPush.  What IS pinned against the
executed reference files: the builds, the observation layouts, set_action's mapping and the reward / termination code
(tests/golden/ref_muscle_octopus.npz, tools/make_muscle_env_golden.py).

Single-agent Gymnasium API only (flat observation); the "multiagent": PyMARL metadata of CrawlEnv is kept as data.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from .. import _capi
from ..spaces import Box
from .arm_push import PARITY_LABEL
from .base import GymEnv as _GymEnv
from .base import VecRodEnvBase


class _VecMuscleOctopusEnv(VecRodEnvBase):
    parity_label = PARITY_LABEL
    metadata = {"render_modes": ["rgb_array"], "render_fps": 25}
    action_low, action_high = 0.0, 1.0
    env_kind = None
    default_final_time = 5.0

    def __init__(
        self,
        num_envs: int,
        final_time: Optional[float] = None,
        time_step: float = 5.0e-5,
        recording_fps: int = 25,
        n_elems: int = 20,
        render_mode: Optional[str] = None,
        *,
        device: int = 0,
        math_mode: int = _capi.MATH_FAST,
        numpy_output: bool = False,
        autoreset: bool = False,
        backend=None,
        muscle_kwargs: Optional[dict] = None,
    ):
        final_time = self.default_final_time if final_time is None else final_time
        cfg = _capi.muscle_octopus_config(self.env_kind, num_envs, final_time=final_time, time_step=time_step,
                                          recording_fps=recording_fps, n_elems=n_elems, math_mode=math_mode)
        super().__init__(num_envs, cfg, render_mode=render_mode, config_generate_video=False, device=device,
                         numpy_output=numpy_output, autoreset=autoreset, backend=backend)
        self.final_time = final_time
        self.time_step = time_step
        self.total_steps = int(self.final_time / self.time_step)
        self.recording_fps = recording_fps
        self.step_skip = int(1.0 / (recording_fps * time_step))
        self.n_arm = int(cfg.n_arm)
        self.n_elems = n_elems
        self.n_seg = n_elems - 1
        self.n_action = int(cfg.n_knots)                  # per arm, as in the reference
        self.reward_range = 100.0
        # per env: the target's x, y, z and the episode's own final_time (0: the config's; CrawlEnv's random one)
        self.targets = np.zeros((num_envs, 4), np.float64)
        radii = _capi.muscle_octopus_radii(n_elems)
        self.backend.set_radius_profile(radii)
        self.backend.set_muscle_layers(*_capi.es_muscle_layers(radii, _capi.MUSCLE_OCTOPUS["base_radius"], **(muscle_kwargs or {})))

    def _draw_reset(self, i):
        return np.array([5.0, 0.0, 0.0, 0.0])             # self._target = np.array([5, 0], dtype=np.float32) (crawl_env.py:172)

    def _queue_from_draws(self, draws, counts):
        tg = np.zeros((self.num_envs, max(1, int(counts.max())), 4))
        for i, d in enumerate(draws):
            for j, v in enumerate(d):
                tg[i, j] = v
        self.backend.queue_push_octo(tg, counts)

    def _reset_backend(self, mask, use_mask, draws=None):
        for i in np.nonzero(mask)[0]:
            self.targets[i] = self._draw(i, draws)
        self.backend.reset_octo(self.targets, mask.astype(np.uint8) if use_mask else None)


class VecCrawlEnv(_VecMuscleOctopusEnv):
    """N parallel OctoCrawl-v0 envs (crawl_env.py:38-303): per arm (sucker location, transverse activation, sucker
    reduction ratio); reward = progress of the head towards (5, 0)."""

    metadata = {"render_modes": ["rgb_array"], "render_fps": 25, "multiagent": ["PyMARL"]}
    env_kind = _capi.ENV_CRAWL
    default_final_time = 10.0
    agent_id = ["LF1", "LF2", "LB2", "LB1", "RB1", "RB2", "RF2", "RF1"]      # crawl_env.py:117-119

    def __init__(self, num_envs: int, final_time: float = 10.0, time_step: float = 5.0e-5, recording_fps: int = 25,
                 n_elems: int = 20, config_random_final_time: bool = False, render_mode: Optional[str] = None, **kw):
        super().__init__(num_envs, final_time, time_step, recording_fps, n_elems, render_mode, **kw)
        self.config_random_final_time = bool(config_random_final_time)

    def _draw_reset(self, i):
        # reset (crawl_env.py:134-136): `if self.config_random_final_time: self.final_time = self.np_random.uniform(3.0, 10.0)`
        # — the first draw of the episode; the target is the constant (5, 0)
        ft = self._rngs[i].uniform(3.0, 10.0) if self.config_random_final_time else 0.0
        return np.array([5.0, 0.0, 0.0, ft])

    @property
    def final_times(self):
        """Every env's current final_time (the configured one unless config_random_final_time drew another)."""
        return np.where(self.targets[:, 3] > 0.0, self.targets[:, 3], self.final_time)

    def _infos(self, times):
        return {"time": times, "TimeLimit.truncated": times > self.final_times}

    def get_env_info(self):
        return dict(n_actions=self.n_action, n_agents=8)                   # crawl_env.py:121-123


class VecArmTwoEnv(_VecMuscleOctopusEnv):
    """N parallel OctoArmTwo-v0 envs (arm_two_env.py:40-343): two arms, per arm three sucker ratios and three knots of each
    muscle layer's activation (cubic interpolation over the arm)."""

    env_kind = _capi.ENV_ARM_TWO

    def __init__(self, num_envs: int, final_time: float = 5.0, time_step: float = 5.0e-5, recording_fps: int = 25,
                 n_elems: int = 20, render_mode: Optional[str] = None, **kw):
        super().__init__(num_envs, final_time, time_step, recording_fps, n_elems, render_mode, **kw)
        self.n_sucker = 3
        _, self.sucker_location = _capi.arm_two_activation_basis(n_elems, 3)
        self.control_location = [0] + self.sucker_location + [n_elems - 1]

    def get_env_info(self):
        return dict(n_actions=self.n_action, n_agents=8)                   # arm_two_env.py:111-112 (8, as written there)


class VecReachEnv(_VecMuscleOctopusEnv):
    """N parallel OctoReach-v0 envs (reach_env.py:37-290): the head held (OneEndFixedBC), every element of every layer of
    every arm actuated; reward = -(closest tip's distance to a random point / 0.25)^2."""

    env_kind = _capi.ENV_REACH

    def __init__(self, num_envs: int, final_time: float = 5.0, time_step: float = 5.0e-5, recording_fps: int = 25,
                 n_elems: int = 20, render_mode: Optional[str] = None, **kw):
        super().__init__(num_envs, final_time, time_step, recording_fps, n_elems, render_mode, **kw)
        self.n_muscle = 3
        self._rest_length_sum = _capi.muscle_octopus_rest_length_sum(n_elems, float(self.cfg.head_radius))

    def _draw_reset(self, i):
        # reach_env.py:141-143: self._target = self.np_random.random(3) * sum(self.shearable_rods[0].rest_lengths)
        return np.concatenate([self._rngs[i].random(3) * self._rest_length_sum, [0.0]])

    def get_env_info(self):
        return dict(n_actions=self.n_action, n_agents=8)


class _MuscleOctopusEnv(_GymEnv):
    parity_label = PARITY_LABEL
    metadata = {"render_modes": ["rgb_array"], "render_fps": 25}
    vec_class = None

    def __init__(self, final_time: Optional[float] = None, time_step: float = 5.0e-5, recording_fps: int = 25,
                 n_elems: int = 20, render_mode: Optional[str] = None, *, device: int = 0,
                 math_mode: int = _capi.MATH_FAST, backend=None, **kw):
        super().__init__()
        if render_mode not in {None, *self.metadata["render_modes"]}:
            raise ValueError(f"Unsupported render mode: {render_mode}")
        self.render_mode = render_mode
        final_time = self.vec_class.default_final_time if final_time is None else final_time
        self._vec = self.vec_class(1, final_time, time_step, recording_fps, n_elems, render_mode=None, device=device,
                                   math_mode=math_mode, numpy_output=True, backend=backend, **kw)
        v = self._vec
        self.final_time, self.time_step, self.recording_fps = final_time, time_step, recording_fps
        self.total_steps, self.step_skip = v.total_steps, v.step_skip
        self.n_arm, self.n_elems, self.n_seg, self.n_action = v.n_arm, v.n_elems, v.n_seg, v.n_action
        self.action_space = Box(0.0, 1.0, shape=(self.n_arm * self.n_action,), dtype=np.float32)
        self._observation_size = (v.obs_dim,)
        self.observation_space = Box(-np.inf, np.inf, shape=self._observation_size, dtype=np.float32)
        self.reward_range = 100.0
        self._prev_action = np.zeros(list(self.action_space.shape), dtype=self.action_space.dtype)
        self.time = np.float64(0.0)
        self.counter = 0

    @property
    def _target(self):
        t = self._vec.targets[0]
        return t[:3].copy() if self._vec.env_kind == _capi.ENV_REACH else np.array(t[:2], dtype=np.float32)

    def get_env_info(self):
        return self._vec.get_env_info()

    def reset(self, *, seed: Optional[int] = None, options: Optional[dict] = None):
        super().reset(seed=seed)
        self._vec._rngs[0] = self.np_random               # env-owned stream (ReachEnv draws its target from it)
        obs, _ = self._vec.reset(seed=None)
        self.time = np.float64(0.0)
        self.counter = 0
        return np.asarray(obs[0], dtype=np.float32).copy(), {}

    def get_state(self):
        obs = self._vec.backend.observe(None)
        return np.asarray(obs[0].cpu().numpy() if hasattr(obs, "cpu") else obs[0], dtype=np.float32).copy()

    def step(self, action):
        a = np.asarray(action, dtype=np.float32).reshape(1, self.n_arm * self.n_action)
        obs, reward, term, trunc, infos = self._vec.step(a)
        self._prev_action = np.reshape(a[0], [self.n_arm, self.n_action])
        self.time = np.float64(infos["time"][0])
        self.counter += 1
        return (np.asarray(obs[0], dtype=np.float32).copy(), float(reward[0]), bool(term[0]), bool(trunc[0]),
                {"time": self.time, "TimeLimit.truncated": bool(infos["TimeLimit.truncated"][0])})

    def render(self):
        from ..render import render_env

        return render_env(self)

    def close(self):
        from ..render import close_env

        close_env(self)
        self._vec.close()


class CrawlEnv(_MuscleOctopusEnv):
    """Drop-in for gym_softrobot's CrawlEnv (octopus/crawl_env.py:38-380), N = 1.  PARITY UNPINNED."""

    metadata = {"render_modes": ["rgb_array"], "render_fps": 25, "multiagent": ["PyMARL"]}
    vec_class = VecCrawlEnv

    def __init__(self, final_time=10.0, time_step=5.0e-5, recording_fps=25, n_elems=20, config_random_final_time=False,
                 render_mode: Optional[str] = None, **kw):
        super().__init__(final_time, time_step, recording_fps, n_elems, render_mode,
                         config_random_final_time=config_random_final_time, **kw)
        self.n_agent = self.n_arm
        self.config_random_final_time = config_random_final_time

    def reset(self, *, seed=None, options=None):
        out = super().reset(seed=seed, options=options)
        self.final_time = float(self._vec.final_times[0])          # crawl_env.py:135-136
        return out

    @property
    def agent_id(self):
        return list(VecCrawlEnv.agent_id)


class ArmTwoEnv(_MuscleOctopusEnv):
    """Drop-in for gym_softrobot's ArmTwoEnv (octopus/arm_two_env.py:40-417), N = 1.  PARITY UNPINNED."""

    vec_class = VecArmTwoEnv

    def __init__(self, final_time=5.0, time_step=5.0e-5, recording_fps=25, n_elems=20, render_mode: Optional[str] = None, **kw):
        super().__init__(final_time, time_step, recording_fps, n_elems, render_mode, **kw)
        self.n_sucker = 3
        self.sucker_location = self._vec.sucker_location
        self.control_location = self._vec.control_location


class ReachEnv(_MuscleOctopusEnv):
    """Drop-in for gym_softrobot's ReachEnv (octopus/reach_env.py:37-369), N = 1.  PARITY UNPINNED."""

    vec_class = VecReachEnv

    def __init__(self, final_time=5.0, time_step=5.0e-5, recording_fps=25, n_elems=20, render_mode: Optional[str] = None, **kw):
        super().__init__(final_time, time_step, recording_fps, n_elems, render_mode, **kw)
        self.n_muscle = 3
