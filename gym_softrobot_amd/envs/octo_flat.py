"""OctoFlat-v0 / OctoFlatLite-v0 on the MI355X batched Cosserat-rod stepper.

Mirrors gym_softrobot/envs/octopus/flat_env.py:40-408 and build_octopus
(gym_softrobot/envs/octopus/build.py:52-217): `n_arm` arms on a frictional plane joined to a
rigid cylindrical head by FixedJoint2Rigid (utils/custom_elastica/joint.py), the head held in
the plane by BodyBoundaryCondition (utils/custom_elastica/constraint.py); every arm is
actuated by its rest curvature (`n_action` cubic-spline knots, zero at both ends) and the
head has to reach a random target.  One env = one workgroup (softrod_octo.hpp).

Only `policy_mode="centralized"` (the registered default) is implemented.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from .. import _capi
from ..spaces import Box, Dict
from .base import GymEnv as _GymEnv
from .base import VecRodEnvBase


class VecOctoFlatEnv(VecRodEnvBase):
    """N parallel OctoFlat-v0 envs resident on one GPU (see VecRodEnvBase).

    Observations come back flat, (N, n_arm*width + 13) float32: the "individual" rows
    followed by "shared" (flat_env.py:231-286); `split_obs` gives the reference's dict."""

    metadata = {"render_modes": ["rgb_array", "human"], "render_fps": 5}
    action_low, action_high = -22.0, 22.0             # flat_env.py:88-95

    def __init__(
        self,
        num_envs: int,
        final_time: float = 5.0,
        time_step: float = 7.0e-5,
        recording_fps: int = 5,
        n_elems: int = 10,
        n_arm: int = 8,
        n_action: int = 3,
        config_generate_video: bool = False,
        config_save_head_data: bool = False,
        policy_mode: str = "centralized",
        render_mode: Optional[str] = None,
        *,
        device: int = 0,
        math_mode: int = _capi.MATH_FAST,
        numpy_output: bool = False,
        autoreset: bool = False,
        backend=None,
    ):
        if policy_mode not in ("centralized", "decentralized"):
            raise NotImplementedError                   # flat_env.py:131-132
        self.config_save_head_data = bool(config_save_head_data)
        cfg = _capi.octo_flat_config(
            num_envs, final_time=final_time, time_step=time_step, recording_fps=recording_fps,
            n_elems=n_elems, n_arm=n_arm, n_action=n_action, math_mode=math_mode,
        )
        super().__init__(num_envs, cfg, render_mode=render_mode,
                         config_generate_video=config_generate_video, device=device,
                         numpy_output=numpy_output, autoreset=autoreset, backend=backend)
        self.final_time = final_time
        self.time_step = time_step
        self.total_steps = int(self.final_time / self.time_step)
        self.recording_fps = recording_fps
        self.step_skip = int(1.0 / (recording_fps * time_step))
        self.n_arm = n_arm
        self.n_elems = n_elems
        self.n_seg = n_elems - 1
        self.n_action = n_action
        self.policy_mode = policy_mode
        self.individual_shape = (n_arm, self.n_seg + (n_elems + 1) * 4 + n_action)
        self.targets = np.zeros((num_envs, 2), np.float64)

    def _draw_reset(self, i):
        # flat_env.py:221: self._target = (2 - 0.5) * self.np_random.random(2) + 0.5
        return (2 - 0.5) * self._rngs[i].random(2) + 0.5

    def _queue_from_draws(self, draws, counts):
        tg = np.zeros((self.num_envs, max(1, int(counts.max())), 2))
        for i, d in enumerate(draws):
            for j, v in enumerate(d):
                tg[i, j] = v
        self.backend.queue_push_octo(tg, counts)

    def _reset_backend(self, mask, use_mask, draws=None):
        for i in np.nonzero(mask)[0]:
            self.targets[i] = self._draw(i, draws)
        self.backend.reset_octo(self.targets, mask.astype(np.uint8) if use_mask else None)

    def split_obs(self, obs):
        """(N, obs_dim) -> {"individual": (N, n_arm, width), "shared": (N, 13)} (views).
        policy_mode "decentralized" (flat_env.py:248-260) appends the one-hot arm index to every
        arm's row; the kernel's rows are the centralized ones, the identity is added here."""
        na, w = self.individual_shape
        ind = obs[:, : na * w].reshape(-1, na, w)
        if self.policy_mode == "decentralized":
            if isinstance(ind, np.ndarray):
                eye = np.broadcast_to(np.eye(na, dtype=ind.dtype), (ind.shape[0], na, na))
                ind = np.concatenate([ind, eye], axis=-1)
            else:
                import torch

                eye = torch.eye(na, dtype=ind.dtype, device=ind.device).expand(ind.shape[0], na, na)
                ind = torch.cat([ind, eye], dim=-1)
        return {"individual": ind, "shared": obs[:, na * w:]}


class FlatEnv(_GymEnv):
    """Drop-in for gym_softrobot's FlatEnv (octopus/flat_env.py:40-408), N = 1."""

    metadata = {"render_modes": ["rgb_array", "human"], "render_fps": 5}

    def __init__(
        self,
        final_time=5.0,
        time_step=7.0e-5,
        recording_fps=5,
        n_elems=10,
        n_arm=8,
        n_action=3,
        config_generate_video=False,
        config_save_head_data=False,
        policy_mode="centralized",
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
        self._vec = VecOctoFlatEnv(
            1, final_time, time_step, recording_fps, n_elems, n_arm, n_action, config_generate_video,
            config_save_head_data, policy_mode, None, device=device, math_mode=math_mode,
            numpy_output=True, backend=backend,
        )
        self.final_time = final_time
        self.time_step = time_step
        self.total_steps = self._vec.total_steps
        self.recording_fps = recording_fps
        self.step_skip = self._vec.step_skip
        self.n_arm = n_arm
        self.n_elems = n_elems
        self.n_seg = n_elems - 1
        self.n_action = n_action
        self.policy_mode = policy_mode
        if policy_mode == "centralized":
            lo = np.repeat(np.ones(n_action) * (-22), n_arm)
            hi = np.repeat(np.ones(n_action) * (22), n_arm)
            self.action_space = Box(lo, hi, shape=(n_arm * n_action,), dtype=np.float32)
            self._observation_size = self._vec.individual_shape
        else:
            # flat_env.py:111-130: the declared spaces are ONE arm's; step() still takes all
            # n_arm * n_action values (set_action reshapes them, :288-289)
            self.action_space = Box(np.ones(n_action) * (-22), np.ones(n_action) * 22, shape=(n_action,),
                                    dtype=np.float32)
            self._observation_size = (self._vec.individual_shape[1] + n_arm,)
        self.observation_space = Dict({
            "individual": Box(-np.inf, np.inf, shape=self._observation_size, dtype=np.float32),
            "shared": Box(-np.inf, np.inf, shape=(13,), dtype=np.float32),
        })
        self.reward_range = 100.0
        self.time = np.float64(0.0)
        self.counter = 0

    @property
    def _target(self):
        return self._vec.targets[0]

    def _state(self, obs):
        d = self._vec.split_obs(np.asarray(obs, dtype=np.float32))
        return {"individual": d["individual"][0].copy(), "shared": d["shared"][0].copy()}

    def reset(self, *, seed: Optional[int] = None, options: Optional[dict] = None):
        super().reset(seed=seed)
        self._vec._rngs[0] = self.np_random   # env-owned stream: the target draw advances env.np_random (flat_env.py:171,221)
        obs, _ = self._vec.reset(seed=None)
        self.time = np.float64(0.0)
        self.counter = 0
        return self._state(obs), {}

    def step(self, action):
        a = np.asarray(action, dtype=np.float32).reshape(1, self.n_arm * self.n_action)
        obs, reward, term, trunc, infos = self._vec.step(a)
        self.time = np.float64(infos["time"][0])
        self.counter += 1
        return (
            self._state(obs),
            float(reward[0]),
            bool(term[0]),
            bool(trunc[0]),
            {"time": self.time, "TimeLimit.truncated": bool(infos["TimeLimit.truncated"][0])},
        )

    def get_state(self):
        """Current observation dict (flat_env.py:231-286)."""
        obs = self._vec.backend.observe(None)
        return self._state(obs.cpu().numpy() if hasattr(obs, "cpu") else obs)

    def summary(self):
        """As the reference's summary() (octopus/flat_env.py:153-170)."""
        print(
            f"""
        {self.final_time=}
        {self.time_step=}
        {self.total_steps=}
        {self.step_skip=}
        simulation time per action: {1.0/self.step_skip=}
        max number of action per episode: {self.total_steps / self.step_skip}

        {self.n_elems=}
        {self.action_space=}
        {self.observation_space=}
        {self.reward_range=}
        """
        )

    def save_data(self, filename_video, fps):
        """The reference renders `rod_parameters_dict` to a video here (flat_env.py:410-420); drawing is out of
        scope: the data is in `rod_parameters_dict`, nothing is written."""
        if getattr(self._vec, "config_generate_video", False):
            raise NotImplementedError("video generation is outside the hot path; use rod_parameters_dict")

    def render(self):
        """None without a render mode; an (H, W, 3) uint8 frame for "rgb_array" (render.py)."""
        from ..render import render_env

        return render_env(self)

    @property
    def rod_parameters_dict_list(self):
        """One RodCallBack dict per arm (flat_env.py:190-198), with config_generate_video=True."""
        return self._vec.rod_parameters_dict_list

    @property
    def head_dict(self):
        """The head's RigidCylinderCallBack dict (flat_env.py:199-206), with config_save_head_data=True."""
        return self._vec.head_dict

    def close(self):
        from ..render import close_env

        close_env(self)
        self._vec.close()
