"""OctoArmSingle-v0 on the MI355X batched Cosserat-rod stepper.

Mirrors gym_softrobot/envs/octopus/arm_single_env.py:41-316 and build_arm
(gym_softrobot/envs/octopus/build.py:220-292): one arm lying on a frictional plane,
actuated by its rest curvature (7 cubic-spline knots), reaching for a target with its
centre of mass.  GravityForces, RodPlaneContactWithAnisotropicFriction and
AnalyticalLinearDamper are compiled-in features (SOFTROD_FEATURES_ARM_SINGLE); the
cubic `interp1d` of `set_action` is a constant basis matrix applied in the kernel
prologue.  `n_elems` other than 50 generalise the reference's hard-coded 7x7 binning of
the curvature observation (:193-194) to np.array_split-style bins.
"""
from __future__ import annotations

from typing import Optional

import numpy as np

from .. import _capi
from ..spaces import Box
from .base import GymEnv as _GymEnv
from .base import VecRodEnvBase


class VecArmSingleEnv(VecRodEnvBase):
    """N parallel OctoArmSingle-v0 envs resident on one GPU (see VecRodEnvBase)."""

    metadata = {"render_modes": ["rgb_array", "human"], "render_fps": 20}
    action_low, action_high = -22.0, 22.0             # arm_single_env.py:84-90

    def __init__(
        self,
        num_envs: int,
        final_time: float = 10.0,
        time_step: float = 7.0e-5,
        recording_fps: int = 20,
        n_elems: int = 50,
        n_action: int = 7,
        control_penalty_coeff: float = 0.001,
        config_generate_video: bool = False,
        policy_mode: str = "centralized",
        render_mode: Optional[str] = None,
        *,
        device: int = 0,
        math_mode: int = _capi.MATH_FAST,
        numpy_output: bool = False,
        autoreset: bool = False,
        backend=None,
        radius_profile=None,
    ):
        """`radius_profile` (extension, n_elems radii): a TAPERED arm, as the reference's muscle arms
        are built (`CosseratRod.straight_rod(base_radius=<array>)`, octopus/arm_push_env.py:160-179);
        None = the uniform arm of build_arm."""
        if n_action != 7:
            raise NotImplementedError("the observation layout of the reference fixes n_action = 7")
        cfg = _capi.arm_single_config(
            num_envs, final_time=final_time, time_step=time_step, recording_fps=recording_fps,
            n_elems=n_elems, control_penalty_coeff=control_penalty_coeff, math_mode=math_mode,
        )
        super().__init__(num_envs, cfg, render_mode=render_mode,
                         config_generate_video=config_generate_video, device=device,
                         numpy_output=numpy_output, autoreset=autoreset, backend=backend)
        self.final_time = final_time
        self.time_step = time_step
        self.total_steps = int(self.final_time / self.time_step)
        self.recording_fps = recording_fps
        self.step_skip = int(1.0 / (recording_fps * time_step))
        self.control_penalty_coeff = control_penalty_coeff
        self.n_elems = n_elems
        self.n_seg = n_elems - 1
        self.policy_mode = policy_mode
        if radius_profile is not None:
            self.backend.set_radius_profile(np.asarray(radius_profile, np.float64))

    def _draw_reset(self, i):
        return None                                   # build_arm draws nothing from the RNG

    def _queue_from_draws(self, draws, counts):
        n, m = self.num_envs, max(1, int(counts.max()))
        start = np.zeros((n, m, 3))
        direction = np.tile(np.array([1.0, 0.0, 0.0]), (n, m, 1))
        normal = np.tile(np.array([0.0, 0.0, 1.0]), (n, m, 1))
        self.backend.queue_push_straight(start, direction, normal, counts)

    def _reset_backend(self, mask, use_mask, draws=None):
        # build_arm draws nothing from the RNG: every reset starts from the same straight arm
        n = self.num_envs
        start = np.zeros((n, 3))
        direction = np.tile(np.array([1.0, 0.0, 0.0]), (n, 1))   # octopus/build.py:236-238
        normal = np.tile(np.array([0.0, 0.0, 1.0]), (n, 1))
        self.backend.reset_straight(start, direction, normal, mask.astype(np.uint8) if use_mask else None)


class ArmSingleEnv(_GymEnv):
    """Drop-in for gym_softrobot's ArmSingleEnv (octopus/arm_single_env.py:41-316), N = 1."""

    metadata = {"render_modes": ["rgb_array", "human"], "render_fps": 20}

    def __init__(
        self,
        final_time=10.0,
        time_step=7.0e-5,
        recording_fps=20,
        n_elems=50,
        n_action=7,
        control_penalty_coeff=0.001,
        config_generate_video=False,
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
        self._vec = VecArmSingleEnv(
            1, final_time, time_step, recording_fps, n_elems, n_action, control_penalty_coeff,
            config_generate_video, policy_mode, None, device=device, math_mode=math_mode,
            numpy_output=True, backend=backend,
        )
        self.final_time = final_time
        self.time_step = time_step
        self.total_steps = self._vec.total_steps
        self.recording_fps = recording_fps
        self.step_skip = self._vec.step_skip
        self.control_penalty_coeff = control_penalty_coeff
        self.n_elems = n_elems
        self.n_seg = n_elems - 1
        self.policy_mode = policy_mode
        self.n_action = n_action
        self.action_space = Box(-22.0, 22.0, shape=(7,), dtype=np.float32)
        self.observation_space = Box(-np.inf, np.inf, shape=(25,), dtype=np.float32)
        self.reward_range = 10.0
        self.kappa_range = [-49.33508476187419, 49.33545827754751]
        self.kappa_rate_range = [-21.063520620377012, 24.664591289161944]
        self._target = np.array([1.0, 0.0])
        self.time = np.float64(0.0)
        self.counter = 0

    def reset(self, *, seed: Optional[int] = None, options: Optional[dict] = None):
        super().reset(seed=seed)
        obs, _ = self._vec.reset()
        self.time = np.float64(0.0)
        self.counter = 0
        return np.asarray(obs[0], dtype=np.float32).copy(), {}

    def step(self, action):
        a = np.asarray(action, dtype=np.float32).reshape(1, 7)
        obs, reward, term, trunc, infos = self._vec.step(a)
        self.time = np.float64(infos["time"][0])
        self.counter += 1
        return (
            np.asarray(obs[0], dtype=np.float32).copy(),
            float(reward[0]),
            bool(term[0]),
            bool(trunc[0]),
            {"time": self.time, "TimeLimit.truncated": bool(infos["TimeLimit.truncated"][0])},
        )

    def get_state(self):
        """Current observation (arm_single_env.py:186-224); like the reference's, a call moves the
        `prev_kappa_state` / `prev_com_state` the rate entries are taken against."""
        obs = self._vec.backend.observe(None)
        return np.asarray(obs[0].cpu().numpy() if hasattr(obs, "cpu") else obs[0], dtype=np.float32).copy()

    def summary(self):
        """As the reference's summary() (octopus/arm_single_env.py:116-133)."""
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
        """The reference renders `rod_parameters_dict` to a video here (arm_single_env.py); drawing is out of
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
