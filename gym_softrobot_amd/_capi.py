"""ctypes binding of include/softrod.h (the C-ABI of libsoftrod_hip.so).

This is the binding a gym-softrobot maintainer would add to call the HIP stepper
from `SoftPendulumEnv.step`.  There is NO CPU fallback: if the
shared library is missing or a call fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
import os
from pathlib import Path

ABI_VERSION = 16

# softrod_feature (include/softrod.h)
FEAT_GRAVITY = 1 << 0
FEAT_POINT_FORCE_NODE0_X = 1 << 1
FEAT_PENDULUM_BC = 1 << 2
FEAT_ANALYTICAL_DAMPER = 1 << 3
FEAT_FIXED_BC = 1 << 4
FEAT_TIP_FORCE = 1 << 5
FEAT_MOVING_BASE_BC = 1 << 6
FEAT_LAPLACE_FILTER = 1 << 7
FEAT_PLANE_CONTACT_ANISO = 1 << 8
FEAT_REST_KAPPA_ACTION = 1 << 9
FEAT_OCTO_HEAD = 1 << 10
FEAT_SPLINE_MUSCLE_TORQUES = 1 << 11
FEAT_SUCKER_CONSTRAINT = 1 << 12
FEAT_COOMM_MUSCLES = 1 << 13
MAX_SUCKERS = 4
MAX_MUSCLES = 4
MUSCLE_LONGITUDINAL = 0
MUSCLE_TRANSVERSE = 1
MATERIAL_ROWS = 16
FEATURES_SOFTPENDULUM = (
    FEAT_GRAVITY | FEAT_POINT_FORCE_NODE0_X | FEAT_PENDULUM_BC | FEAT_ANALYTICAL_DAMPER
)
FEATURES_SOFTPENDULUM3D = (
    FEAT_GRAVITY | FEAT_MOVING_BASE_BC | FEAT_ANALYTICAL_DAMPER | FEAT_LAPLACE_FILTER
)

FEATURES_ARM_SINGLE = (
    FEAT_GRAVITY | FEAT_PLANE_CONTACT_ANISO | FEAT_ANALYTICAL_DAMPER | FEAT_REST_KAPPA_ACTION
)

ENV_NONE = 0
ENV_SOFTPENDULUM = 1
ENV_SOFTPENDULUM3D = 2
ENV_ARM_SINGLE = 3
ENV_OCTO_FLAT = 4
ENV_SOFT_ARM = 5
ENV_ARM_PUSH = 6
ENV_ARM_PULL_WEIGHT = 7
ENV_CRAWL = 8
ENV_ARM_TWO = 9
ENV_REACH = 10
MUSCLE_OCTOPUS_ENVS = (ENV_CRAWL, ENV_ARM_TWO, ENV_REACH)
FEATURES_ARM_PUSH = FEAT_ANALYTICAL_DAMPER | FEAT_SUCKER_CONSTRAINT | FEAT_COOMM_MUSCLES
FEATURES_ARM_PULL_WEIGHT = FEATURES_ARM_PUSH | FEAT_OCTO_HEAD
FEATURES_OCTO_FLAT = FEATURES_ARM_SINGLE | FEAT_OCTO_HEAD
FEATURES_SOFT_ARM = FEAT_FIXED_BC | FEAT_ANALYTICAL_DAMPER | FEAT_SPLINE_MUSCLE_TORQUES

MATH_LIBM = 0
MATH_FAST = 1

LANE_STRIDE = 64  # one wavefront row per rod (softrod_state_view.lane_stride)

_ACTION_DIM = {ENV_NONE: 1, ENV_SOFTPENDULUM: 1, ENV_SOFTPENDULUM3D: 2, ENV_ARM_SINGLE: 7, ENV_OCTO_FLAT: 24,
               ENV_SOFT_ARM: 8, ENV_ARM_PUSH: 2, ENV_ARM_PULL_WEIGHT: 2, 8: 24, 9: 18, 10: 480}
_OBS_DIM = {ENV_NONE: 4, ENV_SOFTPENDULUM: 4, ENV_SOFTPENDULUM3D: 9, ENV_ARM_SINGLE: 25,
            ENV_OCTO_FLAT: 8 * 56 + 13, ENV_SOFT_ARM: 14, ENV_ARM_PUSH: 84, ENV_ARM_PULL_WEIGHT: 84,
            8: 8 * 131, 9: 2 * 52, 10: 8 * 189}          # CrawlEnv / ArmTwoEnv / ReachEnv at their registered sizes
_AUX_DIM = {ENV_NONE: 0, ENV_SOFTPENDULUM: 0, ENV_SOFTPENDULUM3D: 1, ENV_ARM_SINGLE: 0, ENV_OCTO_FLAT: 0,
            ENV_SOFT_ARM: 0, ENV_ARM_PUSH: 0, ENV_ARM_PULL_WEIGHT: 0, 8: 0, 9: 0, 10: 0}


def action_dim(env_kind: int) -> int:
    return _ACTION_DIM[int(env_kind)]


def obs_dim(env_kind: int) -> int:
    return _OBS_DIM[int(env_kind)]


def aux_dim(env_kind: int) -> int:
    return _AUX_DIM[int(env_kind)]


class SoftrodConfig(C.Structure):
    """Mirror of `struct softrod_config` (include/softrod.h)."""

    _fields_ = [
        ("struct_size", C.c_uint32),
        ("features", C.c_uint32),
        ("n_envs", C.c_int32),
        ("n_elem", C.c_int32),
        ("n_substeps", C.c_int32),
        ("math_mode", C.c_int32),
        ("env_kind", C.c_int32),
        ("filter_order", C.c_int32),
        ("dt", C.c_double),
        ("final_time", C.c_double),
        ("base_length", C.c_double),
        ("base_radius", C.c_double),
        ("density", C.c_double),
        ("youngs_modulus", C.c_double),
        ("shear_modulus", C.c_double),
        ("gravity", C.c_double * 3),
        ("damping_constant", C.c_double),
        ("tip_force", C.c_double * 3),
        ("base_step", C.c_double),
        ("base_limit", C.c_double),
        ("alpha_c", C.c_double),
        ("eps_length", C.c_double),
        ("eps_rot_axis", C.c_double),
        ("acos_shift", C.c_double),
        ("eps_sin", C.c_double),
        ("time_two_half_adds", C.c_int32),
        ("damp_before_constrain", C.c_int32),
        ("contact_before_forcing", C.c_int32),
        ("damper_protocol", C.c_int32),
        ("plane_origin", C.c_double * 3),
        ("plane_normal", C.c_double * 3),
        ("contact_k", C.c_double),
        ("contact_nu", C.c_double),
        ("slip_velocity_tol", C.c_double),
        ("surface_tol", C.c_double),
        ("kinetic_mu", C.c_double * 3),
        ("static_mu", C.c_double * 3),
        ("control_penalty_coeff", C.c_double),
        ("target", C.c_double * 2),
        ("kappa_range", C.c_double * 2),
        ("kappa_rate_range", C.c_double * 2),
        ("n_arm", C.c_int32),
        ("n_knots", C.c_int32),
        ("head_radius", C.c_double),
        ("head_density", C.c_double),
        ("joint_k", C.c_double),
        ("joint_nu", C.c_double),
        ("joint_kt", C.c_double),
        ("n_ctrl", C.c_int32),
        ("n_spline_pieces", C.c_int32),
        ("muscle_torque_scale", C.c_double),
        ("max_activation_rate", C.c_double),
        ("arm_target", C.c_double * 3),
        ("n_suckers", C.c_int32),
        ("sucker_index", C.c_int32 * 4),
        ("reserved2", C.c_int32),
        ("sucker_reduction_ratio", C.c_double),
        ("n_muscles", C.c_int32),
        ("muscle_kind", C.c_int32 * 4),
        ("muscle_fl_degree", C.c_int32),
        ("muscle_equiv_load_form", C.c_int32),
        ("muscle_position_current_radius", C.c_int32),
        ("muscle_tm_length_law", C.c_int32),
        ("arm_push_mode", C.c_int32),
        ("head_fixed", C.c_int32),
        ("muscle_fl_coef", C.c_double * 8),
        ("head_center", C.c_double * 3),
        ("head_length", C.c_double),
        ("joint_angle0", C.c_double),
        ("joint_angle_step", C.c_double),
        ("damper_time_step", C.c_double),
    ]

    def copy(self) -> "SoftrodConfig":
        out = SoftrodConfig()
        C.memmove(C.byref(out), C.byref(self), C.sizeof(SoftrodConfig))
        return out


class SoftrodStateView(C.Structure):
    """Mirror of `struct softrod_state_view`."""

    _fields_ = [
        ("n_envs", C.c_int32),
        ("n_elem", C.c_int32),
        ("lane_stride", C.c_int32),
        ("arm_stride", C.c_int32),
        ("position", C.c_void_p),
        ("velocity", C.c_void_p),
        ("director", C.c_void_p),
        ("omega", C.c_void_p),
        ("tangents", C.c_void_p),
        ("time", C.c_void_p),
        ("control", C.c_void_p),
        ("kappa", C.c_void_p),
        ("rest_kappa", C.c_void_p),
        ("env_memory", C.c_void_p),
        ("prev_action", C.c_void_p),
        ("head", C.c_void_p),
        ("bc_targets", C.c_void_p),
        ("sucker_ratio", C.c_void_p),
        ("muscle_activation", C.c_void_p),
        ("sucker_index", C.c_void_p),
        ("material", C.c_void_p),
        ("env_aux", C.c_void_p),
        ("prev_kappa", C.c_void_p),
    ]


def _common(cfg: SoftrodConfig, n_envs, final_time, time_step, recording_fps, n_elems, math_mode):
    cfg.struct_size = C.sizeof(SoftrodConfig)
    cfg.n_envs = int(n_envs)
    cfg.n_elem = int(n_elems)
    cfg.n_substeps = int(1.0 / (recording_fps * time_step))  # soft_pendulum.py:78
    cfg.math_mode = int(math_mode)
    cfg.dt = float(time_step)
    cfg.final_time = float(final_time)
    cfg.alpha_c = 27.0 / 28.0
    cfg.eps_length = 1e-14
    cfg.eps_rot_axis = 1e-14
    cfg.acos_shift = 1e-10
    cfg.eps_sin = 1e-14
    cfg.time_two_half_adds = 1
    cfg.damp_before_constrain = 0  # constrain() precedes dampen() in build.py:81-113


def softpendulum_config(
    n_envs: int = 1,
    *,
    final_time: float = 5.0,
    time_step: float = 1.0e-4,
    recording_fps: int = 25,
    n_elems: int = 50,
    math_mode: int = MATH_FAST,
) -> SoftrodConfig:
    """Host-side equivalent of `softrod_config_softpendulum` with the constructor
    keywords of the reference's SoftPendulumEnv.

    Values: SoftPendulumEnv.__init__ (soft_pendulum.py:59-78) and
    build_soft_pendulum (build.py:18-26,87-113).  `shear_modulus` is not passed by
    build.py:54-61, so PyElastica's default applies: E / (2 (1 + 0.5)).
    """
    cfg = SoftrodConfig()
    _common(cfg, n_envs, final_time, time_step, recording_fps, n_elems, math_mode)
    cfg.features = FEATURES_SOFTPENDULUM
    cfg.env_kind = ENV_SOFTPENDULUM
    cfg.base_length = 1.0
    cfg.base_radius = 0.05
    cfg.density = 1000.0
    cfg.youngs_modulus = 1e6
    cfg.shear_modulus = 1e6 / (2.0 * (1.0 + 0.5))
    cfg.gravity[0], cfg.gravity[1], cfg.gravity[2] = 0.0, -9.80665, 0.0
    cfg.damping_constant = 2e-3
    return cfg


def softpendulum3d_config(
    n_envs: int = 1,
    *,
    final_time: float = 5.0,
    time_step: float = 1.0e-4,
    recording_fps: int = 25,
    n_elems: int = 50,
    math_mode: int = MATH_FAST,
) -> SoftrodConfig:
    """`softrod_config_softpendulum3d`: SoftPendulum3DEnv.__init__
    (soft_pendulum_3d.py:28-58) and build_soft_pendulum_3d
    (soft_pendulum_3d/build.py:43-86)."""
    cfg = SoftrodConfig()
    _common(cfg, n_envs, final_time, time_step, recording_fps, n_elems, math_mode)
    cfg.features = FEATURES_SOFTPENDULUM3D
    cfg.env_kind = ENV_SOFTPENDULUM3D
    cfg.filter_order = 7
    cfg.base_length = 1.0
    cfg.base_radius = 0.1
    cfg.density = 4000.0
    cfg.youngs_modulus = 1e6
    cfg.shear_modulus = 1e6 / (2.0 * (1.0 + 0.5))
    cfg.gravity[0], cfg.gravity[1], cfg.gravity[2] = 0.0, 0.0, -9.80665
    cfg.damping_constant = 1.0
    cfg.base_step = 1e-3
    cfg.base_limit = 0.5
    return cfg


def arm_single_config(
    n_envs: int = 1,
    *,
    final_time: float = 10.0,
    time_step: float = 7.0e-5,
    recording_fps: int = 20,
    n_elems: int = 50,
    control_penalty_coeff: float = 0.001,
    math_mode: int = MATH_FAST,
) -> SoftrodConfig:
    """`softrod_config_arm_single`: ArmSingleEnv.__init__ (octopus/arm_single_env.py:55-113)
    and build_arm (octopus/build.py:30-49,220-292)."""
    cfg = SoftrodConfig()
    _common(cfg, n_envs, final_time, time_step, recording_fps, n_elems, math_mode)
    cfg.features = FEATURES_ARM_SINGLE
    cfg.env_kind = ENV_ARM_SINGLE
    L0, r0 = 0.35, 0.35 * 0.02                      # octopus/build.py:46-49
    cfg.base_length = L0
    cfg.base_radius = r0
    cfg.density = 1000.0                            # :30-33
    cfg.youngs_modulus = 1e6
    cfg.shear_modulus = 1e6 / (2.0 * (1.0 + 0.5))
    g = -9.81                                       # :237
    cfg.gravity[0], cfg.gravity[1], cfg.gravity[2] = 0.0, 0.0, g
    cfg.damping_constant = 1e-2                     # :285
    cfg.plane_origin[0], cfg.plane_origin[1], cfg.plane_origin[2] = 0.0, 0.0, -r0   # :244
    cfg.plane_normal[0], cfg.plane_normal[1], cfg.plane_normal[2] = 0.0, 0.0, 1.0   # :233
    cfg.contact_k = 1e2
    cfg.contact_nu = 1e1
    cfg.slip_velocity_tol = 1e-8
    cfg.surface_tol = 1e-4
    period, froude = 2.0, 0.1
    mu = L0 / (period * period * abs(g) * froude)   # :247
    for i, f in enumerate((1.0, 1.5, 2.0)):          # friction_symmetry False, multiplier 1
        cfg.kinetic_mu[i] = mu * f
        cfg.static_mu[i] = 2 * (mu * f)
    cfg.control_penalty_coeff = float(control_penalty_coeff)
    cfg.target[0], cfg.target[1] = 1.0, 0.0         # arm_single_env.py:165
    cfg.kappa_range[0], cfg.kappa_range[1] = -49.33508476187419, 49.33545827754751
    cfg.kappa_rate_range[0], cfg.kappa_rate_range[1] = -21.063520620377012, 24.664591289161944
    return cfg


def octo_flat_config(
    n_envs: int = 1,
    *,
    final_time: float = 5.0,
    time_step: float = 7.0e-5,
    recording_fps: int = 5,
    n_elems: int = 10,
    n_arm: int = 8,
    n_action: int = 3,
    math_mode: int = MATH_FAST,
) -> SoftrodConfig:
    """FlatEnv.__init__ (octopus/flat_env.py:55-110) and build_octopus
    (octopus/build.py:30-217): per-arm rod constants as arm_single_config, plus the rigid
    head, the joints and the head boundary condition."""
    cfg = arm_single_config(n_envs, final_time=final_time, time_step=time_step,
                            recording_fps=recording_fps, n_elems=n_elems, math_mode=math_mode)
    cfg.features = FEATURES_OCTO_FLAT
    cfg.env_kind = ENV_OCTO_FLAT
    cfg.n_arm = int(n_arm)
    cfg.n_knots = int(n_action)
    cfg.head_radius = 0.04
    cfg.head_density = 700.0
    cfg.joint_k = 1e6
    cfg.joint_nu = 1e-3
    cfg.joint_kt = 1e0
    # Cylinder(start = (0, 0, -r0), direction = e_z, normal = e_y, base_length = 2 r0) and the joints' angles
    # 360 / n_arm * arm_i (octopus/build.py:73-74,95-105,117-132)
    cfg.head_center[0], cfg.head_center[1], cfg.head_center[2] = 0.0, 0.0, -cfg.base_radius + 2.0 * cfg.base_radius / 2
    cfg.head_length = 2.0 * cfg.base_radius
    cfg.joint_angle0, cfg.joint_angle_step = 0.0, 360 / int(n_arm)
    return cfg


def soft_arm_config(n_envs: int = 1, *, n_elems: int = 40, math_mode: int = MATH_FAST) -> SoftrodConfig:
    """`softrod_config_soft_arm`: SoftArmTrackingEnv.__init__ (soft_arm/soft_arm_tracking.py:
    107-158) and the simulator its reset builds (:261-383), game_mode 1 (the registered
    SoftArmTracking-v0).  Lengths are in millimetres there (base_length 1000, radius 50,
    density 1000e-6)."""
    cfg = SoftrodConfig()
    sim_dt, rl_interval = 2.0e-4, 0.01                       # :117-118
    _common(cfg, n_envs, 5.0, sim_dt, 1, n_elems, math_mode)  # max_episode_final_time = 5  :127
    cfg.n_substeps = int(np_rint(rl_interval / sim_dt))      # num_steps_per_update, :119-121
    cfg.features = FEATURES_SOFT_ARM
    cfg.env_kind = ENV_SOFT_ARM
    cfg.base_length = 1000.0                                 # :129
    cfg.base_radius = 50.0                                   # :130
    cfg.density = 1000 * 1e-6                                # :280
    cfg.youngs_modulus = 2e6                                 # :122
    cfg.shear_modulus = 2e6 / (2.0 * (1.0 + 0.5))            # not passed at :272-283: PyElastica's default
    cfg.damping_constant = 2e6 * 1e-7 * 1                    # :269
    cfg.n_ctrl = 4                                           # :133
    cfg.n_spline_pieces = 3
    cfg.muscle_torque_scale = 10 * 50.0 * 2e6                # alpha, :350
    cfg.max_activation_rate = float("inf")                   # :143
    cfg.arm_target[0], cfg.arm_target[1], cfg.arm_target[2] = 500.0, 500.0, 500.0   # :147
    return cfg


# COOMM's force-length law as the paper prints it (Chang et al. 2023, section 2(c)):
# f_l(l) = max{3.06 l^3 - 13.64 l^2 + 18.01 l - 6.44, 0}; ascending powers
COOMM_FL_COEF = (-6.44, 18.01, -13.64, 3.06)


def muscle_defaults(cfg: SoftrodConfig) -> None:
    """The recalled COOMM details as softrod_config switches (include/softrod.h, PARITY UNPINNED) and the
    three layers of create_es_muscle_layers (octopus/build.py:295-338): two longitudinal, one transverse."""
    cfg.n_muscles = 3
    cfg.muscle_kind[0], cfg.muscle_kind[1], cfg.muscle_kind[2] = MUSCLE_LONGITUDINAL, MUSCLE_LONGITUDINAL, MUSCLE_TRANSVERSE
    cfg.muscle_fl_degree = len(COOMM_FL_COEF) - 1
    for k, v in enumerate(COOMM_FL_COEF):
        cfg.muscle_fl_coef[k] = v
    cfg.muscle_equiv_load_form = 0
    cfg.muscle_position_current_radius = 1
    cfg.muscle_tm_length_law = 0


def es_muscle_layers(radius_mean, radius_base: float, init_angle_rotates: bool = True, tm_sign: float = -1.0):
    """(ratio_position [3][3][n], strength [3][n]) for softrod_set_muscle_layers: what
    create_es_muscle_layers(radius_mean, radius_base) (octopus/build.py:295-338) hands to COOMM's constructors,
      LongitudinalMuscle(muscle_init_angle=+pi/2, ratio_muscle_position=(0, -6/9, 0), rest_muscle_area=(r/r_base)^2, max_muscle_stress=0.5)
      LongitudinalMuscle(muscle_init_angle=-pi/2, ...same...)
      TransverseMuscle(rest_muscle_area=(r/r_base)^2, max_muscle_stress=1.0)
    turned into the tables the kernels read.  RECALLED COOMM behaviour, each a keyword here:
      init_angle_rotates  the longitudinal muscle's position is ratio_muscle_position turned about d3 by
                          muscle_init_angle ((0, -2/3) -> (+2/3, 0) and (-2/3, 0): an antagonistic pair on d1);
                          False: the array as given (both muscles at (0, -2/3, 0))
      tm_sign             TransverseMuscle passes -max_muscle_stress: contraction of the radial fibres
                          EXTENDS the arm (the force along the tangent is negative)."""
    import numpy as np

    r = np.asarray(radius_mean, np.float64)
    n = r.size
    area = (r / radius_base) ** 2
    base = np.stack((np.zeros_like(r), -6 / 9 * np.ones_like(r), np.zeros_like(r)), axis=0)
    ratio = np.zeros((3, 3, n))
    for m, ang in enumerate((np.pi / 2, -np.pi / 2)):
        if init_angle_rotates:
            c, s = np.cos(ang), np.sin(ang)
            ratio[m, 0] = c * base[0] - s * base[1]
            ratio[m, 1] = s * base[0] + c * base[1]
            ratio[m, 2] = base[2]
        else:
            ratio[m] = base
    strength = np.stack((0.5 * area, 0.5 * area, tm_sign * 1.0 * area), axis=0)
    return np.ascontiguousarray(ratio), np.ascontiguousarray(strength)


def arm_push_radii(n_elems: int = 40, radius_base: float = 0.012, radius_tip: float = 0.001):
    """radius_mean of ArmPushEnv._build (octopus/arm_push_env.py:160-165)."""
    import numpy as np

    radius = np.linspace(radius_base, radius_tip, n_elems + 1)
    return (radius[:-1] + radius[1:]) / 2


def arm_push_config(
    n_envs: int = 1,
    *,
    final_time: float = 2.5,
    time_step: float = 5.0e-5,
    recording_fps: int = 40,
    mode: str = "discrete",
    math_mode: int = MATH_FAST,
) -> SoftrodConfig:
    """`softrod_config_arm_push`: ArmPushEnv.__init__ (octopus/arm_push_env.py:65-139) and `_build`
    (:158-224): a 40-element arm of length 0.2, density 700, E = 1e4, G = E / 1.5, tapered 12:1
    (arm_push_radii -> softrod_set_radius_profile), AnalyticalLinearDamper(0.05 * 2 * 1e2), one
    ControllableFixConstraint at index 0 and ApplyMuscles over create_es_muscle_layers.  No gravity, no plane."""
    if mode not in ("discrete", "continuous"):
        raise NotImplementedError(f"The mode {mode} is not available.")            # arm_push_env.py:97
    cfg = SoftrodConfig()
    _common(cfg, n_envs, final_time, time_step, recording_fps, 40, math_mode)
    cfg.features = FEATURES_ARM_PUSH
    cfg.env_kind = ENV_ARM_PUSH
    cfg.arm_push_mode = 0 if mode == "discrete" else 1
    cfg.base_length = 0.2                           # L0, :160
    cfg.base_radius = 0.012                         # radius_base (the profile overrides it per element)
    cfg.density = 700.0                             # :174
    cfg.youngs_modulus = 1e4                        # :175
    cfg.shear_modulus = 1e4 / 1.5                   # :176
    cfg.damping_constant = 0.05 * 2 * 1e2           # damp_coefficient * 1e2, :166,183
    cfg.n_suckers = 1                               # :187-195
    cfg.sucker_index[0] = 0
    cfg.sucker_reduction_ratio = 1.0                # SuckerController default (controllable_constraint.py:11)
    # _build registers dampen() BEFORE constrain() (:180-195; tests/golden/ref_muscle_build_records.json "order"):
    # under the registration-order rule the damper runs first here (the other builds register constrain() first)
    cfg.damp_before_constrain = 1
    muscle_defaults(cfg)
    return cfg


def arm_pull_weight_config(
    n_envs: int = 1,
    *,
    final_time: float = 2.5,
    recording_fps: int = 40,
    mode: str = "continuous",
    math_mode: int = MATH_FAST,
) -> SoftrodConfig:
    """`softrod_config_arm_pull_weight`: ArmPullWeightEnv (octopus/arm_push_env.py:516-618) — ArmPushEnv with
    time_step 2.5e-5 (:518), damper 0.05 * 2 * 5e2 (:549), a rigid Cylinder "weight" (:552-567) held by
    BodyBoundaryCondition (:569-575) and joined to the arm's node 0 by FixedJoint2Rigid(k=1e6, nu=1e-2, kt=1, angle=0,
    radius=0.015) (:577-589), the sucker at reduction_ratio 0.9 (:591-599)."""
    cfg = arm_push_config(n_envs, final_time=final_time, time_step=2.5e-5, recording_fps=recording_fps, mode=mode,
                          math_mode=math_mode)
    cfg.features = FEATURES_ARM_PULL_WEIGHT
    cfg.env_kind = ENV_ARM_PULL_WEIGHT
    cfg.damping_constant = 0.05 * 2 * 5e2
    cfg.sucker_reduction_ratio = 0.9
    cfg.n_arm, cfg.n_knots = 1, 1
    rigid_rod_radius, radius_base = 0.015, 0.012
    cfg.head_radius = rigid_rod_radius
    cfg.head_density = 700 * 1.0
    cfg.head_length = radius_base * 2
    # start = (-0.9 * radius, 0, -2 * radius_base), direction e_z: centre = start + direction * length / 2
    cfg.head_center[0], cfg.head_center[1], cfg.head_center[2] = -rigid_rod_radius * 0.9, 0.0, -2 * radius_base + radius_base * 2 / 2
    cfg.joint_k, cfg.joint_nu, cfg.joint_kt = 1e6, 1e-2, 1e0
    cfg.joint_angle0, cfg.joint_angle_step = 0.0, 0.0
    return cfg


MUSCLE_OCTOPUS = {
    # build_muscle_octopus.py:26-47 ARM_MATERIAL / DEFAULT_SCALE_LENGTH / HEAD_PROPERTIES
    "density": 1000.0, "youngs_modulus": 1.5e4, "shear_modulus": 1.5e4 / (1.0 + 0.5), "damping_constant": 0.20,
    "nu_scale": 1e-2, "base_length": 0.25, "base_radius": 0.013, "tip_radius": 0.0042, "head_radius": 0.04,
    "head_density": 50.0, "body_arm_k": 1e6, "body_arm_kt": 1e2, "body_arm_nu": 1e-3, "damper_time_step": 7e-5,
}


def muscle_octopus_radii(n_elem: int = 20):
    """build_arm's `base_radius=np.linspace(base_radius, tip_radius, n_elem)` (build_muscle_octopus.py:60-62): one value
    PER ELEMENT (ArmPushEnv averages n + 1 node values instead)."""
    import numpy as np

    return np.linspace(MUSCLE_OCTOPUS["base_radius"], MUSCLE_OCTOPUS["tip_radius"], n_elem)


def muscle_octopus_arm_frames(env_kind: int, head_radius: float = 0.04):
    """Arm start points, directions and joint angles of build_octopus_muscles (8 arms at 22.5 + 45 i degrees,
    build_muscle_octopus.py:83-93) / build_two_arms (2 arms at 90 + 180 i, :200-210)."""
    import numpy as np
    from scipy.spatial.transform import Rotation as Rot

    if int(env_kind) == ENV_ARM_TWO:
        angles = [90.0 + 180.0 * arm_i for arm_i in range(2)]
    else:
        angles = [45.0 / 2 + 45 * arm_i for arm_i in range(8)]
    pos, dirs = [], []
    for angle in angles:
        rot = Rot.from_euler("z", angle, degrees=True)
        pos.append(rot.apply([head_radius, 0.0, 0.0]))
        dirs.append(rot.apply([1.0, 0.0, 0.0]))
    return np.ascontiguousarray(pos, np.float64), np.ascontiguousarray(dirs, np.float64), angles


def muscle_octopus_rest_length_sum(n_elem: int = 20, head_radius: float = 0.04) -> float:
    """`sum(self.shearable_rods[0].rest_lengths)` of ReachEnv.reset (reach_env.py:141-143) as straight_rod computes the
    rest lengths of arm 0: per-coordinate linspace from start to start + direction * base_length, norms of the differences
    (sum of the three squares in order), added up by Python's sum()."""
    import numpy as np

    pos, dirs, _ = muscle_octopus_arm_frames(ENV_REACH, head_radius)
    start = pos[0]
    end = start + dirs[0] * MUSCLE_OCTOPUS["base_length"]
    position = np.zeros((3, n_elem + 1))
    for i in range(3):
        position[i, ...] = np.linspace(start[i], end[i], n_elem + 1)
    d = position[..., 1:] - position[..., :-1]
    rest = np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2])
    return sum(rest)


def arm_two_activation_basis(n_elems: int = 20, n_sucker: int = 3):
    """W [n_elems][n_sucker] with apply_X = W @ activation: ArmTwoEnv.set_action's
    `interp1d(control_location, [0] + list(activation) + [0], kind="cubic")(range(n_elems))` (arm_two_env.py:237-245)
    applied to unit vectors; control_location = [0] + sucker_location + [n_elems - 1] (:77-81)."""
    import numpy as np
    from scipy.interpolate import interp1d

    sucker_location = [n_elems // (n_sucker * 2) * (2 * i + 1) for i in range(n_sucker)]
    control_location = [0] + sucker_location + [n_elems - 1]
    cols = [interp1d(control_location, [0] + list(np.eye(n_sucker)[j]) + [0], kind="cubic")(range(n_elems)) for j in range(n_sucker)]
    return np.ascontiguousarray(np.stack(cols, axis=1), np.float64), sucker_location


def muscle_octopus_config(
    env_kind: int,
    n_envs: int = 1,
    *,
    final_time: float = None,
    time_step: float = 5.0e-5,
    recording_fps: int = 25,
    n_elems: int = 20,
    math_mode: int = MATH_FAST,
) -> SoftrodConfig:
    """`softrod_config_muscle_octopus`: CrawlEnv / ArmTwoEnv / ReachEnv.__init__ (crawl_env.py:61-125,
    arm_two_env.py:55-113, reach_env.py:53-106) over build_octopus_muscles / build_two_arms
    (build_muscle_octopus.py:66-179,182-291).  No gravity, no plane: the arms float, held by their suckers."""
    env_kind = int(env_kind)
    if env_kind not in MUSCLE_OCTOPUS_ENVS:
        raise ValueError("env_kind must be ENV_CRAWL, ENV_ARM_TWO or ENV_REACH")
    if final_time is None:
        final_time = {ENV_CRAWL: 10.0, ENV_ARM_TWO: 5.0, ENV_REACH: 5.0}[env_kind]
    m = MUSCLE_OCTOPUS
    cfg = SoftrodConfig()
    _common(cfg, n_envs, final_time, time_step, recording_fps, n_elems, math_mode)
    cfg.features = FEATURES_ARM_PULL_WEIGHT
    cfg.env_kind = env_kind
    cfg.arm_push_mode = 1
    cfg.base_length = m["base_length"]
    cfg.base_radius = m["base_radius"]                       # r0: the head's geometry; the arms' radii are a profile
    cfg.density = m["density"]
    cfg.youngs_modulus = m["youngs_modulus"]
    cfg.shear_modulus = m["shear_modulus"]
    cfg.damping_constant = m["damping_constant"] * m["nu_scale"]
    cfg.damper_time_step = m["damper_time_step"]
    cfg.n_arm = 2 if env_kind == ENV_ARM_TWO else 8
    cfg.n_knots = {ENV_CRAWL: 3, ENV_ARM_TWO: 9, ENV_REACH: 3 * int(n_elems)}[env_kind]       # actions per arm
    cfg.head_radius = m["head_radius"]
    cfg.head_density = m["head_density"]
    r0 = m["base_radius"]
    cfg.head_length = r0 * 2
    cfg.head_center[0], cfg.head_center[1], cfg.head_center[2] = 0.0, 0.0, -r0 * 2 + r0 * 2 / 2
    cfg.head_fixed = 1 if env_kind == ENV_REACH else 0
    cfg.joint_k, cfg.joint_nu, cfg.joint_kt = m["body_arm_k"], m["body_arm_nu"], m["body_arm_kt"]
    if env_kind == ENV_ARM_TWO:
        cfg.joint_angle0, cfg.joint_angle_step = 90.0, 180.0
        cfg.n_suckers = 3
        _, loc = arm_two_activation_basis(int(n_elems), 3)
        for j in range(3):
            cfg.sucker_index[j] = loc[j]
    else:
        cfg.joint_angle0, cfg.joint_angle_step = 45.0 / 2, 45.0
        cfg.n_suckers = 1 if env_kind == ENV_CRAWL else 0
        cfg.sucker_index[0] = 0
    cfg.sucker_reduction_ratio = 1.0
    # builds register dampen() first, the envs constrain() the arms afterwards (build_muscle_octopus.py:101-106 before
    # crawl_env.py:148-155)
    cfg.damp_before_constrain = 1
    muscle_defaults(cfg)
    return cfg


def np_rint(x: float) -> float:
    import numpy as np

    return float(np.rint(x))


def spline_table(base_length: float, n_ctrl: int):
    """(breaks, coef) for softrod_set_spline_table: the cubic interpolating spline of
    muscle_torques_with_bspline.py:150-152 — `make_interp_spline` through n_ctrl + 2 equidistant
    points whose two end values are zero — as the piecewise-cubic form of its n_ctrl cardinal
    functions (scipy BSpline -> PPoly)."""
    import numpy as np
    from scipy.interpolate import PPoly, make_interp_spline

    x = np.linspace(0.0, base_length, n_ctrl + 2)
    breaks, coef = None, None
    for j in range(n_ctrl):
        y = np.zeros(n_ctrl + 2)
        y[1 + j] = 1.0
        pp = PPoly.from_spline(make_interp_spline(x, y))
        keep = np.nonzero(np.diff(pp.x) > 0)[0]               # drop the zero-length end pieces
        if coef is None:
            breaks = np.concatenate([pp.x[keep], pp.x[keep[-1] + 1:keep[-1] + 2]])
            coef = np.zeros((len(keep), n_ctrl, 4))
        coef[:, j, :] = pp.c[::-1, keep].T                    # ascending powers of (s - breaks[p])
    return np.ascontiguousarray(breaks, np.float64), np.ascontiguousarray(coef, np.float64)


def action_basis(n_elems: int, n_action: int = 7):
    """W with rest_kappa[0,:] = W @ action: set_action's cubic `interp1d`
    (octopus/arm_single_env.py:226-235) applied to unit vectors."""
    import numpy as np
    from scipy.interpolate import interp1d

    x = np.linspace(0, 1, n_action)
    xs = np.linspace(0, 1, n_elems - 1)
    cols = [interp1d(x, np.eye(n_action)[j], kind="cubic", axis=-1)(xs) for j in range(n_action)]
    return np.ascontiguousarray(np.stack(cols, axis=1), dtype=np.float64)


def octo_action_basis(n_elems: int, n_knots: int = 3):
    """W with rest_kappa[0,:] = W @ knots for FlatEnv.set_action (octopus/flat_env.py:288-311):
    cubic `interp1d` through the knots padded with one zero at each end."""
    import numpy as np
    from scipy.interpolate import interp1d

    x = np.linspace(0, 1, n_knots + 2)
    xs = np.linspace(0, 1, n_elems - 1)
    eye = np.concatenate([np.zeros((n_knots, 1)), np.eye(n_knots), np.zeros((n_knots, 1))], axis=-1)
    w = interp1d(x, eye, kind="cubic", axis=-1)(xs)          # (n_knots, n_elems-1)
    return np.ascontiguousarray(w.T, dtype=np.float64)


def octo_arm_frames(n_arm: int, head_radius: float):
    """Arm start points and directions of build_octopus (octopus/build.py:73-80)."""
    import numpy as np
    from scipy.spatial.transform import Rotation as Rot

    rotation_angle = 360 / n_arm
    pos, dirs = [], []
    for arm_i in range(n_arm):
        rot = Rot.from_euler("z", rotation_angle * arm_i, degrees=True)
        pos.append(rot.apply([head_radius, 0.0, 0.0]))
        dirs.append(rot.apply([1.0, 0.0, 0.0]))
    return np.ascontiguousarray(pos, np.float64), np.ascontiguousarray(dirs, np.float64)


def config_action_dim(cfg: "SoftrodConfig") -> int:
    if int(cfg.env_kind) == ENV_OCTO_FLAT:
        return int(cfg.n_arm) * int(cfg.n_knots)
    if int(cfg.env_kind) in (ENV_ARM_PUSH, ENV_ARM_PULL_WEIGHT):
        return 1 if int(cfg.arm_push_mode) == 0 else 2      # Discrete(2) index / (location, activation)
    if int(cfg.env_kind) in MUSCLE_OCTOPUS_ENVS:
        return int(cfg.n_arm) * int(cfg.n_knots)
    return action_dim(cfg.env_kind)


def config_obs_dim(cfg: "SoftrodConfig") -> int:
    if int(cfg.env_kind) in (ENV_ARM_PUSH, ENV_ARM_PULL_WEIGHT):
        return 2 * (int(cfg.n_elem) + 1) + 2                # arm_push_env.py:104,118-120
    if int(cfg.env_kind) == ENV_OCTO_FLAT:
        n = int(cfg.n_elem)
        return int(cfg.n_arm) * ((n - 1) + 4 * (n + 1) + int(cfg.n_knots)) + 13
    if int(cfg.env_kind) in MUSCLE_OCTOPUS_ENVS:
        n, na, nk = int(cfg.n_elem), int(cfg.n_arm), int(cfg.n_knots)
        if int(cfg.env_kind) == ENV_ARM_TWO:
            return na * ((n - 1) * 2 + nk + na + 3)                    # arm_two_env.py:92-95
        shared = 17 if int(cfg.env_kind) == ENV_CRAWL else 18          # crawl_env.py:91-101, reach_env.py:87-91
        return na * ((n - 1) + (n + 1) * 4 + nk + na + shared)
    return obs_dim(cfg.env_kind)


class SoftrodError(RuntimeError):
    pass


_VP = C.c_void_p
_EXPORTS = {
    # name: (restype, argtypes)
    "softrod_abi_version": (C.c_int, []),
    "softrod_source_hash": (C.c_char_p, []),
    "softrod_action_dim": (C.c_int, [C.c_int]),
    "softrod_obs_dim": (C.c_int, [C.c_int]),
    "softrod_aux_dim": (C.c_int, [C.c_int]),
    "softrod_config_softpendulum": (C.c_int, [C.POINTER(SoftrodConfig), C.c_int]),
    "softrod_config_softpendulum3d": (C.c_int, [C.POINTER(SoftrodConfig), C.c_int]),
    "softrod_config_arm_single": (C.c_int, [C.POINTER(SoftrodConfig), C.c_int]),
    "softrod_config_octo_flat": (C.c_int, [C.POINTER(SoftrodConfig), C.c_int]),
    "softrod_config_soft_arm": (C.c_int, [C.POINTER(SoftrodConfig), C.c_int]),
    "softrod_config_arm_push": (C.c_int, [C.POINTER(SoftrodConfig), C.c_int, C.c_int]),
    "softrod_config_arm_pull_weight": (C.c_int, [C.POINTER(SoftrodConfig), C.c_int]),
    "softrod_config_muscle_octopus": (C.c_int, [C.POINTER(SoftrodConfig), C.c_int, C.c_int]),
    "softrod_set_muscle_layers": (C.c_int, [_VP, _VP, _VP]),
    "softrod_set_spline_table": (C.c_int, [_VP, _VP, _VP]),
    "softrod_config_action_dim": (C.c_int, [C.POINTER(SoftrodConfig)]),
    "softrod_config_obs_dim": (C.c_int, [C.POINTER(SoftrodConfig)]),
    "softrod_reset_octo": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP]),
    "softrod_autoreset_enable": (C.c_int, [_VP, C.c_int]),
    "softrod_queue_push": (C.c_int, [_VP, _VP, _VP, C.c_int, _VP]),
    "softrod_queue_push_straight": (C.c_int, [_VP, _VP, _VP, _VP, _VP, C.c_int, _VP]),
    "softrod_queue_push_octo": (C.c_int, [_VP, _VP, _VP, _VP, _VP, C.c_int, _VP]),
    "softrod_queue_status": (C.c_int, [_VP, _VP, _VP, _VP]),
    "softrod_queue_status_begin": (C.c_int, [_VP, _VP]),
    "softrod_queue_status_poll": (C.c_int, [_VP, C.c_int, _VP, _VP]),
    "softrod_queue_advance": (C.c_int, [_VP, _VP, _VP]),
    "softrod_set_action_basis": (C.c_int, [_VP, _VP]),
    "softrod_set_radius_profile": (C.c_int, [_VP, _VP]),
    "softrod_create": (C.c_int, [C.POINTER(SoftrodConfig), C.c_int, C.POINTER(C.c_void_p)]),
    "softrod_reset": (C.c_int, [_VP, _VP, _VP, _VP]),
    "softrod_reset_straight": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP]),
    "softrod_step": (C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "softrod_step_packed": (C.c_int, [_VP, _VP, _VP, _VP, _VP]),
    "softrod_scatter_rows": (C.c_int, [_VP, _VP, _VP, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_uint32, _VP]),
    "softrod_exchange_alloc": (C.c_int, [C.c_int, C.c_uint64, C.POINTER(C.c_void_p), _VP, C.POINTER(C.c_int)]),
    "softrod_exchange_open": (C.c_int, [C.c_int, _VP, C.c_int, C.POINTER(C.c_void_p)]),
    "softrod_exchange_close": (C.c_int, [C.c_int, _VP]),
    "softrod_exchange_free": (C.c_int, [C.c_int, _VP]),
    "softrod_observe": (C.c_int, [_VP, _VP, _VP, _VP]),
    "softrod_substeps": (C.c_int, [_VP, _VP, C.c_int, _VP]),
    "softrod_state_view_get": (C.c_int, [_VP, C.POINTER(SoftrodStateView)]),
    "softrod_set_timing": (C.c_int, [_VP, C.c_int]),
    "softrod_kernel_times_ms": (C.c_int, [_VP, _VP, C.c_int, C.POINTER(C.c_int)]),
    "softrod_last_kernel_ms": (C.c_int, [_VP, C.POINTER(C.c_float)]),
    "softrod_kernel_tier": (C.c_char_p, [_VP]),
    "softrod_last_error": (C.c_char_p, [_VP]),
    "softrod_destroy": (C.c_int, [_VP]),
}

EXPORTED_SYMBOLS = tuple(_EXPORTS)

_lib = None


def library_path() -> Path:
    env = os.environ.get("SOFTROD_HIP_LIB")
    if env:
        return Path(env)
    return Path(__file__).resolve().parent / "csrc" / "libsoftrod_hip.so"


def load_library() -> C.CDLL:
    """dlopen libsoftrod_hip.so and type its entry points.  Raises if absent."""
    global _lib
    if _lib is not None:
        return _lib
    # torch bundles its own libamdhip64.so.7 (same soname as /opt/rocm's).  The first one
    # mapped wins symbol resolution, and a process with BOTH runtimes breaks (streams and
    # device state are per runtime; observed on the MI355X box as hipGetDeviceCount
    # failing).  The Python host path always runs next to torch, so make sure torch's
    # runtime is the one libsoftrod_hip.so binds to.  A pure C/C++ host never imports
    # torch and binds to the system ROCm runtime instead.
    try:
        import torch  # noqa: F401
    except ImportError:  # pragma: no cover
        pass
    path = library_path()
    if not path.exists():
        raise SoftrodError(
            f"{path} not found: build it with `python __graft_entry__.py build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback."
        )
    lib = C.CDLL(str(path))
    for name, (restype, argtypes) in _EXPORTS.items():
        fn = getattr(lib, name)  # AttributeError if a declared symbol is missing
        fn.restype = restype
        fn.argtypes = argtypes
    if lib.softrod_abi_version() != ABI_VERSION:
        raise SoftrodError("libsoftrod_hip.so ABI version mismatch")
    _lib = lib
    return lib


def library_source_hash() -> str:
    """The source hash the LOADED library was built from (softrod_source_hash, include/softrod.h)."""
    return load_library().softrod_source_hash().decode()


def source_hash() -> str:
    """The same hash taken over the sources on disk (csrc/Makefile's SOURCE_HASH): differs from
    library_source_hash() when the .so is older than the sources."""
    import hashlib

    csrc = Path(__file__).resolve().parent / "csrc"
    files = sorted(csrc.glob("*.hpp"), key=lambda f: f.name) + [csrc / "softrod_capi.hip",
                                                                 csrc.parents[1] / "include" / "softrod.h"]
    h = hashlib.sha256()
    for f in files:
        h.update(f.read_bytes())
    return h.hexdigest()[:16]


def check(rc: int, handle=None) -> None:
    if rc == 0:
        return
    msg = ""
    if _lib is not None:
        raw = _lib.softrod_last_error(handle)  # NULL handle -> this thread's last error
        msg = raw.decode() if raw else ""
    raise SoftrodError(f"softrod call failed with code {rc}: {msg}")
