/*
 * softrod.h — C-ABI of the MI355X-native batched Cosserat-rod stepper.
 *
 * This is the drop-in boundary for the ONE hot path of skim0119/gym-softrobot
 * that BASELINE.json names: `env.step()` of SoftPendulum-v0, i.e.
 *
 *     for _ in range(self.step_skip):
 *         self.time = self.do_step(self.simulator, self.time, self.time_step)
 *                                   (gym_softrobot/envs/soft_pendulum/soft_pendulum.py:183-184)
 *
 * where `do_step` is PyElastica's `PositionVerlet().step` (soft_pendulum.py:137-139)
 * applied to the simulator assembled by `build_soft_pendulum`
 * (gym_softrobot/envs/soft_pendulum/build.py:29-115), followed by the NaN check,
 * reward, truncation test and observation of soft_pendulum.py:196-251 — and for
 * the identical loops of the envs SURVEY.md §8 lists next to it
 * (soft_pendulum_3d.py:127-128, octopus/arm_single_env.py:247-248).
 *
 * The reference has no FFI for this path: its "operator API" is a set of Python
 * classes PyElastica calls back into once per substep (ConstraintBase /
 * NoForces subclasses, build.py:65-79,94-101) plus the registration DSL
 * (build.py:62,81-113).  Those per-substep Python callbacks ARE the bottleneck,
 * so this boundary replaces them with compiled-in, bit-selected features.  Each
 * entry point below cites the reference interface it stands in for.
 *
 * Conventions: every function returns 0 on success or a negative
 * SOFTROD_E* code and never throws; plain pointers and sizes only (no torch
 * types).  `stream` is a hipStream_t passed as void* (NULL = default stream).
 * Pointers documented "device" must be device-accessible; "host" must be host
 * memory.  One handle = one device; a handle is not re-entrant, independent
 * handles are thread-safe.
 *
 * The same declarations are implemented by the HIP library
 * (gym_softrobot_amd/csrc → libsoftrod_hip.so).  The CPU oracle
 * (oracle/softrod_oracle.c) shares only `softrod_config` and is test
 * infrastructure, not a fallback.
 */
#ifndef SOFTROD_H
#define SOFTROD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SOFTROD_ABI_VERSION 16

/* error codes */
#define SOFTROD_OK 0
#define SOFTROD_EINVAL (-1)   /* bad argument / unsupported configuration   */
#define SOFTROD_ENOMEM (-2)   /* device or host allocation failed           */
#define SOFTROD_EHIP (-3)     /* a HIP runtime call failed (see last_error) */
#define SOFTROD_ENODEV (-4)   /* no usable gfx950 device                    */

/*
 * Feature bits: the compiled-in replacements of the reference's per-substep
 * Python hooks.  Order of application inside a substep is fixed to the order
 * PyElastica's PositionVerlet.step gives them (DESIGN.md "substep order").
 * File names below are relative to gym_softrobot/envs/.
 */
enum softrod_feature {
    /* GravityForces(acc_gravity)                  soft_pendulum/build.py:88-91 */
    SOFTROD_FEAT_GRAVITY = 1u << 0,
    /* PendulumPointForces.apply_forces: external_forces[0,0] = action
     * (ASSIGNS, after gravity was added)          soft_pendulum/build.py:94-105 */
    SOFTROD_FEAT_POINT_FORCE_NODE0_X = 1u << 1,
    /* PendulumBoundaryConditions                  soft_pendulum/build.py:65-85 */
    SOFTROD_FEAT_PENDULUM_BC = 1u << 2,
    /* AnalyticalLinearDamper(damping_constant, time_step)
     *                                             soft_pendulum/build.py:108-113 */
    SOFTROD_FEAT_ANALYTICAL_DAMPER = 1u << 3,
    /* PyElastica OneEndFixedBC on node 0 / element 0 (known-answer tests)      */
    SOFTROD_FEAT_FIXED_BC = 1u << 4,
    /* constant force on the last node, PyElastica EndpointForces with
     * start_force = 0 and no ramp (known-answer tests)                          */
    SOFTROD_FEAT_TIP_FORCE = 1u << 5,
    /* MovingBaseConstraint: node 0 pinned to the commanded base position,
     * element 0 director fixed, base velocity imposed
     *                                             soft_pendulum_3d/build.py:23-40 */
    SOFTROD_FEAT_MOVING_BASE_BC = 1u << 6,
    /* LaplaceDissipationFilter(filter_order)      soft_pendulum_3d/build.py:82-85 */
    SOFTROD_FEAT_LAPLACE_FILTER = 1u << 7,
    /* Plane + RodPlaneContactWithAnisotropicFriction(k, nu, slip_velocity_tol,
     * static_mu_array, kinetic_mu_array)          octopus/build.py:236-283    */
    SOFTROD_FEAT_PLANE_CONTACT_ANISO = 1u << 8,
    /* set_action: rest_kappa[0, :] = cubic interp1d of the action
     *                                             octopus/arm_single_env.py:226-235 */
    SOFTROD_FEAT_REST_KAPPA_ACTION = 1u << 9,
    /* OctoFlat: n_arm rods per env joined to a rigid Cylinder head by
     * FixedJoint2Rigid, head held by BodyBoundaryCondition
     *   octopus/build.py:52-132, utils/custom_elastica/joint.py:20-225,
     *   utils/custom_elastica/constraint.py:8-85                               */
    SOFTROD_FEAT_OCTO_HEAD = 1u << 10,
    /* two MuscleTorquesWithVaryingBetaSplines ("normal", "binormal"): control points from the
     * action, rate-limited, a cubic interpolating spline through them evaluated at the
     * cumulative element lengths whenever the points changed, added to external_torques
     *   soft_arm/soft_arm_tracking.py:352-383,
     *   utils/custom_elastica/muscle_torque/muscle_torques_with_bspline.py:128-225       */
    SOFTROD_FEAT_SPLINE_MUSCLE_TORQUES = 1u << 11,
    /* ControllableFixConstraint ("sucker"): constrain_rates scales the velocity of node `index`
     * and the angular velocity of element `index` by (1 - reduction_ratio) while the
     * SuckerController's flag is on; constrain_values is a no-op
     *   octopus/controllable_constraint.py:24-69; registered at octopus/arm_push_env.py:188-195,
     *   587-595, arm_two_env.py:137-152 (several per arm, ratio set from the action :228),
     *   crawl_env.py:148-161.  Up to SOFTROD_MAX_SUCKERS per rod (softrod_config.sucker_index);
     *   the per-env effective ratio lives in softrod_state_view.sucker_ratio.                */
    SOFTROD_FEAT_SUCKER_CONSTRAINT = 1u << 12,
    /* ApplyMuscles(muscles=[LongitudinalMuscle, ..., TransverseMuscle]) of COOMM, registered as a forcing at
     *   octopus/arm_push_env.py:197-212,596-604, build_muscle_octopus.py:160-172,270-282 with the layers of
     *   create_es_muscle_layers (octopus/build.py:295-338).
     * COOMM (git pin uv.lock:173-175, branch refactor-numba-hotloops) is NOT on disk and not installable:
     * what this bit computes restates the published model (Chang, Halder, Gribkova, Tekinalp, Naughton,
     * Gazzola, Mehta: "Energy-shaping control of a muscular octopus arm moving in three dimensions",
     * Proc. R. Soc. A 479:20220593, 2023, section 2(c)) in the operation order recalled from
     * coomm/actuations/muscles/muscle.py — PARITY UNPINNED.  Per substep, per muscle m, per element:
     *   x_m   = radius * ratio_position_m                  muscle position in the material frame
     *   nu_m  = (sigma + e_3) + average(kappa) x x_m       muscle strain; l_m = |nu_m|, t_m = nu_m / l_m
     *   F_m   = activation_m * strength_m * max(fl(l_m), 0)   strength = max_muscle_stress * rest_muscle_area
     *   f     = sum F_m t_m ;  c = sum x_m x (F_m t_m)     internal force / couple of the actuation
     * and the equivalent external loads  F_ext += D^h(Q^T f),  tau_ext += D^h(c_v) + A^h(kappa x c_v D^) +
     * (e Q t) x f l^  with c_v the element couples averaged onto the Voronoi vertices.  The force-length law
     * fl, the layer geometry and strengths are DATA (softrod_config.muscle_fl_coef, softrod_set_muscle_layers);
     * every other recalled detail is a named switch of softrod_config (muscle_*).  Activations live in
     * softrod_state_view.muscle_activation (apply_activation, arm_push_env.py:257-271).                       */
    SOFTROD_FEAT_COOMM_MUSCLES = 1u << 13,
};
#define SOFTROD_MAX_MUSCLES 4
#define SOFTROD_MUSCLE_LONGITUDINAL 0 /* l_m = |nu_m|                                             */
#define SOFTROD_MUSCLE_TRANSVERSE 1   /* on the axis; l_m follows softrod_config.muscle_tm_length_law */
#define SOFTROD_MAX_FL_COEF 8
/* arm_push_env.py:160-196: the damped tapered arm with one sucker and the three muscle layers */
#define SOFTROD_FEATURES_ARM_PUSH                                                 \
    (SOFTROD_FEAT_ANALYTICAL_DAMPER | SOFTROD_FEAT_SUCKER_CONSTRAINT | SOFTROD_FEAT_COOMM_MUSCLES)
/* arm_push_env.py:520-618: the same arm + Cylinder + BodyBoundaryCondition + FixedJoint2Rigid (n_arm = 1) */
#define SOFTROD_FEATURES_ARM_PULL_WEIGHT (SOFTROD_FEATURES_ARM_PUSH | SOFTROD_FEAT_OCTO_HEAD)

#define SOFTROD_FEATURES_SOFTPENDULUM                                             \
    (SOFTROD_FEAT_GRAVITY | SOFTROD_FEAT_POINT_FORCE_NODE0_X |                   \
     SOFTROD_FEAT_PENDULUM_BC | SOFTROD_FEAT_ANALYTICAL_DAMPER)
#define SOFTROD_FEATURES_SOFTPENDULUM3D                                           \
    (SOFTROD_FEAT_GRAVITY | SOFTROD_FEAT_MOVING_BASE_BC |                        \
     SOFTROD_FEAT_ANALYTICAL_DAMPER | SOFTROD_FEAT_LAPLACE_FILTER)
#define SOFTROD_FEATURES_ARM_SINGLE                                               \
    (SOFTROD_FEAT_GRAVITY | SOFTROD_FEAT_PLANE_CONTACT_ANISO |                   \
     SOFTROD_FEAT_ANALYTICAL_DAMPER | SOFTROD_FEAT_REST_KAPPA_ACTION)
#define SOFTROD_FEATURES_OCTO_FLAT                                                \
    (SOFTROD_FEATURES_ARM_SINGLE | SOFTROD_FEAT_OCTO_HEAD)
#define SOFTROD_FEATURES_SOFT_ARM                                                 \
    (SOFTROD_FEAT_FIXED_BC | SOFTROD_FEAT_ANALYTICAL_DAMPER |                    \
     SOFTROD_FEAT_SPLINE_MUSCLE_TORQUES)

/* env_kind: which env's set_action / NaN check / reward / observation the step
 * kernel's prologue and epilogue implement.                                     */
#define SOFTROD_ENV_NONE 0           /* bare rod: softrod_substeps only            */
#define SOFTROD_ENV_SOFTPENDULUM 1   /* soft_pendulum/soft_pendulum.py:149-251     */
#define SOFTROD_ENV_SOFTPENDULUM3D 2 /* soft_pendulum_3d/soft_pendulum_3d.py:93-174 */
#define SOFTROD_ENV_ARM_SINGLE 3     /* octopus/arm_single_env.py:186-316           */
#define SOFTROD_ENV_OCTO_FLAT 4      /* octopus/flat_env.py:231-408                 */
#define SOFTROD_ENV_SOFT_ARM 5       /* soft_arm/soft_arm_tracking.py:160-259       */
#define SOFTROD_ENV_ARM_PUSH 6       /* octopus/arm_push_env.py:225-347 (OctoArmPush-v0 / -v1; parity unpinned: COOMM) */
#define SOFTROD_ENV_ARM_PULL_WEIGHT 7 /* octopus/arm_push_env.py:516-618 ArmPullWeightEnv (OctoArmPullWeight-v0): the same
                                        arm and step(), joined to a rigid Cylinder "weight" by FixedJoint2Rigid; with
                                        SOFTROD_FEATURES_ARM_PULL_WEIGHT (parity unpinned: COOMM)                     */
/* The muscle octopus (octopus/build_muscle_octopus.py): n_arm tapered 20-element arms with COOMM muscle layers joined to a
 * rigid head, SOFTROD_FEATURES_ARM_PULL_WEIGHT's feature set on several arms (parity unpinned: COOMM).  Their set_action /
 * get_state / step live in two small kernels either side of the step kernel (csrc/softrod_mocto.hpp). */
#define SOFTROD_ENV_CRAWL 8          /* octopus/crawl_env.py:175-318 CrawlEnv (OctoCrawl-v0): 8 arms, one sucker each      */
#define SOFTROD_ENV_ARM_TWO 9        /* octopus/arm_two_env.py:166-343 ArmTwoEnv (OctoArmTwo-v0): 2 arms, 3 suckers each   */
#define SOFTROD_ENV_REACH 10         /* octopus/reach_env.py:150-290 ReachEnv (OctoReach-v0): 8 arms, head held            */

/* math_mode (HIP library only; the oracle always uses libm). */
#define SOFTROD_MATH_LIBM 0 /* sqrt/sin/cos/acos/pow evaluated as written   */
#define SOFTROD_MATH_FAST 1 /* range-checked polynomial forms, no libm      */

/*
 * Everything that is identical for all rods of a batch.  Defaults are filled by
 * softrod_config_softpendulum() / softrod_config_softpendulum3d(); the
 * "PyElastica numerics" block holds the constants SURVEY.md App. A marks (?) so
 * that a session with pyelastica==1.0.0 importable can flip them without
 * touching a kernel.
 */
typedef struct softrod_config {
    uint32_t struct_size; /* = sizeof(softrod_config); ABI guard            */
    uint32_t features;    /* OR of softrod_feature                          */
    int32_t n_envs;       /* rods in this handle (batch shard)              */
    int32_t n_elem;       /* elements per rod (n_elems, soft_pendulum.py:64) */
    int32_t n_substeps;   /* step_skip = int(1/(fps*dt)), soft_pendulum.py:78 */
    int32_t math_mode;    /* SOFTROD_MATH_*                                 */
    int32_t env_kind;     /* SOFTROD_ENV_*                                  */
    int32_t filter_order; /* LaplaceDissipationFilter order                 */
    double dt;            /* time_step, soft_pendulum.py:62                 */
    double final_time;    /* soft_pendulum.py:61                            */
    /* straight_rod arguments, build.py:18-26,54-61 */
    double base_length;
    double base_radius;
    double density;
    double youngs_modulus;
    double shear_modulus; /* build.py passes none -> PyElastica default     */
    double gravity[3];    /* build.py:87-91                                 */
    double damping_constant; /* build.py:108                                */
    double tip_force[3];  /* SOFTROD_FEAT_TIP_FORCE only                    */
    /* SoftPendulum3D set_action, soft_pendulum_3d.py:57-58,99-113 */
    double base_step;     /* 1e-3 (applied in float32, as the reference)    */
    double base_limit;    /* 0.5                                            */
    /* PyElastica numerics (UNVERIFIED recollection) */
    double alpha_c;       /* 27/28 shear correction                         */
    double eps_length;    /* 1e-14 added to |dx|                            */
    double eps_rot_axis;  /* 1e-14 added to |w| before normalising the axis */
    double acos_shift;    /* 1e-10 subtracted inside arccos of _inv_rotate  */
    double eps_sin;       /* 1e-14 added inside sin of _inv_rotate          */
                          /* SOFTROD_MATH_FAST expands theta / sin(theta + eps_sin) to
                             first order in eps_sin: softrod_create refuses
                             acos_shift <= 0 or eps_sin > 1e-3 sqrt(2 acos_shift)
                             in that mode (SOFTROD_MATH_LIBM takes any values)  */
    int32_t time_two_half_adds; /* 1: t += dt/2 twice per substep; 0: += dt */
    int32_t damp_before_constrain; /* order inside PyElastica's constrain_rates
                                      operator group.  0 (default): registration
                                      order of the build function — constrain()
                                      before dampen() (build.py:81-113,
                                      soft_pendulum_3d/build.py:66-85), pyelastica
                                      1.0 OperatorGroupFIFO; 1: dampers first
                                      (mixin-__init__ order of pyelastica 0.3.x) */
    /* ---- OctoArmSingle-v0 (octopus/build.py:220-292, arm_single_env.py) ---- */
    int32_t contact_before_forcing; /* order inside synchronize(): 0 (default) =
                                       registration order, gravity then contact
                                       (octopus/build.py:236-283)              */
    int32_t damper_protocol;  /* AnalyticalLinearDamper: 0 (default) = the per-unit-mass protocol the bare
                                 `damping_constant=` keyword selects (build.py:108-113): v *= exp(-nu dt),
                                 omega *= exp(-nu dt m_elem J^-1)^dilatation; 1 = the uniform protocol
                                 (`uniform_damping_constant=`): exp(-nu dt) on every rate.  A switch for
                                 tools/sweep_switches.py like the ones above (was reserved1: same layout) */
    double plane_origin[3];   /* (0, 0, -r0)            octopus/build.py:244   */
    double plane_normal[3];   /* (0, 0, 1)              octopus/build.py:233   */
    double contact_k;         /* 1e2                    :241                    */
    double contact_nu;        /* 1e1                    :242                    */
    double slip_velocity_tol; /* 1e-8                   :245                    */
    double surface_tol;       /* PyElastica class default 1e-4                  */
    double kinetic_mu[3];     /* forward, backward, sideways   :248-256         */
    double static_mu[3];      /* 2 x kinetic                   :257             */
    double control_penalty_coeff; /* 0.001              arm_single_env.py:62    */
    double target[2];         /* (1, 0)                 arm_single_env.py:165   */
    double kappa_range[2];    /* arm_single_env.py:111                          */
    double kappa_rate_range[2]; /* :113                                         */
    /* ---- OctoFlat-v0 (octopus/build.py:30-132, flat_env.py:55-110) ---- */
    int32_t n_arm;            /* 8 rods per env (n_elem is per arm: 10)         */
    int32_t n_knots;          /* n_action per arm: 3 (flat_env.py:62)           */
    double head_radius;       /* 0.04                   octopus/build.py:38     */
    double head_density;      /* 700                    :39                     */
    double joint_k;           /* body_arm_k  1e6        :36                     */
    double joint_nu;          /* 1e-3                   :128                    */
    double joint_kt;          /* body_arm_kt 1e0        :37                     */
    /* ---- SoftArmTracking-v0 (soft_arm/soft_arm_tracking.py:107-158,261-383) ---- */
    int32_t n_ctrl;           /* number_of_control_points per direction: 4  :132 */
    int32_t n_spline_pieces;  /* polynomial pieces of the interpolant (3 for 4 + 2
                                 points, not-a-knot); see softrod_set_spline_table */
    double muscle_torque_scale; /* alpha = torque_scale * radius * E   :350        */
    double max_activation_rate; /* max_rate_of_change_of_activation: inf  :143     */
    double arm_target[3];     /* target_location (game_mode 1)          :147      */
    /* ---- ControllableFixConstraint (octopus/controllable_constraint.py:24-69) ---- */
    int32_t n_suckers;        /* constraints of this kind registered on the rod (0..4)      */
    int32_t sucker_index[4];  /* SuckerController.index of each (node AND element index)    */
    int32_t reserved2;
    double sucker_reduction_ratio; /* initial reduction_ratio of every sucker (1.0, :11); the
                                 controllers are on after finalize (arm_push_env.py:222)  */
    /* ---- SOFTROD_FEAT_COOMM_MUSCLES (octopus/build.py:295-338; COOMM itself NOT on disk: every field below
     *      restates a recalled detail and is a switch for the day the muscle-env fixtures exist) ---- */
    int32_t n_muscles;        /* layers of create_es_muscle_layers: 3 (two longitudinal, one transverse)   */
    int32_t muscle_kind[4];   /* SOFTROD_MUSCLE_* of each layer                                            */
    int32_t muscle_fl_degree; /* degree of the force-length polynomial (3)                                 */
    int32_t muscle_equiv_load_form; /* internal load -> equivalent external load: 0 (default) = the physical form,
                                 F = D^h(Q^T f), tau = D^h(c) + A^h(kappa x c D^) + (e Q t) x f l^;
                                 1 = PyElastica's own internal-load form applied to (f, c): Q^T f / e, c / eps^3,
                                 (Q t) x f l^                                                               */
    int32_t muscle_position_current_radius; /* 1 (default): x_m = rod.radius * ratio, the CURRENT (volume-preserving)
                                 radius; 0: the rest radius                                               */
    int32_t muscle_tm_length_law; /* TransverseMuscle's normalised length: 0 (default) = 1 / sqrt(|nu_m|)
                                 (incompressible cross-section: the radial fibre shortens as the arm extends);
                                 1 = |nu_m| like a longitudinal muscle                                      */
    int32_t arm_push_mode;    /* ArmPushEnv(mode=...): 0 "discrete" (OctoArmPush-v0), 1 "continuous" (-v1)
                                 arm_push_env.py:90-130                                                    */
    int32_t head_fixed;       /* 1: the rigid body ALSO carries OneEndFixedBC (reach_env.py:126-130): position, directors
                                 held at their initial values, velocities at zero — the head does not move          */
    double muscle_fl_coef[8]; /* fl(l) = sum_k coef[k] l^k, clipped at 0 from below; the cubic of the paper,
                                 max{3.06 l^3 - 13.64 l^2 + 18.01 l - 6.44, 0}: (-6.44, 18.01, -13.64, 3.06)      */
    /* ---- the rigid body of the SOFTROD_FEAT_OCTO_HEAD sets, as Cylinder(start, direction = e_z, normal = e_y,
     *      base_length, base_radius = head_radius, density = head_density) allocates it, and its joints ---- */
    double head_center[3];    /* start + direction * base_length / 2.  FlatEnv: (0, 0, 0) (octopus/build.py:95-105);
                                 ArmPullWeightEnv: (-0.9 * 0.015, 0, -0.012) (arm_push_env.py:553-566)            */
    double head_length;       /* Cylinder base_length.  FlatEnv: 2 r0; ArmPullWeightEnv: 2 * radius_base = 0.024  */
    double joint_angle0;      /* FixedJoint2Rigid(angle=...) of arm a, degrees: joint_angle0 + a * joint_angle_step. */
    double joint_angle_step;  /* FlatEnv: 0, 360 / n_arm (octopus/build.py:73-74,117-132); ArmPullWeightEnv: 0, 0
                                 (arm_push_env.py:585-587); build_octopus_muscles: 22.5, 45; build_two_arms: 90, 180
                                 (build_muscle_octopus.py:85-86,202-203)                                           */
    double damper_time_step;  /* AnalyticalLinearDamper(time_step=...) when it is NOT the stepper's dt: the muscle octopus
                                 registers its dampers with 7e-5 (build_muscle_octopus.py:101-106,217-222) and is stepped
                                 with 5e-5 (crawl_env.py:63).  0 = dt.                                              */
} softrod_config;

#define SOFTROD_MAX_SUCKERS 4
#define SOFTROD_MAX_CTRL 8           /* control points per direction             */
#define SOFTROD_MAX_SPLINE_PIECES 8

/* Per-env I/O widths implied by env_kind (OctoFlat: at the reference's n_arm = 8,
 * n_elem = 10, n_knots = 3; ArmPush: continuous mode at n_elem = 40). */
int softrod_action_dim(int env_kind); /* 1, 2, 7, 24, 8, 2                  */
int softrod_obs_dim(int env_kind);    /* 4, 9, 25, 8*56 + 13 = 461, 14, 84  */
int softrod_aux_dim(int env_kind);    /* 0, 1 (info["tilt"], float64), 0    */
/* The same for a given configuration: OctoFlat widths follow n_arm, n_elem and n_knots
 * (flat_env.py:112-141): action n_arm*n_knots; obs n_arm*((n-1) + 4(n+1) + n_knots) + 13,
 * the "individual" rows followed by "shared" (flat_env.py:231-286).  ArmPush (arm_push_env.py:100-130): action 1
 * (discrete: the index 0 / 1 as a float32) or 2 (continuous); obs 2 (n_elem + 1) + 2. */
int softrod_config_action_dim(const softrod_config* cfg);
int softrod_config_obs_dim(const softrod_config* cfg);

typedef struct softrod_handle softrod_handle;

/*
 * Borrowed device pointers to the resident state (valid until destroy).
 * Layout (DESIGN.md "HBM layout"): structure-of-arrays, component-major, one
 * row of `lane_stride` entries per rod (64 for n_elem <= 63: lane k of the rod's
 * wavefront owns node k, element k and Voronoi vertex k; 128 for n_elem <= 126:
 * lane k owns indices 2k and 2k+1; OctoFlat: 64 per wavefront of the env's
 * workgroup, arms `arm_stride` slots apart):
 *     position[(c * n_envs + env) * lane_stride + node]        c = 0..2
 *     director[((r*3 + c) * n_envs + env) * lane_stride + elem]  row r of Q, lab comp. c
 * Mirrors rod.position_collection / velocity_collection / director_collection /
 * omega_collection / tangents of the reference (soft_pendulum.py:152-154).
 */
typedef struct softrod_state_view {
    int32_t n_envs, n_elem, lane_stride;
    int32_t arm_stride; /* OctoFlat: arm a of an env is slots a*arm_stride ..
                           a*arm_stride + n_elem of the env's row; else 0        */
    double* position; /* [3][n_envs][lane_stride] */
    double* velocity; /* [3][n_envs][lane_stride] */
    double* director; /* [9][n_envs][lane_stride] */
    double* omega;    /* [3][n_envs][lane_stride] */
    double* tangents; /* [3][n_envs][lane_stride]  as of the last force evaluation */
    double* time;     /* [n_envs]  simulated time (soft_pendulum.py:141,184) */
    double* control;  /* [4][n_envs]  MovingBaseController position x,y and
                         velocity x,y (soft_pendulum_3d/build.py:15-20);
                         SoftArmTracking: tick (substeps since reset, :222) and
                         the target position x,y,z (wsol[tick], :218-219)     */
    double* kappa;    /* [3][n_envs][lane_stride]  rod.kappa as of the last force
                         evaluation (arm_single_env.py:189); maintained by the
                         feature sets with SOFTROD_FEAT_REST_KAPPA_ACTION     */
    double* rest_kappa; /* [3][n_envs][lane_stride]  rod.rest_kappa (:235);
                         SoftArmTracking: rows 0, 1 = torque_magnitude_cache of the
                         normal / binormal muscle (muscle_torques_with_bspline.py:156) */
    double* env_memory; /* [n_envs][lane_stride]  ArmSingle: prev_kappa_state
                         [0..n-2] (arm_single_env.py:172,190-198); its
                         prev_com_state lives in control[0..1].  SoftArmTracking:
                         [0 .. 2 n_ctrl) points_cached[1, 1:-1] of the two muscles,
                         [16], [17] their initial_call_flag                    */
    float* prev_action; /* [n_envs][7]  the env's _prev_action, written by
                         softrod_step (soft_pendulum.py:165), cleared by reset
                         only where the reference does (soft_pendulum_3d.py:68);
                         OctoFlat: [n_envs][n_arm*n_knots]                     */
    double* head;     /* [20][n_envs]  OctoFlat rigid head (octopus/build.py:103-105):
                         position[3], velocity[3], directors[9] (row-major),
                         omega[3], target[2] (flat_env.py:221); else unused    */
    double* bc_targets; /* [12][n_envs]  what the boundary condition holds node 0 / element 0
                         to: fixed_position[3], fixed_directors[9] (row-major), captured at
                         reset (build.py:81-85: constrained_position_idx=(0,),
                         constrained_director_idx=(0,)).  With the rows above it makes the
                         view a complete snapshot: copying every array out and back in
                         restores a batch exactly (checkpoint / resume).            */
    double* sucker_ratio; /* [SOFTROD_MAX_SUCKERS][n_envs]  effective reduction ratio of each
                         ControllableFixConstraint: SuckerController.reduction_ratio while its
                         flag is on, 0 while it is off (x * (1 - 0) is x).  Written by the
                         caller between steps (arm_two_env.py:228 sets it from the action) */
    double* muscle_activation; /* [SOFTROD_MAX_MUSCLES][n_envs][lane_stride]  MuscleForce.activation of each layer, per
                         element (apply_activation broadcasts a scalar, arm_push_env.py:257-271, or takes an
                         array, arm_two_env.py:246-248, reach_env.py:176-179).  Written by the ArmPush prologue
                         from the action; with env_kind NONE by the caller between steps.  NULL without
                         SOFTROD_FEAT_COOMM_MUSCLES                                                          */
    int32_t* sucker_index; /* [SOFTROD_MAX_SUCKERS][n_envs]  SuckerController.index of each sucker of each env,
                         Python indexing: i >= 0 holds node i and element i; i < 0 holds node n_elem + 1 + i and
                         element n_elem + i (`velocity_collection[..., -1]` is the LAST NODE, `omega_collection
                         [..., -1]` the LAST ELEMENT: arm_push_env.py:262, controllable_constraint.py:66-69).
                         Starts as softrod_config.sucker_index; the ArmPush prologue rewrites row 0 from the
                         action (arm_push_env.py:257,262,270; crawl_env.py:236-238 does the same)              */
    double* material; /* [SOFTROD_MATERIAL_ROWS][lane_stride]  per-node / per-element constants of
                         a TAPERED rod (softrod_set_radius_profile), shared by all envs; NULL
                         for a uniform rod.  Read-only for the caller.                     */
    /* ---- the muscle octopus envs (SOFTROD_ENV_CRAWL / _ARM_TWO / _REACH); NULL otherwise.  Their sucker_ratio /
     *      sucker_index rows are [SOFTROD_MAX_SUCKERS][n_envs * n_arm] (arm a of env e at e * n_arm + a) ---- */
    double* env_aux;  /* [8][n_envs]  rows 0-2 the env's target (crawl_env.py:172, arm_two_env.py:160: (5, 0);
                         reach_env.py:141-143: np_random.random(3) * sum(rest_lengths)), rows 3-4 the head's x, y before
                         the step (`xposbefore`, crawl_env.py:248), row 5 the episode's own final_time (0: the config's).  (rod.kappa[0] of get_state, crawl_env.py:178,
                         is row 0 of `kappa`: what the last substep's force evaluation cached; zeros after a reset)  */
    float* prev_kappa; /* [n_envs][n_arm * (n_elem - 1)]  ArmTwoEnv._prev_kappa (arm_two_env.py:103,196-203): float32,
                         NOT cleared by reset() — it belongs to the env object, as in the reference            */
} softrod_state_view;

/* Fill `cfg` with SoftPendulumEnv.__init__ defaults (soft_pendulum.py:59-78)
 * and build_soft_pendulum's constants (build.py:18-26,87-113).             */
int softrod_config_softpendulum(softrod_config* cfg, int n_envs);
/* Same for SoftPendulum3DEnv (soft_pendulum_3d.py:28-58) and
 * build_soft_pendulum_3d (soft_pendulum_3d/build.py:43-86).                 */
int softrod_config_softpendulum3d(softrod_config* cfg, int n_envs);
/* Same for ArmSingleEnv (octopus/arm_single_env.py:55-113) and build_arm
 * (octopus/build.py:220-292).                                               */
int softrod_config_arm_single(softrod_config* cfg, int n_envs);
/* Same for FlatEnv (octopus/flat_env.py:55-110) and build_octopus
 * (octopus/build.py:30-217): 8 arms of 10 elements, rigid head, joints.     */
int softrod_config_octo_flat(softrod_config* cfg, int n_envs);
/* Same for SoftArmTrackingEnv (soft_arm/soft_arm_tracking.py:107-158) and the
 * simulator its reset builds (:261-383), game_mode 1.                       */
int softrod_config_soft_arm(softrod_config* cfg, int n_envs);

/* Replaces the constant part of
 *   make_interp_spline(points_cached[0], points_cached[1])(cumsum(system.lengths))
 * (muscle_torques_with_bspline.py:150-158): the interpolant through the n_ctrl + 2 control
 * points (the two end values are zero) is linear in the control values,
 *   S(s) = sum_j y_j phi_j(s),
 * and every phi_j is a piecewise cubic on fixed breakpoints.  The caller passes that form:
 *   breaks: host [n_spline_pieces + 1] float64, ascending
 *   coef:   host [n_spline_pieces][n_ctrl][4] float64, ascending powers of (s - breaks[p])
 * (scipy: PPoly.from_spline(make_interp_spline(x, e_j)); s outside the breaks uses the end
 * pieces, as BSpline's default extrapolation does).  Needed before the first softrod_step
 * of a handle with SOFTROD_FEAT_SPLINE_MUSCLE_TORQUES.                          */
int softrod_set_spline_table(softrod_handle* h, const double* breaks, const double* coef);

/* Replaces: CosseratRod.straight_rod(..., base_radius=<array of n_elements radii>, ...) — a
 * TAPERED rod, as the muscle-arm envs allocate it
 *   radius = np.linspace(radius_base, radius_tip, n_elem + 1); radius_mean = (radius[:-1] + radius[1:]) / 2
 *                                   octopus/arm_push_env.py:160-179, build_muscle_octopus.py
 * Every per-element constant of the allocation (area, volume, nodal masses, second moments,
 * shear / bend matrices and their Voronoi average, the damper's per-element coefficients,
 * the contact radius) then varies along the rod: the library keeps them as per-lane rows
 * (softrod_state_view.material) instead of kernel-argument scalars and runs its general
 * instantiation.  radius: host [n_elem] float64.  Call before the first reset; rods of up to
 * 63 elements; not with SOFTROD_FEAT_OCTO_HEAD.                                            */
#define SOFTROD_MATERIAL_ROWS 16
int softrod_set_radius_profile(softrod_handle* h, const double* radius);

/* Replaces: the geometry and strength of the muscle layers handed to ApplyMuscles — for the reference's
 * create_es_muscle_layers(radius_mean, radius_base) (octopus/build.py:295-338):
 *   LongitudinalMuscle(muscle_init_angle = +-pi/2, ratio_muscle_position = (0, -6/9, 0) per element,
 *                      rest_muscle_area = (radius_mean / radius_base)^2, max_muscle_stress = 0.5)  x 2
 *   TransverseMuscle(rest_muscle_area = (radius_mean / radius_base)^2, max_muscle_stress = 1.0)
 * ratio_position: host [n_muscles][3][n_elem] float64, the muscle's position in the material frame in units of
 *                 the element radius (how muscle_init_angle enters it is the caller's: _capi.es_muscle_layers);
 * strength:       host [n_muscles][n_elem] float64 = max_muscle_stress * rest_muscle_area, SIGNED (a transverse
 *                 muscle extends the arm: negative).
 * Needed before the first softrod_step / softrod_substeps of a handle with SOFTROD_FEAT_COOMM_MUSCLES; rods of
 * up to 63 elements, one rod per env.                                                                     */
int softrod_set_muscle_layers(softrod_handle* h, const double* ratio_position, const double* strength);

/* Same for ArmPullWeightEnv (octopus/arm_push_env.py:516-618; registered as OctoArmPullWeight-v0 with mode
 * "continuous", gym_softrobot/__init__.py:48-52): time_step 2.5e-5 (1000 substeps per env.step), damper 0.05 * 2 * 5e2,
 * the sucker at reduction_ratio 0.9, a Cylinder of radius 0.015 / length 0.024 / density 700 joined to node 0 by
 * FixedJoint2Rigid(k = 1e6, nu = 1e-2, kt = 1, angle = 0, radius = 0.015) and held by BodyBoundaryCondition.      */
int softrod_config_arm_pull_weight(softrod_config* cfg, int n_envs);

/* The muscle octopus envs: env_kind = SOFTROD_ENV_CRAWL (CrawlEnv.__init__ / reset, crawl_env.py:61-174 over
 * build_octopus_muscles, build_muscle_octopus.py:66-179), SOFTROD_ENV_ARM_TWO (arm_two_env.py:55-164 over build_two_arms,
 * :182-291) or SOFTROD_ENV_REACH (reach_env.py:53-148).  Arms: 20 elements, 0.25 long, density 1000, E 1.5e4, G 1e4,
 * radii linspace(0.013, 0.0042, n) through softrod_set_radius_profile, layers through softrod_set_muscle_layers (both
 * once: every arm is the same rod); head Cylinder(start (0, 0, -0.026), e_z, e_y, 0.026, 0.04, 50); joints k 1e6, kt 1e2,
 * nu 1e-3; dampers 0.2 * 1e-2 at time_step 7e-5; dt 5e-5, 800 substeps per env.step.  Resets go through softrod_reset_octo
 * / softrod_queue_push_octo (arm frames from the caller, build_muscle_octopus.py:87-93; `target`: FOUR numbers per env
 * here — the target's x, y, z (CrawlEnv / ArmTwoEnv: 5, 0, 0; ReachEnv: its random point) and the episode's own
 * final_time, 0 = softrod_config.final_time (CrawlEnv(config_random_final_time=True) draws one per reset,
 * crawl_env.py:135-136)).                                                                                         */
int softrod_config_muscle_octopus(softrod_config* cfg, int n_envs, int env_kind);

/* Same as softrod_config_arm_single for ArmPushEnv (octopus/arm_push_env.py:65-224): the 40-element arm tapered
 * 12:1 (the radii themselves go through softrod_set_radius_profile), damper, one sucker, three muscle layers
 * (softrod_set_muscle_layers); mode 0 = "discrete" (OctoArmPush-v0), 1 = "continuous" (OctoArmPush-v1).      */
int softrod_config_arm_push(softrod_config* cfg, int n_envs, int mode);

/* Replaces the constant part of set_action's
 *   interp1d(linspace(0,1,n_action), action, kind="cubic")(linspace(0,1,n_seg))
 * (octopus/arm_single_env.py:226-235): cubic not-a-knot interpolation through fixed
 * knots evaluated at fixed abscissae is linear in the action, rest_kappa[0,:] = W a.
 * basis: host [n_elem-1][action_dim] float64 (row-major), computed by the caller
 * with the same scipy call on unit vectors.                                  */
int softrod_set_action_basis(softrod_handle* h, const double* basis);

/* Replaces: BaseSimulator() + build_soft_pendulum(...) + simulator.finalize()
 * (soft_pendulum.py:115-138) for a whole batch: allocates resident device
 * state for cfg->n_envs rods on HIP device `device`.                        */
int softrod_create(const softrod_config* cfg, int device, softrod_handle** out);

/* Replaces: the state part of SoftPendulumEnv.reset (soft_pendulum.py:108-147)
 * i.e. CosseratRod.straight_rod with direction=(cos t, sin t, 0),
 * normal=(sin t, -cos t, 0) (build.py:47-61), time = 0.
 * theta0: host [n_envs] radians (drawn by the caller exactly as build.py:47-49).
 * mask:   host [n_envs] (non-zero = reset this rod) or NULL = all.          */
int softrod_reset(softrod_handle* h, const double* theta0, const uint8_t* mask,
                  void* stream);

/* General straight rod (CosseratRod.straight_rod arguments, build.py:54-61):
 * start/direction/normal are host [n_envs][3]; mask as above.  Also clears the
 * moving-base controller of the reset rods (soft_pendulum_3d.py:67).  Used by
 * SoftPendulum3D (direction = (sin tilt, 0, cos tilt), normal = +y,
 * soft_pendulum_3d/build.py:51-53) and by the known-answer tests.           */
int softrod_reset_straight(softrod_handle* h, const double* start,
                           const double* direction, const double* normal,
                           const uint8_t* mask, void* stream);

/* Replaces: the state part of FlatEnv.reset (octopus/flat_env.py:171-229) ->
 * build_octopus (octopus/build.py:52-217): n_arm straight rods (normal e_z), the
 * Cylinder head at the origin, finalize()'s first constraint pass, time = 0.
 *   arm_start, arm_direction  host [n_envs][n_arm][3]: the
 *       Rotation.from_euler("z", 360/n_arm * i, degrees=True).apply(...) results of
 *       build.py:76-80, computed by the caller as the reference does
 *   target                    host [n_envs][2]: (2 - 0.5) * np_random.random(2) + 0.5
 *   mask                      as softrod_reset                                 */
int softrod_reset_octo(softrod_handle* h, const double* arm_start,
                       const double* arm_direction, const double* target,
                       const uint8_t* mask, void* stream);

/* Replaces: Env.step for every rod (soft_pendulum.py:176-251,
 * soft_pendulum_3d.py:115-174): set_action -> n_substeps x PositionVerlet.step
 * -> NaN check -> reward -> truncation -> get_state.  Asynchronous on `stream`.
 *   actions     device [n_envs][action_dim] float32
 *   obs         device [n_envs][obs_dim]    float32
 *   reward      device [n_envs]             float64
 *   terminated  device [n_envs]             uint8
 *   truncated   device [n_envs]             uint8
 *   aux         device [n_envs][aux_dim]    float64 or NULL (info extras)    */
int softrod_step(softrod_handle* h, const float* actions, float* obs,
                 double* reward, uint8_t* terminated, uint8_t* truncated,
                 double* aux, void* stream);

/* softrod_step with all per-env outputs in ONE buffer, for the multi-GPU path (one
 * all-gather per env.step, unpacked with views only): packed is device
 * [n_envs][ro + 4] 32-bit words per env, ro = obs_dim rounded up to even:
 *   [ obs (obs_dim x float32) | pad to ro | reward (float64, 2 words, 8-byte
 *     aligned) | terminated, truncated (bytes 0 and 1 of one word) | 0 ]       */
int softrod_step_packed(softrod_handle* h, const float* actions, float* packed,
                        double* aux, void* stream);

/* ---- Device-side auto-reset ---------------------------------
 * The reference has no vector env and no auto-reset: a finished env is reset by its
 * caller (gym_softrobot/debug/make.py:16-23).  For a resident batch that would cost a
 * device->host read of the flags on every step.  Instead the caller stages, ahead of
 * time, the next `depth` resets of every env — drawn on the host from the env's own
 * NumPy stream exactly as the reference's build function draws them
 * (build.py:47-49, soft_pendulum_3d/build.py:51, flat_env.py:221), so the streams stay
 * bit-identical — and softrod_step / softrod_step_packed then apply Gymnasium-1.0
 * VectorEnv NEXT_STEP semantics on the device: an env whose previous step returned
 * terminated or truncated consumes its next staged reset INSTEAD of stepping; that call
 * returns its reset observation, reward 0 and both flags clear.  An env that needs a
 * record when none is staged stays finished and is counted in `underflow`.
 *
 *   softrod_autoreset_enable   once per handle; depth = records kept per env
 *   softrod_queue_push*        stage counts[e] (<= max_count) more resets of env e; the
 *                              arrays are host [n_envs][max_count]... with the per-reset
 *                              layout of softrod_reset / _reset_straight / _reset_octo.
 *                              Fails if staged-but-unconsumed + new records would exceed
 *                              depth (as of the last softrod_queue_status).
 *   softrod_queue_status       synchronises; consumed: host [n_envs] records used so far
 *   softrod_queue_status_begin / _poll   the same read without stalling the stream: _begin
 *                              enqueues the copy of the counters (pinned host memory) and an
 *                              event; _poll returns 1 and fills consumed / underflow once the
 *                              event has passed, 0 before; with wait != 0 it waits for that
 *                              event only (work enqueued after _begin keeps the GPU busy
 *                              meanwhile).  The values are as of _begin, which is safe:
 *                              consumed only grows, so a top-up computed from them stages
 *                              at most what fits.  A second _begin SUPERSEDES a read still in
 *                              flight (its result is dropped; _poll then reports the newer
 *                              one); _poll without a _begin is an error.  An env that found no
 *                              record goes on stepping its finished episode and goes on
 *                              reporting truncated (its clock only grows) or terminated (a NaN
 *                              state stays NaN), so a shortage shows in the step outputs at once
 *                              and in `underflow` at the next status read
 *   softrod_queue_advance      mark by[e] staged records of env e as used (a manual reset
 *                              took the env's next draw), or all of them if by[e] < 0
 *                              (the env's stream was re-seeded); synchronises
 * softrod_reset* of an env clears its pending auto-reset.                            */
int softrod_autoreset_enable(softrod_handle* h, int depth);
int softrod_queue_push(softrod_handle* h, const double* theta0, const int32_t* counts,
                       int max_count, void* stream);
int softrod_queue_push_straight(softrod_handle* h, const double* start,
                                const double* direction, const double* normal,
                                const int32_t* counts, int max_count, void* stream);
int softrod_queue_push_octo(softrod_handle* h, const double* arm_start,
                            const double* arm_direction, const double* target,
                            const int32_t* counts, int max_count, void* stream);
int softrod_queue_status(softrod_handle* h, int32_t* consumed, int32_t* underflow,
                         void* stream);
int softrod_queue_status_begin(softrod_handle* h, void* stream);
int softrod_queue_status_poll(softrod_handle* h, int wait, int32_t* consumed, int32_t* underflow);
int softrod_queue_advance(softrod_handle* h, const int32_t* by, void* stream);

/* Multi-GPU without a collective call per step (no reference counterpart; gym_softrobot_amd/
 * distributed.py, transport "p2p"): copy this handle's `n_envs` packed rows (`row_words` 32-bit
 * words each, as softrod_step_packed wrote them) into rows first_row .. first_row + n_envs - 1 of
 * EVERY buffer in `peer_buffers` (host array of n_peers <= SOFTROD_MAX_PEERS device pointers: the
 * other ranks' exchange buffers, IPC-mapped, and this rank's own), with ONE small kernel on `stream`
 * — enqueued right behind the step kernel it is a few microseconds in order, where a collective on
 * a second stream costs this workload ~37 us of cross-queue dependency latency per step (DESIGN.md
 * §4).  The stores are system-scope write-through; between GPUs they travel over xGMI.
 * tag_word >= 0: once ALL rows of this call have been acknowledged, 32-bit word `tag_word` of every
 * peer buffer receives `tag` (system-scope release): the generation word of this rank in the
 * receivers' buffers (distributed.py keeps `world` of them behind the rows and checks them in
 * sync()).  tag_word < 0: rows only.  The tagged form counts arrivals in ONE per-handle word
 * (allocated and zeroed in softrod_create): tagged scatters of one handle must be stream-ordered
 * with each other — two in flight on different streams would corrupt each other's tag.         */
#define SOFTROD_MAX_PEERS 16
int softrod_scatter_rows(softrod_handle* h, const float* packed, const uint64_t* peer_buffers,
                         int n_peers, int row_words, int64_t first_row, int64_t tag_word,
                         uint32_t tag, void* stream);

/* Exchange buffers of transport "p2p": device memory that OTHER GPUs store into.  A GPU's L2 does
 * not snoop a peer's stores into its HBM, so such a buffer must not be an ordinary (coarse-grained,
 * L2-cached) allocation: softrod_exchange_alloc returns UNCACHED device memory
 * (hipDeviceMallocUncached; *memory_kind = SOFTROD_EXCHANGE_FINEGRAINED where only
 * hipDeviceMallocFinegrained is granted), zeroed, together with its 64-byte IPC handle (NULL: not
 * wanted).  softrod_exchange_open maps a handle received from the process that owns `owner_device`
 * (peer access device -> owner_device is enabled first when they differ; pass -1 to skip) and
 * returns the pointer valid in THIS process; _close unmaps it, _free releases an allocation of
 * _alloc.  No handle argument: these belong to the process, not to a batch.                     */
#define SOFTROD_IPC_HANDLE_BYTES 64
#define SOFTROD_EXCHANGE_UNCACHED 1
#define SOFTROD_EXCHANGE_FINEGRAINED 2
int softrod_exchange_alloc(int device, uint64_t bytes, void** dev_ptr, uint8_t* ipc_handle,
                           int* memory_kind);
int softrod_exchange_open(int device, const uint8_t* ipc_handle, int owner_device, void** dev_ptr);
int softrod_exchange_close(int device, void* dev_ptr);
int softrod_exchange_free(int device, void* dev_ptr);

/* Replaces: get_state() at reset (soft_pendulum.py:145-161,
 * soft_pendulum_3d.py:93-98).  prev_action is device [n_envs][action_dim]
 * float32, or NULL = the resident copy of the last stepped action.          */
int softrod_observe(softrod_handle* h, const float* prev_action, float* obs,
                    void* stream);

/* Run `n` bare PositionVerlet substeps with fixed per-env forcing inputs and no
 * env epilogue (the inner loop of soft_pendulum.py:183-184 alone); `actions`
 * (device [n_envs] float32 or NULL) feeds SOFTROD_FEAT_POINT_FORCE_NODE0_X.
 * Used by the known-answer tests and by the kernel micro-benchmarks.        */
int softrod_substeps(softrod_handle* h, const float* actions, int n, void* stream);

int softrod_state_view_get(softrod_handle* h, softrod_state_view* out);

/* Kernel timing with HIP events recorded on the launch stream around each
 * softrod_step / softrod_substeps kernel (no host synchronisation at record
 * time, so it can stay on inside a timed region).
 *   softrod_set_timing(h, n)      n > 0: keep event pairs for the next n launches
 *                                 (ring restarts at 0); n = 0: off (default).
 *   softrod_kernel_times_ms(...)  synchronises on the recorded events and writes
 *                                 the per-launch durations, oldest first; *count
 *                                 receives how many were written (<= cap).
 *   softrod_last_kernel_ms(...)   duration of the most recent timed launch.    */
int softrod_set_timing(softrod_handle* h, int n_launches);
int softrod_kernel_times_ms(softrod_handle* h, float* out_ms, int cap, int* count);
int softrod_last_kernel_ms(softrod_handle* h, float* ms);

/* Which step-kernel tier softrod_step launches for this handle, as a short stable string: the
 * kernel template's name, its specialisation and the workgroup shape, e.g.
 * "softrod_step_fast_kernel<SoftPendulum,epl=1>" or "softrod_octo_step_kernel<zup,2 waves,4 envs/wg>".
 * The tier follows from softrod_config alone.  The A/B switches of the measurement builds
 * (SOFTROD_OCTO_ONE_WAVE, SOFTROD_OCTO_ONE_ENV_PER_BLOCK, SOFTROD_NO_WINDOW, SOFTROD_WINDOW_PAIRED,
 * SOFTROD_WINDOW_REFRESH) are honoured by softrod_create ONLY when SOFTROD_DEBUG_SWITCHES=1 is set as
 * well, so a stray variable cannot change what a product process runs; this call is how a host (and
 * tests/test_gpu_debug_switches.py) sees the outcome.  No reference counterpart (ABI v15).        */
const char* softrod_kernel_tier(softrod_handle* h);

const char* softrod_last_error(softrod_handle* h);
int softrod_destroy(softrod_handle* h);
int softrod_abi_version(void);
/* First 16 hex digits of the SHA-256 over the sources this library was built from (the .hpp files of csrc/ in
 * name order, softrod_capi.hip, include/softrod.h; the Makefile passes it as SOFTROD_SOURCE_HASH).
 * Profiles record it, and bench.py refuses to price a kernel against instruction counts / traffic
 * measured on a different build (profiles/valu_counts.json, hbm_traffic.json).  No reference
 * counterpart: measurement plumbing.                                                            */
const char* softrod_source_hash(void);

#ifdef __cplusplus
}
#endif
#endif /* SOFTROD_H */
