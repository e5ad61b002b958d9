// softrod_kernels.hpp — CDNA4 (gfx950) device code of the batched Cosserat-rod stepper.
//
// Mapping (DESIGN.md "Execution mapping"): ONE ROD PER 64-LANE WAVEFRONT.  Lane k owns
// node k (x, v), element k (Q, omega) and Voronoi vertex k (kappa, couples); a rod of
// n_elem <= 63 elements therefore lives in one wave's registers.  The whole inner
// loop of `env.step` (soft_pendulum.py:183-184: step_skip x PositionVerlet.step) runs
// register-resident: HBM is read once and written once per env.step.  Every
// nearest-neighbour stencil of the discretisation (x[k+1]-x[k], n[k]-n[k-1],
// Q[k+1]Q[k]^T, 1/2(a[k]+a[k-1]), the first pass of the Laplace filter) is a DPP wave shift
// (v_mov_b32_dpp wave_shl:1 / wave_shr:1, 2 per fp64 value) — no LDS round trip, no
// barrier.  LDS appears in two places only: the 13-tap stencil that replaces six passes of
// the order-7 Laplace filter (laplace_filter_rates_lds7) and OctoFlat's head (softrod_octo.hpp).
//
// Lane k owns nodes EPL*k .. EPL*k+EPL-1 (and the elements / Voronoi vertices of the same
// indices): EPL = 1 for rods of up to 63 elements, EPL = 2 up to 126 (BASELINE config 3,
// "100 elements").  Neighbours inside a lane are plain registers; only the last slot's
// "next" and the first slot's "previous" cross lanes, so the DPP count per substep does not
// grow with EPL.  Global rows are 64*EPL wide, index = node index.
//
// File map: softrod_contact.hpp (rod-plane contact), softrod_fast.hpp (the default step
// kernel), softrod_planar.hpp (SoftPendulum's planar substep), softrod_octo.hpp (OctoFlat: one
// env per workgroup), softrod_window.hpp (64..102-element arms on two overlapping windows);
// this file holds the state layout, the env prologues/epilogues, the reset /
// observe / auto-reset kernels and the LIBM kernel.
//
// Two step kernels share this file's state layout and env prologue/epilogue:
//   softrod_step_libm_kernel  SOFTROD_MATH_LIBM (EPL = 1): the substep exactly as PyElastica
//                             writes it (sqrt / sincos / acos / pow / divisions, two
//                             half kinematic steps, constrain_values after each) — the
//                             on-device reference the fast kernel is tested against.
//   softrod_step_fast_kernel  SOFTROD_MATH_FAST (default): same mathematics,
//                             reorganised for the fp64 VALU (softrod_fast.hpp).
//
// The arithmetic restates PyElastica's PositionVerlet substep for the simulators that
// build_soft_pendulum (gym_softrobot/envs/soft_pendulum/build.py:29-115),
// build_soft_pendulum_3d (gym_softrobot/envs/soft_pendulum_3d/build.py:43-86), build_arm and
// build_octopus (gym_softrobot/envs/octopus/build.py:220-292, 52-217) assemble;
// the order of operations is documented in DESIGN.md "substep order" and mirrored by
// the CPU oracle (oracle/softrod_oracle.c), against which tests/ check this file.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/softrod.h"

namespace softrod {

constexpr int kLanes = 64;

// Compile-time feature sets: the step kernels are instantiated for the feature masks of
// the registered envs (so that unused features cost neither instructions nor registers)
// plus one instantiation that reads the mask at run time (known-answer tests).
constexpr unsigned kRuntimeFeatures = 0xFFFFFFFFu;
// arm_push_env.py:160-196 without its COOMM muscles: AnalyticalLinearDamper + ControllableFixConstraint
constexpr unsigned kFeaturesTaperedSuckerArm = SOFTROD_FEAT_ANALYTICAL_DAMPER | SOFTROD_FEAT_SUCKER_CONSTRAINT;
// Internal pseudo-feature (never in softrod_config.features): the contact plane's normal is
// exactly e_z, set by the host in RodParams.features and in the template mask.
constexpr unsigned kFeatPlaneZup = 1u << 30;
constexpr int kRuntimeEnv = -1;
// The fast kernel carries the COOMM muscle layers (softrod_muscle.hpp) in the instantiations compiled FOR a feature
// set that has them: OctoArmPush (SOFTROD_FEATURES_ARM_PUSH) and the clamped / free muscle rod of the known-answer
// tests; any other mix with muscles runs the LIBM kernel (softrod_create says so).
constexpr unsigned kFeaturesMuscleRod = SOFTROD_FEAT_FIXED_BC | SOFTROD_FEAT_ANALYTICAL_DAMPER | SOFTROD_FEAT_COOMM_MUSCLES;
template <unsigned F>
constexpr bool kMusclesCompiled = F != kRuntimeFeatures && (F & SOFTROD_FEAT_COOMM_MUSCLES) != 0;

// Wave-uniform parameters: passed by value, so they land in SGPRs (kernarg segment).
struct RodParams {
    int n_envs, n_elem, time_two_half_adds;
    unsigned features;
    int env_kind, filter_order, damp_before_constrain, pad0;
    double dt, half_dt, final_time;
    double rest_len, inv_rest_len;   // uniform straight rod: all elements alike
    double rest_vor, inv_rest_vor;
    double J[3], invJ[3];            // diagonal mass second moment of inertia
    double shear[3];                 // diag(ac*G*A, ac*G*A, E*A)
    double bend[3];                  // diag(E*I1, E*I2, G*I3) on Voronoi vertices
    double mass_node;                // interior node mass; the two end nodes carry half
    double gravity[3];
    double tip_force[3];
    double damp_t;                   // exp(-nu*dt)
    double damp_r[3];                // exp(-nu*dt*m_e/J_i)
    double damp_logr[3];             // -nu*dt*m_e/J_i  (fast path: exp(e*logr))
    double eps_length, eps_rot_axis, acos_shift, eps_sin;
    double two_acos_shift, neg_eps_sin;   // 2 acos_shift + 1e-200, -eps_sin (host-computed: the planar loop holds them in SGPRs)
    double base_limit, step_time;    // SoftPendulum3D set_action
    float base_step, control_penalty_coeff;
    // OctoArmSingle-v0
    int contact_before_forcing, n_action;
    double mass_total;               // sum of nodal masses (center of mass)
    double target[2], kappa_range[2], kappa_rate_range[2];
    double plane_origin[3], plane_normal[3];
    double contact_k, contact_nu, slip_tol, surface_tol;
    double kin_mu[3], stat_mu[3];
    double r0_sqrt_rest_len;         // radius_k = r0 sqrt(l_rest / l_k)  (volume preserving)
    // OctoFlat-v0: n_arm rods per wave, `seg` slots apart (0 = one rod per wave), + rigid head
    int seg, n_arm, seg_shift, head_fixed;     // head_fixed: OneEndFixedBC on the rigid body too (reach_env.py:126-130)
    double head_mass, head_invJ[3], head_radius;   // the planar head only ever turns about d3
    double head_center[3], joint_angle0, joint_angle_step;   // Cylinder centre at reset; FixedJoint2Rigid angle of arm a (degrees)
    // SoftArmTracking: the two spline muscles (muscle_torques_with_bspline.py:98-126)
    int n_ctrl, n_pieces;
    double muscle_scale, max_rate, base_length, arm_target[3];
    const double* spline;   // device: breaks[SOFTROD_MAX_SPLINE_PIECES + 1], then coef[p][j][4] (j < n_ctrl)
    double joint_k, joint_nu, joint_kt;
    // ControllableFixConstraint (octopus/controllable_constraint.py:24-69)
    int n_suckers, sucker_index[SOFTROD_MAX_SUCKERS], pad2;
    double sucker_ratio0;            // SuckerController.reduction_ratio as configured (a reset restores it for ArmPush)
    // clock table (StatePtrs.time_tab): entries, and 1 / (n_substeps dt) to find an env's entry
    int tab_len, tab_n_sub;
    double inv_step_time;
    // SOFTROD_FEAT_COOMM_MUSCLES (softrod_muscle.hpp): layer kinds, the force-length polynomial, the recalled-detail switches
    int n_muscles, muscle_kind[SOFTROD_MAX_MUSCLES], fl_degree, muscle_form, muscle_cur_radius, muscle_tm_law, push_mode, pad3;
    double fl_coef[SOFTROD_MAX_FL_COEF];
};

// Rows of the per-lane material table of a TAPERED rod (softrod_set_radius_profile): what
// CosseratRod.straight_rod derives from a per-element radius.  [row][slot], shared by all envs.
enum MatRow { kMatMass = 0, kMatShear01, kMatShear2, kMatBend01, kMatBend2, kMatJ0, kMatJ2, kMatInvJ0, kMatInvJ2,
              kMatDampLog0, kMatDampLog2, kMatDampR0, kMatDampR2, kMatR0s, kMatInvR0s, kMatRows = SOFTROD_MATERIAL_ROWS };

struct StatePtrs {
    double* pos;   // [3][N][64]
    double* vel;   // [3][N][64]
    double* dir;   // [9][N][64]
    double* omg;   // [3][N][64]
    double* tan;   // [3][N][64]  tangents as of the last force evaluation
    double* time;  // [N]
    double* bc;    // [12][N]     BC targets: pos0[3], Q0[9] of node/element 0
    double* ctrl;  // [4][N]      moving-base controller: position x,y ; velocity x,y
                   //             (ArmSingle: [0..1] = prev_com_state)
    double* kap;   // [3][N][64]  kappa as of the last force evaluation
    double* rkap;  // [3][N][64]  rest_kappa
    float* prev_action;   // [N][7]   the env's _prev_action (soft_pendulum.py:97-99,165)
    double* envmem;       // [N][64]  ArmSingle prev_kappa_state
    const double* basis;  // [(n_elem-1)][n_action]  rest_kappa[0,:] = basis @ action
    double* head;  // [20][N]  OctoFlat rigid head: x[3], v[3], Q[9], w[3], target[2]
    // device-side auto-reset (softrod_autoreset_enable), all nullptr when off
    uint8_t* needs_reset;   // [N] set by the step epilogue: terminated | truncated
    uint8_t* skip;          // [N] set by the auto-reset pass: this env was reset instead of stepped
    const double* queue;    // [depth][N][record]  pre-drawn reset records
    int* q_consumed;        // [N] records used so far
    const int* q_produced;  // [N] records staged so far
    int* q_underflow;       // [1] envs that needed a record when none was staged
    int q_depth, q_record;
    const struct RodParams* params;   // device copy of the kernel's RodParams (cold paths read it)
    const struct StatePtrs* self;     // device copy of this struct (cold paths read it)
    const double* time_tab; // [tab_len] the float64 clock after k env.steps from a reset, accumulated on the
                            // host exactly as PositionVerlet.step advances it (2 n_substeps additions of
                            // dt/2 per env.step, soft_pendulum.py:183-184); nullptr: none
    const double* mat;      // [kMatRows][64*EPL] material table of a tapered rod, or nullptr (uniform)
    double* sucker;         // [SOFTROD_MAX_SUCKERS][N] effective reduction ratio of each sucker
    int* sucker_idx;        // [SOFTROD_MAX_SUCKERS][N] SuckerController.index of each sucker (Python indexing)
    double* mact;           // [SOFTROD_MAX_MUSCLES][N][64*EPL] muscle activations per element, or nullptr
    const double* mtab;     // [SOFTROD_MAX_MUSCLES][4][64*EPL] ratio_position x, y, z and strength per element, or nullptr
    // the muscle octopus envs (softrod_mocto.hpp): `sucker` / `sucker_idx` are [SOFTROD_MAX_SUCKERS][N * n_arm] there
    double* aux;            // [8][N] target x, y, z; the head's x, y before the step
    float* prev_kappa;      // [N][n_arm * (n_elem - 1)] ArmTwoEnv._prev_kappa
};

// ---------------------------------------------------------------------------------
// cross-lane: full-wave shift by one lane (GFX9 DPP wave_shl / wave_shr)
// ---------------------------------------------------------------------------------
#ifdef SOFTROD_USE_BPERMUTE
__device__ __forceinline__ double from_next(double x) {
    double y = __shfl_down(x, 1);
    return (threadIdx.x & 63) == 63 ? 0.0 : y;
}
__device__ __forceinline__ double from_prev(double x) {
    double y = __shfl_up(x, 1);
    return (threadIdx.x & 63) == 0 ? 0.0 : y;
}
#else
// lane k receives lane k+1's value; lane 63 receives 0.
__device__ __forceinline__ double from_next(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x130, 0xf, 0xf, true);  // wave_shl:1, bound_ctrl:0 -> 0
    hi = __builtin_amdgcn_mov_dpp(hi, 0x130, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
// lane k receives lane k-1's value; lane 0 receives 0.
__device__ __forceinline__ double from_prev(double x) {
    int lo = __double2loint(x), hi = __double2hiint(x);
    lo = __builtin_amdgcn_mov_dpp(lo, 0x138, 0xf, 0xf, true);  // wave_shr:1, bound_ctrl:0 -> 0
    hi = __builtin_amdgcn_mov_dpp(hi, 0x138, 0xf, 0xf, true);
    return __hiloint2double(hi, lo);
}
#endif

__device__ __forceinline__ double wave_sum(double x) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) x += __shfl_xor(x, off);
    return x;  // fixed butterfly -> bitwise deterministic
}

template <unsigned F>
__device__ __forceinline__ bool has(const RodParams& P, unsigned bit) {
    if constexpr (F == kRuntimeFeatures) return (P.features & bit) != 0;
    else return (F & bit) != 0;
}
template <int E>
__device__ __forceinline__ int env_of(const RodParams& P) {
    if constexpr (E == kRuntimeEnv) return P.env_kind;
    else return E;
}
// ArmPullWeightEnv IS ArmPushEnv with another _build (octopus/arm_push_env.py:516-618): same set_action, step, get_state
__device__ __forceinline__ bool is_push_env(int env) { return env == SOFTROD_ENV_ARM_PUSH || env == SOFTROD_ENV_ARM_PULL_WEIGHT; }
// Index of a slot inside its rod.  One rod per wave: the slot index itself.  OctoFlat packs
// n_arm rods `seg` slots apart (seg a power of two; the slots between an arm's last node and
// the next arm's first are ghosts with zero stiffness — PyElastica's own memory-block idea;
// slots past the last arm get an index beyond every rod so that all validity tests fail).
// For OctoFlat `raw` is the slot index within the env's block of waves (threadIdx.x).
template <unsigned F = kRuntimeFeatures>
__device__ __forceinline__ int slot_local(const RodParams& P, int raw) {
    if constexpr (F != kRuntimeFeatures) {
        if constexpr ((F & SOFTROD_FEAT_OCTO_HEAD) != 0)
            return (raw & (P.seg - 1)) | (raw >= P.n_arm * P.seg ? (1 << 20) : 0);
        else return raw;
    } else {
        return P.seg ? ((raw & (P.seg - 1)) | (raw >= P.n_arm * P.seg ? (1 << 20) : 0)) : raw;
    }
}

// ---------------------------------------------------------------------------------
// per-lane register state (EPL slots per lane)
// ---------------------------------------------------------------------------------
template <int EPL>
struct LaneN {
    double x[EPL][3], v[EPL][3];
    double Q[EPL][9], w[EPL][3];
    double t[EPL][3];
    double kap[EPL][3], rk[EPL][3];
    // SOFTROD_FEAT_SPLINE_MUSCLE_TORQUES (wave-uniform; dead code elsewhere): points_cached of the
    // normal / binormal muscle, the control points of this env.step, and bits 0-1 "profile d
    // must be rebuilt", bits 2-3 initial_call_flag d.  rk[.][0..1] hold the two torque profiles.
    double pc[8];
    float pin[8];
    int mflag;
};

template <int EPL>
struct ConstN {
    double hx[EPL], hq[EPL];
    double cf[EPL], ca[EPL][3];
    double cw01[EPL], cw2[EPL];
    double s01[EPL], s2[EPL];
    double b01[EPL], bd[EPL];
    double mass[EPL], mass_next[EPL], inv_mass_pair[EPL];   // 1 / (m_k + m_{k+1}): element velocity
    double gm[EPL][3];                                      // gravity * mass: the weight the contact law sees (0 past the end, 0 if contact precedes forcing)
    // tapered rods only (TAPER instantiations; dead otherwise): the constants a uniform rod
    // keeps in scalar registers (RodParams), per element
    double j01[EPL], j2[EPL], dlog0[EPL], dlog2[EPL], dr0[EPL], dr2[EPL], r0s[EPL], ir0s[EPL];
    // SOFTROD_FEAT_COOMM_MUSCLES only (dead otherwise): the layers' position ratios and activation x strength
    double mr[EPL][SOFTROD_MAX_MUSCLES][3], amp[EPL][SOFTROD_MAX_MUSCLES];
    const double* mtab_lane;   // the same from memory (run-time-mask / LIBM kernels): StatePtrs.mtab / .mact at this lane's slot 0
    const double* mact_lane;
};

// value of index+1 / index-1 for a per-slot array
template <int EPL>
__device__ __forceinline__ void shift_next(const double (&a)[EPL], double (&o)[EPL]) {
#pragma unroll
    for (int s = 0; s + 1 < EPL; ++s) o[s] = a[s + 1];
    o[EPL - 1] = from_next(a[0]);
}
template <int EPL>
__device__ __forceinline__ void shift_prev(const double (&a)[EPL], double (&o)[EPL]) {
    o[0] = from_prev(a[EPL - 1]);
#pragma unroll
    for (int s = 1; s < EPL; ++s) o[s] = a[s - 1];
}

template <int EPL, unsigned F>
__device__ __forceinline__ void load_lane(const StatePtrs& S, size_t N, int rod, int lane, LaneN<EPL>& L) {
    constexpr size_t W = (size_t)kLanes * EPL;
    const size_t base = (size_t)rod * W + (size_t)lane * EPL;
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            L.x[s][c] = S.pos[c * N * W + base + s];
            L.v[s][c] = S.vel[c * N * W + base + s];
            L.w[s][c] = S.omg[c * N * W + base + s];
            L.t[s][c] = S.tan[c * N * W + base + s];
            if (F == kRuntimeFeatures || (F & (SOFTROD_FEAT_REST_KAPPA_ACTION | SOFTROD_FEAT_SPLINE_MUSCLE_TORQUES))) {
                L.kap[s][c] = S.kap[c * N * W + base + s];
                L.rk[s][c] = S.rkap[c * N * W + base + s];
            } else {
                L.kap[s][c] = 0.0;
                L.rk[s][c] = 0.0;
            }
        }
#pragma unroll
        for (int c = 0; c < 9; ++c) L.Q[s][c] = S.dir[c * N * W + base + s];
    }
}

template <int EPL, unsigned F>
__device__ __forceinline__ void store_lane(const StatePtrs& S, size_t N, int rod, int lane, const LaneN<EPL>& L) {
    constexpr size_t W = (size_t)kLanes * EPL;
    const size_t base = (size_t)rod * W + (size_t)lane * EPL;
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            S.pos[c * N * W + base + s] = L.x[s][c];
            S.vel[c * N * W + base + s] = L.v[s][c];
            S.omg[c * N * W + base + s] = L.w[s][c];
            S.tan[c * N * W + base + s] = L.t[s][c];
            if (F == kRuntimeFeatures || (F & (SOFTROD_FEAT_REST_KAPPA_ACTION | SOFTROD_FEAT_SPLINE_MUSCLE_TORQUES)))
                S.kap[c * N * W + base + s] = L.kap[s][c];
        }
#pragma unroll
        for (int c = 0; c < 9; ++c) S.dir[c * N * W + base + s] = L.Q[s][c];
    }
}


struct BcTargets {
    double pos[3];
    double Q[9];
    double vel[3];      // imposed base velocity (moving base), else 0
    double keep[SOFTROD_MAX_SUCKERS];   // 1 - effective reduction ratio of each sucker (this env)
    int snode[SOFTROD_MAX_SUCKERS], selem[SOFTROD_MAX_SUCKERS];   // the node / element each sucker holds (this env)
};

// Per-env action handling done once per env.step before the substeps.
struct EnvAction {
    float a[8];         // raw action (float32), as many as the env has
    double force;       // SoftPendulum: point_force[0] (float32 value held in float64)
    double mu[SOFTROD_MAX_MUSCLES];   // ArmPush: the activations set_action applied (uniform over the elements)
    bool mu_set;
};

__device__ __forceinline__ void load_bc(const StatePtrs& S, size_t N, int rod, BcTargets& B) {
#pragma unroll
    for (int i = 0; i < 3; ++i) B.pos[i] = S.bc[(size_t)i * N + rod];
#pragma unroll
    for (int i = 0; i < 9; ++i) B.Q[i] = S.bc[(size_t)(3 + i) * N + rod];
    B.vel[0] = B.vel[1] = B.vel[2] = 0.0;
#pragma unroll
    for (int j = 0; j < SOFTROD_MAX_SUCKERS; ++j) { B.keep[j] = 1.0; B.snode[j] = -1; B.selem[j] = -1; }
}

// SuckerController.index with Python indexing: i >= 0 holds node i and element i; i < 0 counts from the end of
// each array, velocity_collection has n + 1 columns and omega_collection n (arm_push_env.py:262: index = -1)
__device__ __forceinline__ void sucker_targets(const RodParams& P, int index, int& node, int& elem) {
    node = index >= 0 ? index : P.n_elem + 1 + index;
    elem = index >= 0 ? index : P.n_elem + index;
}

// ControllableFixConstraint.constrain_rates (octopus/controllable_constraint.py:45-69):
//   velocity_collection[..., index] *= 1.0 - reduction_ratio ; omega_collection[..., index] *= ...
// NOT idempotent (unless the ratio is 1), so unlike the other constraints it is applied once per
// substep only and never at kernel entry.
template <unsigned F>
__device__ __forceinline__ void load_suckers(const RodParams& P, const StatePtrs& S, size_t N, int rod, BcTargets& B) {
    if (has<F>(P, SOFTROD_FEAT_SUCKER_CONSTRAINT)) {
#pragma unroll
        for (int j = 0; j < SOFTROD_MAX_SUCKERS; ++j) {
            B.keep[j] = (j < P.n_suckers) ? 1.0 - S.sucker[(size_t)j * N + rod] : 1.0;
            if (j < P.n_suckers) sucker_targets(P, S.sucker_idx[(size_t)j * N + rod], B.snode[j], B.selem[j]);
        }
    }
}
template <unsigned F, int EPL>
__device__ __forceinline__ void sucker_rates_n(const RodParams& P, const BcTargets& B, int lane, LaneN<EPL>& L) {
    if (has<F>(P, SOFTROD_FEAT_SUCKER_CONSTRAINT)) {
#pragma unroll
        for (int j = 0; j < SOFTROD_MAX_SUCKERS; ++j) {
            if (j >= P.n_suckers) continue;
#pragma unroll
            for (int s = 0; s < EPL; ++s) {
                const double kn = (lane * EPL + s) == B.snode[j] ? B.keep[j] : 1.0;
                const double ke = (lane * EPL + s) == B.selem[j] ? B.keep[j] : 1.0;
#pragma unroll
                for (int c = 0; c < 3; ++c) { L.v[s][c] *= kn; L.w[s][c] *= ke; }
            }
        }
    }
}

// ---- boundary conditions act on node 0 / element 0 = lane 0, slot 0 -------------------------
template <unsigned F, int EPL>
__device__ __forceinline__ void constrain_values_n(const RodParams& P, const BcTargets& B, int lane,
                                                   LaneN<EPL>& L) {
    const bool l0 = (lane == 0);
    if (has<F>(P, SOFTROD_FEAT_PENDULUM_BC)) {
        L.x[0][1] = l0 ? B.pos[1] : L.x[0][1];
        L.x[0][2] = l0 ? B.pos[2] : L.x[0][2];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            L.Q[0][j] = l0 ? B.Q[j] : L.Q[0][j];
            L.Q[0][6 + j] = l0 ? B.Q[6 + j] : L.Q[0][6 + j];
        }
    }
    if (has<F>(P, SOFTROD_FEAT_FIXED_BC | SOFTROD_FEAT_MOVING_BASE_BC)) {
#pragma unroll
        for (int j = 0; j < 3; ++j) L.x[0][j] = l0 ? B.pos[j] : L.x[0][j];
#pragma unroll
        for (int j = 0; j < 9; ++j) L.Q[0][j] = l0 ? B.Q[j] : L.Q[0][j];
    }
}

template <unsigned F, int EPL>
__device__ __forceinline__ void constrain_rates_n(const RodParams& P, const BcTargets& B, int lane,
                                                  LaneN<EPL>& L) {
    const bool l0 = (lane == 0);
    if (has<F>(P, SOFTROD_FEAT_PENDULUM_BC)) {
        L.v[0][1] = l0 ? 0.0 : L.v[0][1];
        L.v[0][2] = l0 ? 0.0 : L.v[0][2];
        L.w[0][0] = l0 ? 0.0 : L.w[0][0];
        L.w[0][2] = l0 ? 0.0 : L.w[0][2];
    }
    if (has<F>(P, SOFTROD_FEAT_FIXED_BC | SOFTROD_FEAT_MOVING_BASE_BC)) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            L.v[0][j] = l0 ? B.vel[j] : L.v[0][j];
            L.w[0][j] = l0 ? 0.0 : L.w[0][j];
        }
    }
}

// ---- LaplaceDissipationFilter over the slot-interleaved arrays ---------------------------------
template <int EPL>
__device__ __forceinline__ void laplace_filter_n(double (&rate)[EPL], const double (&q)[EPL], int order) {
    double f[EPL], nx[EPL], pv[EPL];
#pragma unroll
    for (int s = 0; s < EPL; ++s) f[s] = rate[s];
    for (int i = 0; i < order; ++i) {
        shift_next<EPL>(f, nx);
        shift_prev<EPL>(f, pv);
#pragma unroll
        for (int s = 0; s < EPL; ++s) f[s] = ((-nx[s] - pv[s]) + 2.0 * f[s]) * q[s];
    }
#pragma unroll
    for (int s = 0; s < EPL; ++s) rate[s] = rate[s] - f[s];
}

template <int EPL>
__device__ __forceinline__ void laplace_filter_rates_n(const RodParams& P, int lane, LaneN<EPL>& L) {
    const int n = P.n_elem;
    double qn[EPL], qe[EPL];
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        const int idx = slot_local(P, lane * EPL + s);
        qn[s] = (idx >= 1 && idx <= n - 1) ? 0.25 : 0.0;
        qe[s] = (idx >= 1 && idx <= n - 2) ? 0.25 : 0.0;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double rv[EPL], rw[EPL];
#pragma unroll
        for (int s = 0; s < EPL; ++s) {
            const int idx = slot_local(P, lane * EPL + s);
            rv[s] = (idx <= n) ? L.v[s][c] : 0.0;
            rw[s] = (idx < n) ? L.w[s][c] : 0.0;
        }
        laplace_filter_n<EPL>(rv, qn, P.filter_order);
        laplace_filter_n<EPL>(rw, qe, P.filter_order);
#pragma unroll
        for (int s = 0; s < EPL; ++s) {
            const int idx = slot_local(P, lane * EPL + s);
            L.v[s][c] = (idx <= n) ? rv[s] : L.v[s][c];
            L.w[s][c] = (idx < n) ? rw[s] : L.w[s][c];
        }
    }
}

// The same filter for the reference's order 7 (soft_pendulum_3d/build.py:82-85) with six of the
// seven passes of the omega fields folded into ONE 13-tap stencil staged through LDS, while the
// velocity fields keep their DPP passes: the two halves run on different pipes (LDS / VALU)
// and overlap; all six fields through LDS, or none, measured 8 % slower (profiles/README.md).
//
// A pass is f <- q (2 f - S+ f - S- f) with q = 1/4 inside and 0 on the two boundary entries,
// i.e. after the first pass f vanishes on the boundary and every further pass is the free
// 3-point operator L = (2 - S+ - S-)/4 acting on the ODD extension of f about both
// boundaries (f(-j) = -f(j), f(N+j) = -f(N-j)): oddness keeps the boundary at zero and the
// interior sees exactly the masked passes.  So f_7 = L^6 f_1 = sum_j c_j f_1(k+j),
// c_j = (-1)^j C(12, 6+j) / 4^6.  Pass 1 runs in registers (DPP shifts) as before; f_1 and
// its reflections go to LDS once; each entry then reads its 13 taps.  Per field: 13 FMAs and
// 13 LDS reads instead of 6 x (4 DPP moves + 3 fp64 ops); the block is one wavefront, so
// the barrier between the writes and the reads costs nothing.  Needs N >= 6.
#ifndef SOFTROD_FILTER_V_LDS
#define SOFTROD_FILTER_V_LDS 0      // how many of the three v fields take the LDS stencil as well (from v_z down); -1: omega_3 on DPP too
#endif
template <int EPL>
__device__ __forceinline__ void laplace_filter_rates_lds7(const RodParams& P, int lane, LaneN<EPL>& L) {
    constexpr int M = 6, W = kLanes * EPL, KV = SOFTROD_FILTER_V_LDS, NL = 3 + KV, ND = 3 - KV;
    __shared__ double lds[NL][W + 2 * M];
    const int n = P.n_elem;
    // c_j for j = 0..6: C(12, 6+j) / 4096 with alternating sign
    constexpr double c[M + 1] = {924.0 / 4096.0, -792.0 / 4096.0, 495.0 / 4096.0, -220.0 / 4096.0,
                                 66.0 / 4096.0, -12.0 / 4096.0, 1.0 / 4096.0};
    // The masks of the first pass are factors (0.25 inside, 0 on the boundary entries and beyond),
    // not selects: the rates of the slots past the rod's end are exact zeros (sanitize_unused_rates
    // at kernel entry; nothing in a substep moves them), so a product with them is a zero, not a
    // NaN.  (Dropping the selects of the write-back as well — an entry whose correction is zero
    // keeps its value by subtracting that zero — is 24 instructions fewer and measured SLOWER,
    // 1.99 against 1.74 ms: profiles/README.md r2k.)
    double r[6][EPL], f1[6][EPL], q[2][EPL];
    bool inner[2][EPL];        // [0]: nodes 1..n-1, [1]: elements 1..n-2
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        const int idx = slot_local(P, lane * EPL + s);
        inner[0][s] = (idx >= 1 && idx <= n - 1);
        inner[1][s] = (idx >= 1 && idx <= n - 2);
        q[0][s] = inner[0][s] ? 0.25 : 0.0;
        q[1][s] = inner[1][s] ? 0.25 : 0.0;
#pragma unroll
        for (int c3 = 0; c3 < 3; ++c3) {
            r[c3][s] = L.v[s][c3]; r[3 + c3][s] = L.w[s][c3];
        }
    }
    // LDS row k holds field lfield(k): the omega fields, then v_z, v_y; the other v fields keep DPP
    auto lfield = [](int k) { return k < 3 ? 3 + k : 5 - k; };      // 3, 4, 5, 2, 1
    // pass 1 in registers: the fields that go to LDS first, the others while those writes are on
    // their way (no measurable difference to "all six, then the writes")
    auto pass1 = [&](int fld) {
        double nx[EPL], pv[EPL];
        shift_next<EPL>(r[fld], nx);
        shift_prev<EPL>(r[fld], pv);
#pragma unroll
        for (int s = 0; s < EPL; ++s) f1[fld][s] = ((-nx[s] - pv[s]) + 2.0 * r[fld][s]) * q[fld / 3][s];
    };
#pragma unroll
    for (int k = 0; k < NL; ++k) pass1(lfield(k));
    // stage f_1 and its odd reflections
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const int fld = lfield(k);
        const int N = (fld < 3) ? n : n - 1;
#pragma unroll
        for (int s = 0; s < EPL; ++s) {
            const int idx = slot_local(P, lane * EPL + s);
            if (idx <= N) lds[k][M + idx] = f1[fld][s];
            if (idx >= 1 && idx <= M) lds[k][M - idx] = -f1[fld][s];
            if (idx >= N - M && idx <= N - 1) lds[k][M + 2 * N - idx] = -f1[fld][s];
        }
    }
    // register (DPP) field k: the v fields, then omega_3 (KV = -1: only two omega fields in LDS)
    auto dfield = [](int k) { return k < 3 ? k : 8 - k; };       // 0, 1, 2, 5
#pragma unroll
    for (int k = 0; k < ND; ++k) pass1(dfield(k));
    __syncthreads();
    // The remaining six passes: in registers (DPP) for the first ND v fields, as the 13 taps out of
    // LDS for the others.  Interleaved — the taps of an LDS field are requested before the DPP
    // passes of a register field and summed after them, so that the LDS round trip runs under VALU
    // work of the same wave.
    double tap[M + 1][EPL], tpm[M + 1][EPL];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
        const int lf = lfield(k);
#pragma unroll
        for (int s = 0; s < EPL; ++s) {
            const double* row = &lds[k][lane * EPL + s];
            tap[0][s] = row[M];
#pragma unroll
            for (int j = 1; j <= M; ++j) { tap[j][s] = row[M + j]; tpm[j][s] = row[M - j]; }
        }
#pragma unroll
        for (int kd = k; kd < ND; kd += NL) {       // (more register fields than LDS fields: several per turn)
            const int fld = dfield(kd);
            double f[EPL], nx[EPL], pv[EPL];
#pragma unroll
            for (int s = 0; s < EPL; ++s) f[s] = f1[fld][s];
            for (int i = 0; i < 6; ++i) {
                shift_next<EPL>(f, nx);
                shift_prev<EPL>(f, pv);
#pragma unroll
                for (int s = 0; s < EPL; ++s) f[s] = ((-nx[s] - pv[s]) + 2.0 * f[s]) * q[fld / 3][s];
            }
#pragma unroll
            for (int s = 0; s < EPL; ++s) {
                const int idx = slot_local(P, lane * EPL + s);
                if (fld < 3) L.v[s][fld] = (idx <= n) ? r[fld][s] - f[s] : L.v[s][fld];
                else L.w[s][fld - 3] = (idx < n) ? r[fld][s] - f[s] : L.w[s][fld - 3];
            }
        }
#pragma unroll
        for (int s = 0; s < EPL; ++s) {
            double acc = c[0] * tap[0][s];
#pragma unroll
            for (int j = 1; j <= M; ++j) acc = fma(c[j], tap[j][s] + tpm[j][s], acc);
            // (a select: outside the interior the taps read LDS words nobody wrote)
            const double out = r[lf][s] - (inner[lf / 3][s] ? acc : 0.0);
            const int idx = slot_local(P, lane * EPL + s);
            if (lf < 3) L.v[s][lf] = (idx <= n) ? out : L.v[s][lf];
            else L.w[s][lf - 3] = (idx < n) ? out : L.w[s][lf - 3];
        }
    }
    __syncthreads();     // the next substep overwrites the staging rows
}

// Kernel entry, SOFTROD_FEAT_LAPLACE_FILTER: the rates in the slots past the rod's end are made
// the exact zeros the filter's mask-as-factor form relies on (they are zeros after any reset; this
// covers a state written through the state view).
template <int EPL>
__device__ __forceinline__ void sanitize_unused_rates(const RodParams& P, int lane, LaneN<EPL>& L) {
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        const int idx = slot_local(P, lane * EPL + s);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            L.v[s][c] = (idx <= P.n_elem) ? L.v[s][c] : 0.0;
            L.w[s][c] = (idx < P.n_elem) ? L.w[s][c] : 0.0;
        }
    }
}

// Kernel entry, SOFTROD_FEAT_PLANE_CONTACT_ANISO (fast math): the contact law keeps its masks as factors
// that vanish off the rod (zero normal force -> zero friction) instead of selects, and the slot
// past the last element feeds the tip node's element -> node average: a non-finite value written
// into the unused slots through the state view would turn those zero factors into NaN (0 x NaN).
// Positions, directors past the rod's end that are not finite become 0, the rates there exact zeros.
template <int EPL>
__device__ __forceinline__ void sanitize_unused_slots(const RodParams& P, int lane, LaneN<EPL>& L) {
    sanitize_unused_rates<EPL>(P, lane, L);
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        const int idx = slot_local(P, lane * EPL + s);
#pragma unroll
        for (int c = 0; c < 3; ++c)
            L.x[s][c] = (idx <= P.n_elem || fabs(L.x[s][c]) <= 1.0e300) ? L.x[s][c] : 0.0;
#pragma unroll
        for (int c = 0; c < 9; ++c)
            L.Q[s][c] = (idx < P.n_elem || fabs(L.Q[s][c]) <= 1.0e300) ? L.Q[s][c] : 0.0;
    }
}

template <int EPL>
__device__ __forceinline__ void laplace_filter_rates_fast(const RodParams& P, int lane, LaneN<EPL>& L) {
    if (P.filter_order == 7 && P.n_elem >= 8 && P.seg == 0) laplace_filter_rates_lds7<EPL>(P, lane, L);
    else laplace_filter_rates_n<EPL>(P, lane, L);
}

// ---- SOFTROD_MATH_FAST primitives (softrod_fast.hpp explains where they are used) ----
// 1/x: v_rcp_f64 seed (2^-23) + one third-order correction r (1 + e + e^2), e = 1 - x r:
// relative error e^3 ~ 2^-69 before the final rounding.
//
// Diagnostic builds (tools/fastmath_cost.sh; never the shipped library): SOFTROD_DIAG_IEEE_DIV
// replaces both by the correctly rounded IEEE division / square root, to measure what the
// third-order forms cost in parity horizon ("fast-math cost").
__device__ __forceinline__ double fast_rcp(double x) {
#ifdef SOFTROD_DIAG_IEEE_DIV
    return 1.0 / x;
#endif
    const double r = __builtin_amdgcn_rcp(x);
    const double e = fma(-x, r, 1.0);
    return fma(r, fma(e, e, e), r);
}
// 1/sqrt(x): v_rsq_f64 seed + one third-order correction: with r = (1 + d)/sqrt(x),
// e = 1/2 - x r^2/2 = -(d + d^2/2) and 1 + e + 3/2 e^2 = 1/(1 + d) + O(d^3).
__device__ __forceinline__ double fast_rsqrt(double x) {
#ifdef SOFTROD_DIAG_IEEE_DIV
    return 1.0 / sqrt(x);
#endif
    const double r = __builtin_amdgcn_rsq(x);
    const double e = fma(-(0.5 * x) * r, r, 0.5);
    return fma(r, e * fma(1.5, e, 1.0), r);
}

}  // namespace softrod
#include "softrod_contact.hpp"
#include "softrod_muscle.hpp"
namespace softrod {

// The OctoFlat kernel runs out of scalar registers (some 40 wave-uniform doubles live in its loop
// next to a dozen lane masks: 32 spilled words were reloaded with v_readlane every substep).  Its
// contact constants therefore live in LDS and are read where they are used — LDS reads issue
// beside the VALU, v_readlane issues ON it.  stage: once per workgroup before the first substep
// (a barrier follows in the caller); fetch: per substep.
#ifndef SOFTROD_OCTO_CONTACT_LDS
#define SOFTROD_OCTO_CONTACT_LDS 1
#endif
// (one function owns the array, so that every access keeps its LDS address space: handed out as a
// pointer it became a generic one, flat loads with a vector address per constant)
// (and ONE function, not two instantiations of a template: each would own an array of its own)
__device__ __forceinline__ ContactParams contact_params_lds_impl(const RodParams& P, const bool stage) {
    __shared__ double c[16];
    ContactParams C;
    if (stage) {
        if (threadIdx.x == 0) {
            c[0] = P.contact_k; c[1] = P.contact_nu; c[2] = P.slip_tol; c[3] = P.surface_tol;
            c[4] = P.kin_mu[0]; c[5] = P.kin_mu[1]; c[6] = P.kin_mu[2];
            c[7] = P.stat_mu[0]; c[8] = P.stat_mu[1]; c[9] = P.stat_mu[2];
            c[10] = P.r0_sqrt_rest_len; c[11] = 1.0 / P.r0_sqrt_rest_len; c[12] = P.plane_origin[2];
            c[4] = 0.5 * (P.kin_mu[0] + P.kin_mu[1]); c[5] = 0.5 * (P.kin_mu[0] - P.kin_mu[1]);      // mean, half difference
            c[7] = 0.5 * (P.stat_mu[0] + P.stat_mu[1]); c[8] = 0.5 * (P.stat_mu[0] - P.stat_mu[1]);
        }
        return C;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        C.origin[i] = P.plane_origin[i]; C.normal[i] = P.plane_normal[i];
        C.kin_mu[i] = c[4 + i]; C.stat_mu[i] = c[7 + i];     // ([0], [1]: see kin_am / stat_am; only [2] is read)
    }
    C.kin_am[0] = c[4]; C.kin_am[1] = c[5]; C.stat_am[0] = c[7]; C.stat_am[1] = c[8];
    C.origin[2] = c[12];
    C.k = c[0]; C.nu = c[1]; C.slip_tol = c[2]; C.surface_tol = c[3];
    C.r0_sqrt_rest_len = c[10];
    C.inv_r0_sqrt_rest_len = c[11];
    return C;
}
__device__ __forceinline__ void stage_contact_params(const RodParams& P) { (void)contact_params_lds_impl(P, true); }
__device__ __forceinline__ ContactParams contact_params_lds(const RodParams& P) { return contact_params_lds_impl(P, false); }

__device__ __forceinline__ ContactParams contact_params(const RodParams& P) {
    ContactParams C;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        C.origin[i] = P.plane_origin[i]; C.normal[i] = P.plane_normal[i];
        C.kin_mu[i] = P.kin_mu[i]; C.stat_mu[i] = P.stat_mu[i];
    }
    C.k = P.contact_k; C.nu = P.contact_nu; C.slip_tol = P.slip_tol; C.surface_tol = P.surface_tol;
    C.r0_sqrt_rest_len = P.r0_sqrt_rest_len;
    C.inv_r0_sqrt_rest_len = 1.0 / P.r0_sqrt_rest_len;
    C.kin_am[0] = 0.5 * (P.kin_mu[0] + P.kin_mu[1]); C.kin_am[1] = 0.5 * (P.kin_mu[0] - P.kin_mu[1]);
    C.stat_am[0] = 0.5 * (P.stat_mu[0] + P.stat_mu[1]); C.stat_am[1] = 0.5 * (P.stat_mu[0] - P.stat_mu[1]);
    return C;
}

// ---------------------------------------------------------------------------------
// env prologue (set_action) and epilogue (NaN check, reward, truncation, observation)
//   SoftPendulum    soft_pendulum.py:163-166,196-251
//   SoftPendulum3D  soft_pendulum_3d.py:99-113,130-174
//   ArmSingle       octopus/arm_single_env.py:186-235,252-316
// ---------------------------------------------------------------------------------
template <int EPL>
__device__ __forceinline__ void sum_tangents(const RodParams& P, int lane, const LaneN<EPL>& L, double tm[3]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double p = 0.0;
#pragma unroll
        for (int s = 0; s < EPL; ++s) p += (slot_local(P, lane * EPL + s) < P.n_elem) ? L.t[s][c] : 0.0;
        tm[c] = wave_sum(p) / (double)P.n_elem;
    }
}

template <int EPL>
__device__ __forceinline__ void com_xy_n(const RodParams& P, const ConstN<EPL>& C, int lane,
                                         const LaneN<EPL>& L, double com[2]) {
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        double p = 0.0;
#pragma unroll
        for (int s = 0; s < EPL; ++s)
            p += (slot_local(P, lane * EPL + s) <= P.n_elem) ? C.mass[s] * L.x[s][c] : 0.0;
        com[c] = wave_sum(p) / P.mass_total;
    }
}

// Per-env outputs.  pack = 0: separate arrays obs[N][od], reward[N], terminated[N],
// truncated[N].  pack = 1 (multi-GPU path): `obs` points at rows of ro + 4 32-bit words,
// ro = od rounded up to even (so the float64 reward is 8-byte aligned in every row):
//   [obs (od floats) | pad | reward (float64, 2 words) | terminated, truncated (bytes 0, 1) | 0]
// so that ONE all-gather moves everything and the receiver unpacks with views only.
__device__ __forceinline__ float* out_row(float* obs, int rod, int od, int pack) {
    return obs + (size_t)rod * (size_t)(pack ? od + (od & 1) + 4 : od);
}
__device__ __forceinline__ void emit_scalars(float* row, int od, int pack, int rod, double r, bool te,
                                             bool tr, double* __restrict__ reward,
                                             uint8_t* __restrict__ terminated,
                                             uint8_t* __restrict__ truncated,
                                             uint8_t* needs_reset = nullptr) {
    if (needs_reset) needs_reset[rod] = (te || tr) ? 1 : 0;
    if (pack) {
        const int ro = od + (od & 1);
        if (od & 1) row[od] = 0.0f;
        row[ro] = __int_as_float(__double2loint(r));
        row[ro + 1] = __int_as_float(__double2hiint(r));
        row[ro + 2] = __int_as_float((te ? 1 : 0) | ((tr ? 1 : 0) << 8));
        row[ro + 3] = 0.0f;
    } else {
        reward[rod] = r;
        terminated[rod] = te ? 1 : 0;
        truncated[rod] = tr ? 1 : 0;
    }
}

template <int EPL>
__device__ __forceinline__ void arm_get_state_n(const RodParams& P, const StatePtrs& S, size_t N, int rod,
                                                int lane, const ConstN<EPL>& C, const LaneN<EPL>& L,
                                                const float* pa, float* __restrict__ o) {
    constexpr size_t W = (size_t)kLanes * EPL;
    const int nv = P.n_elem - 1;
    double kap[EPL], rate[EPL];
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        const int idx = lane * EPL + s;
        const size_t m = (size_t)rod * W + idx;
        kap[s] = L.kap[s][0];
        rate[s] = kap[s] - S.envmem[m];
        if (idx < nv) S.envmem[m] = kap[s];
    }
    double mk[7], mr[7];
    int lo = 0;
#pragma unroll
    for (int b = 0; b < 7; ++b) {
        const int sz = nv / 7 + (b < nv % 7 ? 1 : 0);
        double pk = 0.0, pr = 0.0;
#pragma unroll
        for (int s = 0; s < EPL; ++s) {
            const int idx = lane * EPL + s;
            const bool in = idx < nv && idx >= lo && idx < lo + sz;
            pk += in ? kap[s] : 0.0;
            pr += in ? rate[s] : 0.0;
        }
        mk[b] = wave_sum(pk) / (double)sz;
        mr[b] = wave_sum(pr) / (double)sz;
        lo += sz;
    }
    double com[2];
    com_xy_n<EPL>(P, C, lane, L, com);
    if (lane == 0) {
        const double pc0 = S.ctrl[(size_t)0 * N + rod], pc1 = S.ctrl[(size_t)1 * N + rod];
        S.ctrl[(size_t)0 * N + rod] = com[0];
        S.ctrl[(size_t)1 * N + rod] = com[1];
#pragma unroll
        for (int b = 0; b < 7; ++b) {
            o[b] = (float)((mk[b] - P.kappa_range[0]) / (P.kappa_range[1] - P.kappa_range[0]));
            o[7 + b] = (float)((mr[b] - P.kappa_rate_range[0]) /
                               (P.kappa_rate_range[1] - P.kappa_rate_range[0]));
        }
        o[14] = (float)(com[0] - pc0);
        o[15] = (float)(com[1] - pc1);
#pragma unroll
        for (int i = 0; i < 7; ++i) o[16 + i] = pa[i];
        o[23] = (float)P.target[0];
        o[24] = (float)P.target[1];
    }
}

// SoftArmTrackingEnv.get_state (soft_arm/soft_arm_tracking.py:160-207): segment means of
// kappa[0], kappa[1] as of the last force evaluation, scaled by base_length / 2 pi; the tip
// position / base_length; the target / 1000.  st: [2 n_ctrl + 6], valid in every lane.
template <int EPL>
__device__ __forceinline__ void soft_arm_get_state_n(const RodParams& P, const StatePtrs& S, size_t N, int rod,
                                                     int lane, const LaneN<EPL>& L, double (&st)[14]) {
    const int n = P.n_elem, ns = P.n_ctrl, nv = n - 1, avg = nv / ns;
#pragma unroll
    for (int i = 0; i < 14; ++i) st[i] = 0.0;
#pragma unroll
    for (int c = 0; c < 2; ++c)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (i >= ns) continue;
            const int lo = avg * i, hi = (i == ns - 1) ? nv : avg * (i + 1);
            double part = 0.0;
#pragma unroll
            for (int s = 0; s < EPL; ++s) {
                const int idx = lane * EPL + s;
                part += (idx >= lo && idx < hi) ? L.kap[s][c] : 0.0;
            }
            st[c * ns + i] = (wave_sum(part) / (double)(hi - lo)) * P.base_length / (2.0 * M_PI);
        }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double xs = 0.0;
#pragma unroll
        for (int s = 0; s < EPL; ++s) xs = (n % EPL == s) ? L.x[s][c] : xs;
        st[2 * ns + c] = __shfl(xs, n / EPL) / P.base_length;
        st[2 * ns + 3 + c] = S.ctrl[(size_t)(1 + c) * N + rod] / 1000.0;
    }
}

template <int EPL>
__device__ __forceinline__ double theta_n(const RodParams& P, int lane, const LaneN<EPL>& L) {
    double tm[3];
    sum_tangents<EPL>(P, lane, L, tm);
    const double th = atan(tm[0] / tm[1]);
    const double two_pi = 2.0 * M_PI;
    double m = fmod(th + M_PI, two_pi);
    if (m != 0.0 && m < 0.0) m += two_pi;
    return m - M_PI;
}

template <int EPL>
__device__ __forceinline__ double tilt_n(const RodParams& P, int lane, const LaneN<EPL>& L) {
    double tm[3];
    sum_tangents<EPL>(P, lane, L, tm);
    const double nrm = sqrt(tm[0] * tm[0] + tm[1] * tm[1] + tm[2] * tm[2]);
    return acos(fmin(fmax(tm[2] / nrm, -1.0), 1.0));
}

// observation width of the single-rod envs (softrod_obs_dim)
__device__ __forceinline__ int env_obs_dim(const RodParams& P) {
    return (P.env_kind == SOFTROD_ENV_SOFTPENDULUM3D) ? 9 : (P.env_kind == SOFTROD_ENV_ARM_SINGLE) ? 25
         : (P.env_kind == SOFTROD_ENV_SOFT_ARM) ? 2 * P.n_ctrl + 6
         : is_push_env(P.env_kind) ? 2 * (P.n_elem + 1) + 2 : 4;
}

// ArmPushEnv.get_state (octopus/arm_push_env.py:225-245): position_collection[0], velocity_collection[0], then
// np.eye(2)[previous_action] (discrete) or the previous action (continuous); float32.  Every lane writes its own
// node's two entries.  -> does any entry hold a NaN (`np.any(np.isnan(states))`, :335)?  `fix`: np.nan_to_num.
template <int EPL>
__device__ __forceinline__ bool push_get_state_n(const RodParams& P, int lane, const LaneN<EPL>& L, const float* pa,
                                                 float* __restrict__ o, bool fix) {
    const int n = P.n_elem;
    float tail[2];
    if (P.push_mode == 0) { const int a = (int)pa[0]; tail[0] = a == 0 ? 1.0f : 0.0f; tail[1] = a == 0 ? 0.0f : 1.0f; }
    else { tail[0] = pa[0]; tail[1] = pa[1]; }
    bool bad = isnan(tail[0]) || isnan(tail[1]);
#pragma unroll
    for (int s = 0; s < EPL; ++s)
        bad = bad || ((lane * EPL + s) <= n && (isnan(L.x[s][0]) || isnan(L.v[s][0])));
    bad = __any(bad);
    auto clean = [&](float v) {
        if (!(fix && bad)) return v;
        return isnan(v) ? 0.0f : (isinf(v) ? copysignf(3.4028234663852886e38f, v) : v);
    };
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        const int idx = lane * EPL + s;
        if (idx <= n) { o[idx] = clean((float)L.x[s][0]); o[n + 1 + idx] = clean((float)L.v[s][0]); }
    }
    if (lane == 0) { o[2 * n + 2] = clean(tail[0]); o[2 * n + 3] = clean(tail[1]); }
    return bad;
}

template <int E, int EPL>
__device__ __forceinline__ void env_observe_n(const RodParams& P, const StatePtrs& S, size_t N, int rod,
                                              int lane, const ConstN<EPL>& C, const LaneN<EPL>& L,
                                              const float* pa, float* __restrict__ o) {
    // o: this rod's observation row
    const int env = env_of<E>(P);
    if (env == SOFTROD_ENV_SOFTPENDULUM3D) {
        const double tilt = tilt_n<EPL>(P, lane, L);
        if (lane == 0) {
            o[0] = (float)L.x[0][0]; o[1] = (float)L.x[0][1]; o[2] = (float)L.x[0][2];
            o[3] = (float)L.v[0][0]; o[4] = (float)L.v[0][1]; o[5] = (float)L.v[0][2];
            o[6] = pa[0]; o[7] = pa[1];
            o[8] = (float)tilt;
        }
    } else if (env == SOFTROD_ENV_ARM_SINGLE) {
        arm_get_state_n<EPL>(P, S, N, rod, lane, C, L, pa, o);
    } else if (env == SOFTROD_ENV_SOFT_ARM) {
        double st[14];
        soft_arm_get_state_n<EPL>(P, S, N, rod, lane, L, st);
        if (lane == 0)
            for (int i = 0; i < 2 * P.n_ctrl + 6; ++i) o[i] = (float)st[i];
    } else if (is_push_env(env)) {
        (void)push_get_state_n<EPL>(P, lane, L, pa, o, false);
    } else {
        const double th = theta_n<EPL>(P, lane, L);
        if (lane == 0) {
            o[0] = (float)L.x[0][0];
            o[1] = (float)L.v[0][0];
            o[2] = pa[0];
            o[3] = (float)th;
        }
    }
}

template <int E, int EPL>
__device__ __forceinline__ void env_epilogue_n(const RodParams& P, const StatePtrs& S, size_t N, int rod,
                                               int lane, const ConstN<EPL>& C, const LaneN<EPL>& L,
                                               double time, const EnvAction& A, float* __restrict__ obs,
                                               double* __restrict__ reward,
                                               uint8_t* __restrict__ terminated,
                                               uint8_t* __restrict__ truncated,
                                               double* __restrict__ aux, const int pack) {
    bool bad = false;
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        bool b = false;
#pragma unroll
        for (int c = 0; c < 3; ++c) b = b || isnan(L.x[s][c]) || isnan(L.v[s][c]);
        bad = bad || ((slot_local(P, lane * EPL + s) <= P.n_elem) && b);
    }
    const bool invalid = __any(bad);
    const int env = env_of<E>(P);
    if (lane == 0 && env != SOFTROD_ENV_SOFT_ARM) {   // set_action: self._prev_action[:] = action
#pragma unroll
        for (int i = 0; i < 7; ++i) S.prev_action[7 * (size_t)rod + i] = A.a[i];
    }
    if (env == SOFTROD_ENV_SOFTPENDULUM3D) {
        const double tilt = tilt_n<EPL>(P, lane, L);
        if (lane == 0) {
            const double bx = S.ctrl[(size_t)0 * N + rod], by = S.ctrl[(size_t)1 * N + rod];
            const double base_distance = sqrt(bx * bx + by * by);
            const float ctl = 1e-3f * (A.a[0] * A.a[0] + A.a[1] * A.a[1]);
            double r = -(tilt * tilt + 0.1 * (base_distance * base_distance) + (double)ctl);
            if (invalid) r = -50.0;
            float* o = out_row(obs, rod, 9, pack);
            // '>=' here, '>' in SoftPendulum
            emit_scalars(o, 9, pack, rod, r, invalid, time >= P.final_time, reward, terminated, truncated,
                         S.needs_reset);
            if (aux) aux[rod] = tilt;
            o[0] = (float)L.x[0][0]; o[1] = (float)L.x[0][1]; o[2] = (float)L.x[0][2];
            o[3] = (float)L.v[0][0]; o[4] = (float)L.v[0][1]; o[5] = (float)L.v[0][2];
            o[6] = A.a[0]; o[7] = A.a[1];
            o[8] = (float)tilt;
        }
    } else if (env == SOFTROD_ENV_SOFT_ARM) {
        // step(), soft_arm_tracking.py:226-259
        double st[14];
        soft_arm_get_state_n<EPL>(P, S, N, rod, lane, L, st);
        const int od = 2 * P.n_ctrl + 6;
        if (lane == 0) {
            double d2 = 0.0;
            bool nan = false;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const double d = (st[2 * P.n_ctrl + 3 + c] * 1000.0 - st[2 * P.n_ctrl + c] * P.base_length) / 1000.0;
                d2 += d * d;
            }
            for (int i = 0; i < od; ++i) nan = nan || isnan(st[i]);
            const double nrm = sqrt(d2);
            double r = -(nrm * nrm);
            if (nan) r = -100.0;
            float* o = out_row(obs, rod, od, pack);
            const double tick = S.ctrl[(size_t)0 * N + rod];
            emit_scalars(o, od, pack, rod, r, nan, tick * P.dt >= P.final_time, reward, terminated, truncated,
                         S.needs_reset);
            for (int i = 0; i < od; ++i) {
                double v = st[i];
                if (nan) v = isnan(v) ? 0.0 : (isinf(v) ? copysign(1.7976931348623157e308, v) : v);   // np.nan_to_num
                o[i] = (float)v;
            }
        }
    } else if (is_push_env(env)) {
        // ArmPushEnv.step after the loop (octopus/arm_push_env.py:288-347).  _isnan_check covers position,
        // velocity, director, alpha, omega and the centre of mass (:298-309; alpha = J^-1 tau e of the last substep
        // is NaN only where omega became NaN in that substep)
        bool qbad = false;
#pragma unroll
        for (int s = 0; s < EPL; ++s) {
            bool q = false;
#pragma unroll
            for (int c = 0; c < 3; ++c) q = q || isnan(L.w[s][c]);
#pragma unroll
            for (int c = 0; c < 9; ++c) q = q || isnan(L.Q[s][c]);
            qbad = qbad || ((lane * EPL + s) < P.n_elem && q);
        }
        double com[2];
        com_xy_n<EPL>(P, C, lane, L, com);
        const bool nan_state = invalid || __any(qbad) || isnan(com[0]) || isnan(com[1]);
        const int od = 2 * (P.n_elem + 1) + 2;
        float* o = out_row(obs, rod, od, pack);
        const bool nan_obs = push_get_state_n<EPL>(P, lane, L, A.a, o, true);
        if (lane == 0) {
            const double p0 = S.ctrl[(size_t)0 * N + rod], p1 = S.ctrl[(size_t)1 * N + rod];   // prev_cm_pos (prologue)
            double forward = 0.0, survive = 0.0;
            bool term = false;
            if (nan_state) { term = true; survive = -20.0; }
            else forward = sqrt(com[0] * com[0] + com[1] * com[1]) - sqrt(p0 * p0 + p1 * p1);
            double rw = forward + survive;
            if (isnan(rw)) { term = true; rw = -20.0; }
            if (nan_obs) { term = true; rw = -20.0; }
            emit_scalars(o, od, pack, rod, rw, term, time > P.final_time, reward, terminated, truncated, S.needs_reset);
        }
    } else if (env == SOFTROD_ENV_ARM_SINGLE) {
        double pw = 0.0;
#pragma unroll
        for (int s = 0; s < EPL; ++s)
            pw += (slot_local(P, lane * EPL + s) < P.n_elem)
                      ? L.w[s][0] * L.w[s][0] + L.w[s][1] * L.w[s][1] + L.w[s][2] * L.w[s][2] : 0.0;
        const bool blown = invalid || (sqrt(wave_sum(pw)) > 250.0);
        double com[2];
        com_xy_n<EPL>(P, C, lane, L, com);
        if (lane == 0) {
            float sq = 0.0f;
#pragma unroll
            for (int i = 0; i < 7; ++i) sq += A.a[i] * A.a[i];
            const float pen = P.control_penalty_coeff * (sq / 7.0f);
            double forward = 0.0, survive = 0.0;
            bool term = false;
            if (blown) { term = true; survive = -1.0; }
            else {
                const double dx = com[0] - P.target[0], dy = com[1] - P.target[1];
                const double dist = sqrt(dx * dx + dy * dy);
                forward = exp(-dist / 0.35) - 0.096;
                if (dist < 0.1) { survive = 5.0; term = true; }
            }
            // blown: forward_reward is still the Python float 0.0, so the reference's sum
            // `0.0 - np.float32 + (-1.0)` is float32 arithmetic (NumPy 2 promotion)
            const double rw = blown ? (double)((0.0f - pen) + (-1.0f)) : forward - (double)pen + survive;
            emit_scalars(out_row(obs, rod, 25, pack), 25, pack, rod, rw, term,
                         time > P.final_time, reward, terminated, truncated, S.needs_reset);
        }
        arm_get_state_n<EPL>(P, S, N, rod, lane, C, L, A.a, out_row(obs, rod, 25, pack));
    } else {
        const double th = theta_n<EPL>(P, lane, L);
        if (lane == 0) {
            double forward = 0.0, survive = 0.0;
            if (invalid) survive = -50.0;
            else forward = fabs(L.x[0][0]) * 10.0 + th * th;
            float* o = out_row(obs, rod, 4, pack);
            emit_scalars(o, 4, pack, rod, forward - 0.0 + survive, invalid, time > P.final_time, reward,
                         terminated, truncated, S.needs_reset);
            o[0] = (float)L.x[0][0];
            o[1] = (float)L.v[0][0];
            o[2] = A.a[0];
            o[3] = (float)th;
        }
    }
}

template <unsigned F, int EPL, bool TAPER = false>
__device__ __forceinline__ void build_const(const RodParams& P, int lane, const EnvAction& A,
                                            ConstN<EPL>& C, const double* __restrict__ mat = nullptr) {
    const int n = P.n_elem;
    const bool damp = has<F>(P, SOFTROD_FEAT_ANALYTICAL_DAMPER);
    const double ct = damp ? P.damp_t : 1.0;
    constexpr int W = kLanes * EPL;
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        const int raw = lane * EPL + s;
        const int idx = slot_local<F>(P, raw);
        const bool first = (idx == 0);
        const bool held_q = first && has<F>(P, SOFTROD_FEAT_PENDULUM_BC | SOFTROD_FEAT_FIXED_BC |
                                               SOFTROD_FEAT_MOVING_BASE_BC);
        const bool held_x = first && has<F>(P, SOFTROD_FEAT_FIXED_BC | SOFTROD_FEAT_MOVING_BASE_BC);
        const bool node_valid = idx <= n, elem_valid = idx < n, vor_valid = idx < n - 1;
        double mass = (idx == 0 || idx == n) ? 0.5 * P.mass_node : P.mass_node;
        double mass_next = (idx + 1 == n) ? 0.5 * P.mass_node : P.mass_node;
        double shear01 = P.shear[0], shear2 = P.shear[2], bend01 = P.bend[0], bend2 = P.bend[2];
        double invJ0 = P.invJ[0], invJ2 = P.invJ[2];
        if constexpr (TAPER) {      // CosseratRod.straight_rod with an array of radii: per-element constants
            // (the table is one wave wide; the arms of a multi-wave env repeat it: softrod_set_radius_profile)
            const int wide = raw;
            const int raw = wide & (W - 1);
            const int nx = raw + 1 < W ? raw + 1 : raw;
            mass = node_valid ? mat[kMatMass * W + raw] : 1.0;
            mass_next = (idx + 1 <= n) ? mat[kMatMass * W + nx] : 1.0;
            shear01 = mat[kMatShear01 * W + raw]; shear2 = mat[kMatShear2 * W + raw];
            bend01 = mat[kMatBend01 * W + raw]; bend2 = mat[kMatBend2 * W + raw];
            invJ0 = mat[kMatInvJ0 * W + raw]; invJ2 = mat[kMatInvJ2 * W + raw];
            C.j01[s] = mat[kMatJ0 * W + raw]; C.j2[s] = mat[kMatJ2 * W + raw];
            C.dlog0[s] = mat[kMatDampLog0 * W + raw]; C.dlog2[s] = mat[kMatDampLog2 * W + raw];
            C.dr0[s] = mat[kMatDampR0 * W + raw]; C.dr2[s] = mat[kMatDampR2 * W + raw];
            C.r0s[s] = mat[kMatR0s * W + raw]; C.ir0s[s] = mat[kMatInvR0s * W + raw];
        } else {
            C.j01[s] = P.J[0]; C.j2[s] = P.J[2];
            C.dlog0[s] = P.damp_logr[0]; C.dlog2[s] = P.damp_logr[2];
            C.dr0[s] = P.damp_r[0]; C.dr2[s] = P.damp_r[2];
            C.r0s[s] = P.r0_sqrt_rest_len; C.ir0s[s] = 1.0 / P.r0_sqrt_rest_len;
        }
        // masses of slots past the rod's end are zeros (so that a product with them needs no select);
        // 1 / (m_k + m_{k+1}) stays finite there
        C.mass[s] = node_valid ? mass : 0.0;
        C.mass_next[s] = (idx + 1 <= n) ? mass_next : 0.0;
        C.inv_mass_pair[s] = 1.0 / (mass + mass_next);
#pragma unroll
        for (int c = 0; c < 3; ++c)      // (zero when the contact operator runs BEFORE the forcing: it then sees no weight)
            C.gm[s][c] = (has<F>(P, SOFTROD_FEAT_GRAVITY) && !P.contact_before_forcing) ? P.gravity[c] * C.mass[s] : 0.0;
        C.hx[s] = held_x ? 0.0 : 1.0;
        C.hq[s] = held_q ? 0.0 : 1.0;
        const double cdm = node_valid ? ct * P.dt / mass : 0.0;
        C.cf[s] = cdm;
        double fe0 = 0.0, fe1 = 0.0, fe2 = 0.0;
        if (has<F>(P, SOFTROD_FEAT_GRAVITY)) {
            fe0 = P.gravity[0] * mass; fe1 = P.gravity[1] * mass; fe2 = P.gravity[2] * mass;
        }
        if (has<F>(P, SOFTROD_FEAT_POINT_FORCE_NODE0_X)) fe0 = first ? A.force : fe0;
        if (has<F>(P, SOFTROD_FEAT_TIP_FORCE) && idx == n) {
            fe0 += P.tip_force[0]; fe1 += P.tip_force[1]; fe2 += P.tip_force[2];
        }
        C.ca[s][0] = cdm * fe0; C.ca[s][1] = cdm * fe1; C.ca[s][2] = cdm * fe2;
        C.cw01[s] = elem_valid ? P.dt * invJ0 : 0.0;
        C.cw2[s] = elem_valid ? P.dt * invJ2 : 0.0;
        C.s01[s] = elem_valid ? shear01 : 0.0;
        C.s2[s] = elem_valid ? shear2 : 0.0;
        C.b01[s] = vor_valid ? bend01 : 0.0;
        C.bd[s] = vor_valid ? bend2 - bend01 : 0.0;
    }
}

template <unsigned F, int E, int EPL>
__device__ __forceinline__ void set_action_n(const RodParams& P, const StatePtrs& S, size_t N, int rod,
                                             int lane, const float* __restrict__ actions, EnvAction& A,
                                             BcTargets& B, LaneN<EPL>& L) {
    constexpr size_t W = (size_t)kLanes * EPL;
#pragma unroll
    for (int i = 0; i < 8; ++i) A.a[i] = 0.0f;
    A.force = 0.0;
#pragma unroll
    for (int m = 0; m < SOFTROD_MAX_MUSCLES; ++m) A.mu[m] = 0.0;
    A.mu_set = false;
    const int env = env_of<E>(P);
    if (is_push_env(env)) {
        // ArmPushEnv.set_action (octopus/arm_push_env.py:247-274): the sucker's index and the layers' activations
        if (actions) {
            int index;
            if (P.push_mode == 0) {
                A.a[0] = actions[rod];
                const bool hold_base = (int)A.a[0] == 0;
                index = hold_base ? 0 : -1;                              // :257,262
                A.mu[0] = hold_base ? -0.0 * 1.0 : 0.0;                  // :258-260, :263-265
                A.mu[1] = 0.0;
                A.mu[2] = hold_base ? 0.5 * 1.0 : 0.0;
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    if (m >= P.n_muscles || !S.mact) continue;
#pragma unroll
                    for (int s = 0; s < EPL; ++s) S.mact[((size_t)m * N + rod) * W + lane * EPL + s] = A.mu[m];
                }
            } else {
                A.a[0] = actions[2 * (size_t)rod]; A.a[1] = actions[2 * (size_t)rod + 1];
                // int(np.clip(location * self.n_elem, 0, self.n_elem - 1)): np.float32 * int stays float32 (:270)
                const float loc = fminf(fmaxf(A.a[0] * (float)P.n_elem, 0.0f), (float)(P.n_elem - 1));
                index = (int)loc;
                // only the transverse layer is written (:271); the others keep what they hold (zeros after a reset)
#pragma unroll
                for (int m = 0; m < 3; ++m) {
                    if (m >= P.n_muscles || !S.mact) continue;
                    A.mu[m] = (m == 2) ? (double)A.a[1] : S.mact[((size_t)m * N + rod) * W];
                }
                if (P.n_muscles > 2 && S.mact) {
#pragma unroll
                    for (int s = 0; s < EPL; ++s) S.mact[((size_t)2 * N + rod) * W + lane * EPL + s] = A.mu[2];
                }
            }
            A.mu_set = true;
            sucker_targets(P, index, B.snode[0], B.selem[0]);
            if (lane == 0) S.sucker_idx[rod] = index;
        }
    } else if (env == SOFTROD_ENV_SOFTPENDULUM3D) {
        if (actions) { A.a[0] = actions[2 * rod]; A.a[1] = actions[2 * rod + 1]; }
        if (has<F>(P, SOFTROD_FEAT_MOVING_BASE_BC)) {
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const double pos = S.ctrl[(size_t)i * N + rod];
                double vel = S.ctrl[(size_t)(2 + i) * N + rod];
                double next = pos;
                if (actions) {
                    const float disp = P.base_step * A.a[i];
                    next = fmin(fmax(pos + (double)disp, -P.base_limit), P.base_limit);
                    vel = (next - pos) / P.step_time;
                    if (lane == 0) {
                        S.ctrl[(size_t)i * N + rod] = next;
                        S.ctrl[(size_t)(2 + i) * N + rod] = vel;
                    }
                }
                B.pos[i] = next;
                B.vel[i] = vel;
            }
        }
    } else if (env == SOFTROD_ENV_ARM_SINGLE) {
        if (actions) {
#pragma unroll
            for (int i = 0; i < 7; ++i) A.a[i] = actions[7 * (size_t)rod + i];
            if (has<F>(P, SOFTROD_FEAT_REST_KAPPA_ACTION)) {
#pragma unroll
                for (int s = 0; s < EPL; ++s) {
                    const int idx = lane * EPL + s;
                    double rk0 = 0.0;
                    if (idx < P.n_elem - 1) {
                        const double* wrow = S.basis + (size_t)idx * 7;
#pragma unroll
                        for (int j = 0; j < 7; ++j) rk0 += wrow[j] * (double)A.a[j];
                    }
                    L.rk[s][0] = rk0;
                    S.rkap[(size_t)rod * W + idx] = rk0;
                }
            }
        }
    } else if (env == SOFTROD_ENV_SOFT_ARM) {
        // step(): spline_points_func_array_*[:] = action halves (soft_arm_tracking.py:211-216)
        if (actions) {
#pragma unroll
            for (int i = 0; i < 8; ++i) A.a[i] = (i < 2 * P.n_ctrl) ? actions[(size_t)(2 * P.n_ctrl) * rod + i] : 0.0f;
        }
    } else {
        if (actions) A.a[0] = actions[rod];
        A.force = (double)A.a[0];
    }
}

// ArmPushEnv.step: `prev_cm_pos = self.shearable_rod.compute_position_center_of_mass()[:2]` before the loop
// (octopus/arm_push_env.py:280), kept in the control rows [0..1] for the epilogue's reward.  Only when the
// launch integrates (n_sub > 0): with n_substeps = 0 the step is the epilogue alone on the resident state and
// prev_cm_pos is whatever the control rows hold (fixture replay, tests/test_gpu_muscle_fixtures.py).
template <unsigned F, int E, int EPL>
__device__ __forceinline__ void push_store_prev_com(const RodParams& P, const StatePtrs& S, size_t N, int rod, int lane,
                                                    const LaneN<EPL>& L, int n_sub) {
    if (!is_push_env(env_of<E>(P)) || n_sub <= 0) return;
    ConstN<EPL> C0;
    EnvAction A0;
#pragma unroll
    for (int i = 0; i < 8; ++i) A0.a[i] = 0.0f;
    A0.force = 0.0;
    A0.mu_set = false;
    if (S.mat) build_const<F, EPL, true>(P, lane, A0, C0, S.mat);
    else build_const<F, EPL>(P, lane, A0, C0);
    double com[2];
    com_xy_n<EPL>(P, C0, lane, L, com);
    if (lane == 0) { S.ctrl[(size_t)0 * N + rod] = com[0]; S.ctrl[(size_t)1 * N + rod] = com[1]; }
}

// ---- SoftArmTracking: MuscleTorquesWithVaryingBetaSplines x 2 --------------------------------
// State of the two forcing objects between launches lives in the env-memory row:
// [0 .. 2 n_ctrl) points_cached, [8 .. 8 + 2 n_ctrl) the control points last set, [16], [17]
// initial_call_flag; the torque profiles in rest-kappa rows 0 and 1.
template <int EPL>
__device__ __forceinline__ void muscle_load(const RodParams& P, const StatePtrs& S, int rod, bool have_action,
                                            const EnvAction& A, LaneN<EPL>& L) {
    const double* row = S.envmem + (size_t)rod * kLanes * EPL;
    const int nc = P.n_ctrl;
    int flag = 0;
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        bool same = true;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const bool on = j < nc;
            L.pc[4 * d + j] = on ? row[d * nc + j] : 0.0;
            L.pin[4 * d + j] = on ? (have_action ? A.a[d * nc + j] : (float)row[8 + d * nc + j]) : 0.0f;
            same = same && (L.pc[4 * d + j] == (double)L.pin[4 * d + j]);
        }
        const bool init = row[16 + d] != 0.0;
        flag |= (init ? 4 : 0) << d;
        flag |= ((!same || !init) ? 1 : 0) << d;          // muscle_torques_with_bspline.py:137-140
    }
    L.mflag = flag;
}
template <int EPL>
__device__ __forceinline__ void muscle_store(const RodParams& P, const StatePtrs& S, int rod, int lane,
                                             const LaneN<EPL>& L) {
    double* row = S.envmem + (size_t)rod * kLanes * EPL;
    const int nc = P.n_ctrl;
    if (lane == 0) {
#pragma unroll
        for (int d = 0; d < 2; ++d) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                if (j < nc) { row[d * nc + j] = L.pc[4 * d + j]; row[8 + d * nc + j] = (double)L.pin[4 * d + j]; }
            row[16 + d] = ((L.mflag >> (2 + d)) & 1) ? 1.0 : 0.0;
        }
    }
    constexpr size_t W = (size_t)kLanes * EPL;
    const size_t N = (size_t)P.n_envs;
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        const size_t m = (size_t)rod * W + (size_t)lane * EPL + s;
        S.rkap[0 * N * W + m] = L.rk[s][0];
        S.rkap[1 * N * W + m] = L.rk[s][1];
    }
}
// apply_torques' rebuild branch (:137-158): rate filter on points_cached, then the interpolant
// (piecewise-cubic form, softrod_set_spline_table) at np.cumsum(system.lengths).  Rare — once
// per env.step, twice when the filter's rounding leaves points_cached one ulp off the input.
template <int EPL>
__device__ __forceinline__ void spline_muscle_rebuild(const RodParams& P, int lane, const double (&len)[EPL],
                                                   LaneN<EPL>& L) {
    const int n = P.n_elem, nc = P.n_ctrl, np = P.n_pieces;
    double run[EPL], tot = 0.0;
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        tot += (lane * EPL + s < n) ? len[s] : 0.0;
        run[s] = tot;
    }
    double inc = tot;
#pragma unroll
    for (int off = 1; off < kLanes; off <<= 1) {
        const double y = __shfl_up(inc, off);
        inc += (lane >= off) ? y : 0.0;
    }
    const double excl = inc - tot;
    const double* breaks = P.spline;
    const double* coef = P.spline + (SOFTROD_MAX_SPLINE_PIECES + 1);
#pragma unroll
    for (int d = 0; d < 2; ++d) {
        if (!((L.mflag >> d) & 1)) continue;
        bool same = true;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const double in = (double)L.pin[4 * d + j];
            const double diff = in - L.pc[4 * d + j];
            const double sg = (double)((diff > 0.0) - (diff < 0.0));
            L.pc[4 * d + j] += sg * fmin(P.max_rate, fabs(diff));              // filter_activation, :221-225
            same = same && (L.pc[4 * d + j] == in);
        }
#pragma unroll
        for (int s = 0; s < EPL; ++s) {
            const double cum = excl + run[s];
            int p = 0;
            for (int q = 1; q < np; ++q) p += (cum >= breaks[q]) ? 1 : 0;
            const double ds = cum - breaks[p];
            double val = 0.0;
            for (int j = 0; j < nc; ++j) {
                const double* c = coef + (size_t)(p * nc + j) * 4;
                val += L.pc[4 * d + j] * fma(fma(fma(c[3], ds, c[2]), ds, c[1]), ds, c[0]);
            }
            L.rk[s][d] = (lane * EPL + s < n) ? P.muscle_scale * val : 0.0;     // :156-158
        }
        L.mflag = (L.mflag & ~(1 << d)) | (4 << d) | ((same ? 0 : 1) << d);
    }
}

// =================================================================================
// SOFTROD_MATH_LIBM: the substep as written by PyElastica
// =================================================================================

// overload_operator_kinematic_numba + _get_rotation_matrix:
//   x += h v ;  Q <- R(h w) Q  with R the transposed Rodrigues matrix.
__device__ __forceinline__ void libm_kinematic_step(const RodParams& P, double h, LaneN<1>& L) {
#pragma unroll
    for (int i = 0; i < 3; ++i) L.x[0][i] += h * L.v[0][i];
    // _get_rotation_matrix(scale = h, omega): axis = omega / (|omega| + eps), angle = h |omega|
    double a0 = L.w[0][0], a1 = L.w[0][1], a2 = L.w[0][2];
    const double th0 = sqrt(a0 * a0 + a1 * a1 + a2 * a2);
    const double den = th0 + P.eps_rot_axis;
    a0 /= den; a1 /= den; a2 /= den;
    const double th = th0 * h;
    double up, cs;
    sincos(th, &up, &cs);
    const double usq = 1.0 - cs;
    double R[9];
    R[0] = 1.0 - usq * (a1 * a1 + a2 * a2);
    R[4] = 1.0 - usq * (a0 * a0 + a2 * a2);
    R[8] = 1.0 - usq * (a0 * a0 + a1 * a1);
    R[1] = up * a2 + usq * a0 * a1;  R[3] = -up * a2 + usq * a0 * a1;
    R[2] = -up * a1 + usq * a0 * a2; R[6] = up * a1 + usq * a0 * a2;
    R[5] = up * a0 + usq * a1 * a2;  R[7] = -up * a0 + usq * a1 * a2;
    double Qn[9];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            Qn[i * 3 + j] = R[i * 3 + 0] * L.Q[0][0 * 3 + j] + R[i * 3 + 1] * L.Q[0][1 * 3 + j] +
                            R[i * 3 + 2] * L.Q[0][2 * 3 + j];
#pragma unroll
    for (int i = 0; i < 9; ++i) L.Q[0][i] = Qn[i];
}

// The material constants of this lane's element / Voronoi vertex: RodParams for a uniform rod,
// the per-lane table for a tapered one (softrod_set_radius_profile).
struct LibmMat {
    double shear[3], bend[3], J[3], invJ[3], damp_r[3], mass_next, r0s;
};
__device__ __forceinline__ void libm_material(const RodParams& P, const double* __restrict__ mat, int lane,
                                              LibmMat& M) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        M.shear[i] = P.shear[i]; M.bend[i] = P.bend[i]; M.J[i] = P.J[i]; M.invJ[i] = P.invJ[i];
        M.damp_r[i] = P.damp_r[i];
    }
    M.mass_next = (lane + 1 == P.n_elem) ? 0.5 * P.mass_node : P.mass_node;
    M.r0s = P.r0_sqrt_rest_len;
    if (mat) {
        constexpr int W = kLanes;
        const int nx = lane + 1 < W ? lane + 1 : lane;
        M.shear[0] = M.shear[1] = mat[kMatShear01 * W + lane]; M.shear[2] = mat[kMatShear2 * W + lane];
        M.bend[0] = M.bend[1] = mat[kMatBend01 * W + lane]; M.bend[2] = mat[kMatBend2 * W + lane];
        M.J[0] = M.J[1] = mat[kMatJ0 * W + lane]; M.J[2] = mat[kMatJ2 * W + lane];
        M.invJ[0] = M.invJ[1] = mat[kMatInvJ0 * W + lane]; M.invJ[2] = mat[kMatInvJ2 * W + lane];
        M.damp_r[0] = M.damp_r[1] = mat[kMatDampR0 * W + lane]; M.damp_r[2] = mat[kMatDampR2 * W + lane];
        M.mass_next = mat[kMatMass * W + nx];
        M.r0s = mat[kMatR0s * W + lane];
    }
}

// Internal forces/torques + forcing + dynamic update + dampers + rate constraints:
// steps (3)-(6) of the substep (DESIGN.md "substep order").
__device__ __forceinline__ void libm_dynamic_step(const RodParams& P, const LibmMat& M, const ConstN<1>& CM,
                                                  const BcTargets& B, int lane, double action, double mass,
                                                  LaneN<1>& L) {
    const int n = P.n_elem;
    const bool node_valid = lane <= n;
    const bool elem_valid = lane < n;
    const bool vor_valid = lane < n - 1;

    // ---- geometry: lengths, tangents, dilatation (_compute_all_dilatations) ----
    const double xn0 = from_next(L.x[0][0]), xn1 = from_next(L.x[0][1]), xn2 = from_next(L.x[0][2]);
    const double d0 = xn0 - L.x[0][0], d1 = xn1 - L.x[0][1], d2 = xn2 - L.x[0][2];
    const double len = sqrt(d0 * d0 + d1 * d1 + d2 * d2) + P.eps_length;
    L.t[0][0] = d0 / len; L.t[0][1] = d1 / len; L.t[0][2] = d2 / len;
    const double e = len / P.rest_len;

    // ---- shear/stretch: sigma = e Q t - z ; n = S sigma ----
    const double qt0 = L.Q[0][0] * L.t[0][0] + L.Q[0][1] * L.t[0][1] + L.Q[0][2] * L.t[0][2];
    const double qt1 = L.Q[0][3] * L.t[0][0] + L.Q[0][4] * L.t[0][1] + L.Q[0][5] * L.t[0][2];
    const double qt2 = L.Q[0][6] * L.t[0][0] + L.Q[0][7] * L.t[0][1] + L.Q[0][8] * L.t[0][2];
    const double n0 = M.shear[0] * (e * qt0);
    const double n1 = M.shear[1] * (e * qt1);
    const double n2 = M.shear[2] * (e * qt2 - 1.0);

    // ---- internal force: difference of Q^T n / e ----
    double cs0 = (L.Q[0][0] * n0 + L.Q[0][3] * n1 + L.Q[0][6] * n2) / e;
    double cs1 = (L.Q[0][1] * n0 + L.Q[0][4] * n1 + L.Q[0][7] * n2) / e;
    double cs2 = (L.Q[0][2] * n0 + L.Q[0][5] * n1 + L.Q[0][8] * n2) / e;
    cs0 = elem_valid ? cs0 : 0.0;
    cs1 = elem_valid ? cs1 : 0.0;
    cs2 = elem_valid ? cs2 : 0.0;
    const double f0 = cs0 - from_prev(cs0);
    const double f1 = cs1 - from_prev(cs1);
    const double f2 = cs2 - from_prev(cs2);

    // ---- bend/twist: kappa = -log(Q_{k+1} Q_k^T)/D  (_inv_rotate) ----
    double Qn[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) Qn[i] = from_next(L.Q[0][i]);
    const double len_n = from_next(len);
#define SR_ROWDOT(i, j) (Qn[3 * (i)] * L.Q[0][3 * (j)] + Qn[3 * (i) + 1] * L.Q[0][3 * (j) + 1] + \
                         Qn[3 * (i) + 2] * L.Q[0][3 * (j) + 2])
    const double vec0 = SR_ROWDOT(2, 1) - SR_ROWDOT(1, 2);
    const double vec1 = SR_ROWDOT(0, 2) - SR_ROWDOT(2, 0);
    const double vec2 = SR_ROWDOT(1, 0) - SR_ROWDOT(0, 1);
    const double trace = (SR_ROWDOT(0, 0) + SR_ROWDOT(1, 1)) + SR_ROWDOT(2, 2);
#undef SR_ROWDOT
    const double theta = acos(0.5 * trace - 0.5 - P.acos_shift);
    const double fk = (-0.5 * theta / sin(theta + P.eps_sin)) / P.rest_vor;
    const double k0 = vec0 * fk, k1 = vec1 * fk, k2 = vec2 * fk;
    L.kap[0][0] = k0; L.kap[0][1] = k1; L.kap[0][2] = k2;
    const double m0 = M.bend[0] * (k0 - L.rk[0][0]), m1 = M.bend[1] * (k1 - L.rk[0][1]),
                 m2 = M.bend[2] * (k2 - L.rk[0][2]);
    const double vd = 0.5 * (len_n + len) / P.rest_vor;
    const double e3 = 1.0 / (vd * vd * vd);
    double c20 = m0 * e3, c21 = m1 * e3, c22 = m2 * e3;
    const double dv3 = P.rest_vor * e3;
    double c30 = (k1 * m2 - k2 * m1) * dv3;
    double c31 = (k2 * m0 - k0 * m2) * dv3;
    double c32 = (k0 * m1 - k1 * m0) * dv3;
    c20 = vor_valid ? c20 : 0.0; c21 = vor_valid ? c21 : 0.0; c22 = vor_valid ? c22 : 0.0;
    c30 = vor_valid ? c30 : 0.0; c31 = vor_valid ? c31 : 0.0; c32 = vor_valid ? c32 : 0.0;
    // difference + trapezoid, Voronoi -> element (zeros beyond both ends do the end rules)
    double tq0 = (c20 - from_prev(c20)) + 0.5 * (c30 + from_prev(c30));
    double tq1 = (c21 - from_prev(c21)) + 0.5 * (c31 + from_prev(c31));
    double tq2 = (c22 - from_prev(c22)) + 0.5 * (c32 + from_prev(c32));

    // shear/stretch couple (Q t) x n * l_rest
    tq0 += (qt1 * n2 - qt2 * n1) * P.rest_len;
    tq1 += (qt2 * n0 - qt0 * n2) * P.rest_len;
    tq2 += (qt0 * n1 - qt1 * n0) * P.rest_len;

    // transport (J w / e) x w and unsteady dilatation (J w / e) (de/dt) / e
    const double vn0 = from_next(L.v[0][0]), vn1 = from_next(L.v[0][1]), vn2 = from_next(L.v[0][2]);
    const double rv = (L.x[0][0] * L.v[0][0] + L.x[0][1] * L.v[0][1]) + L.x[0][2] * L.v[0][2];
    const double rvn = (xn0 * vn0 + xn1 * vn1) + xn2 * vn2;
    const double rp1v = (xn0 * L.v[0][0] + xn1 * L.v[0][1]) + xn2 * L.v[0][2];
    const double rvp1 = (L.x[0][0] * vn0 + L.x[0][1] * vn1) + L.x[0][2] * vn2;
    const double dil_rate = (rv + rvn - rvp1 - rp1v) / len / P.rest_len;
    const double jw0 = M.J[0] * L.w[0][0] / e, jw1 = M.J[1] * L.w[0][1] / e, jw2 = M.J[2] * L.w[0][2] / e;
    tq0 += jw1 * L.w[0][2] - jw2 * L.w[0][1];
    tq1 += jw2 * L.w[0][0] - jw0 * L.w[0][2];
    tq2 += jw0 * L.w[0][1] - jw1 * L.w[0][0];
    tq0 += jw0 * dil_rate / e; tq1 += jw1 * dil_rate / e; tq2 += jw2 * dil_rate / e;

    // ---- synchronize(): operators in registration order.  Forcing: gravity, then the
    // point force ASSIGNS F_ext[0,0]; contact (octopus/build.py:274-283) after it unless
    // contact_before_forcing. ----
    double fe0 = 0.0, fe1 = 0.0, fe2 = 0.0;
    const bool has_contact = (P.features & SOFTROD_FEAT_PLANE_CONTACT_ANISO) != 0;
    const double mass_next = M.mass_next;
    const double xn[1][3] = {{xn0, xn1, xn2}}, vn[1][3] = {{vn0, vn1, vn2}};
    ConstN<1> CK;
    ContactParams CP = contact_params(P);
    CP.r0_sqrt_rest_len = M.r0s;
    CP.inv_r0_sqrt_rest_len = 1.0 / M.r0s;
    CK.mass[0] = mass;
    CK.mass_next[0] = mass_next;
    CK.inv_mass_pair[0] = 1.0 / (mass + mass_next);
    const double len1[1] = {len};
    if (has_contact && P.contact_before_forcing) {
        const double F[1][3] = {{f0, f1, f2}};
        double tq[1][3] = {{tq0, tq1, tq2}}, fc[1][3];
        plane_contact_n<1, false, false>(CP, P, lane, CK, L, xn, vn, len1, F, tq, fc);
        fe0 = fc[0][0]; fe1 = fc[0][1]; fe2 = fc[0][2];
        tq0 = tq[0][0]; tq1 = tq[0][1]; tq2 = tq[0][2];
    }
    if (P.features & SOFTROD_FEAT_GRAVITY) {
        fe0 += P.gravity[0] * mass; fe1 += P.gravity[1] * mass; fe2 += P.gravity[2] * mass;
    }
    if (P.features & SOFTROD_FEAT_POINT_FORCE_NODE0_X) fe0 = (lane == 0) ? action : fe0;
    if (P.features & SOFTROD_FEAT_TIP_FORCE) {
        const bool tip = (lane == n);
        fe0 += tip ? P.tip_force[0] : 0.0;
        fe1 += tip ? P.tip_force[1] : 0.0;
        fe2 += tip ? P.tip_force[2] : 0.0;
    }
    if (P.features & SOFTROD_FEAT_COOMM_MUSCLES) {       // ApplyMuscles: sqrt and divisions as written (FASTM = false)
        const double ea[1] = {e}, ila[1] = {1.0 / len}, qta[1][3] = {{qt0, qt1, qt2}};
        const double kva[1][3] = {{vor_valid ? k0 : 0.0, vor_valid ? k1 : 0.0, vor_valid ? k2 : 0.0}};
        const double e3a[1] = {e3}, r0a[1] = {M.r0s};
        double fm[1][3] = {{0.0, 0.0, 0.0}}, tm[1][3] = {{0.0, 0.0, 0.0}};
        muscle_loads_n<1, false, false>(P, CM, lane, L, ea, ila, qta, kva, e3a, r0a, fm, tm);
        fe0 += fm[0][0]; fe1 += fm[0][1]; fe2 += fm[0][2];
        tq0 += tm[0][0]; tq1 += tm[0][1]; tq2 += tm[0][2];
    }
    if (has_contact && !P.contact_before_forcing) {
        const double F[1][3] = {{f0 + (node_valid ? fe0 : 0.0), f1 + (node_valid ? fe1 : 0.0),
                                 f2 + (node_valid ? fe2 : 0.0)}};
        double tq[1][3] = {{tq0, tq1, tq2}}, fc[1][3];
        plane_contact_n<1, false, false>(CP, P, lane, CK, L, xn, vn, len1, F, tq, fc);
        fe0 += fc[0][0]; fe1 += fc[0][1]; fe2 += fc[0][2];
        tq0 = tq[0][0]; tq1 = tq[0][1]; tq2 = tq[0][2];
    }

    // ---- accelerations and rate update (v += dt a ; w += dt alpha) ----
    const double a0 = (f0 + fe0) / mass, a1 = (f1 + fe1) / mass, a2 = (f2 + fe2) / mass;
    const double al0 = (M.invJ[0] * tq0) * e, al1 = (M.invJ[1] * tq1) * e, al2 = (M.invJ[2] * tq2) * e;
    L.v[0][0] += node_valid ? P.dt * a0 : 0.0;
    L.v[0][1] += node_valid ? P.dt * a1 : 0.0;
    L.v[0][2] += node_valid ? P.dt * a2 : 0.0;
    L.w[0][0] += elem_valid ? P.dt * al0 : 0.0;
    L.w[0][1] += elem_valid ? P.dt * al1 : 0.0;
    L.w[0][2] += elem_valid ? P.dt * al2 : 0.0;

    // ---- dampers (registration order) and constrain_rates ----
    if (!P.damp_before_constrain) {
        constrain_rates_n<kRuntimeFeatures, 1>(P, B, lane, L);
        sucker_rates_n<kRuntimeFeatures, 1>(P, B, lane, L);
    }
    if (P.features & SOFTROD_FEAT_ANALYTICAL_DAMPER) {
        L.v[0][0] *= P.damp_t; L.v[0][1] *= P.damp_t; L.v[0][2] *= P.damp_t;
        L.w[0][0] *= pow(M.damp_r[0], e);
        L.w[0][1] *= pow(M.damp_r[1], e);
        L.w[0][2] *= pow(M.damp_r[2], e);
    }
    if (P.features & SOFTROD_FEAT_LAPLACE_FILTER) laplace_filter_rates_n<1>(P, lane, L);
    if (P.damp_before_constrain) {
        constrain_rates_n<kRuntimeFeatures, 1>(P, B, lane, L);
        sucker_rates_n<kRuntimeFeatures, 1>(P, B, lane, L);
    }
}

// ---------------------------------------------------------------------------------
// LIBM kernel: one env.step (or `n_sub` bare substeps) for every rod of the shard.
// grid = n_envs blocks of one wavefront; one node per lane (n_elem <= 63).
// ---------------------------------------------------------------------------------
// SOFTROD_LIBM_WAVES: waves per SIMD the register allocator leaves room for.  1: the kernel takes the ~390 registers
// it asks for (256 VGPRs + 138 AGPRs, no scratch); 2: capped at 256, the rest spills (measured: profiles/README.md r6).
#ifndef SOFTROD_LIBM_WAVES
#define SOFTROD_LIBM_WAVES 1
#endif
__global__ void __launch_bounds__(kLanes, SOFTROD_LIBM_WAVES)
softrod_step_libm_kernel(const RodParams P, const StatePtrs S, const float* __restrict__ actions,
                         float* __restrict__ obs, double* __restrict__ reward,
                         uint8_t* __restrict__ terminated, uint8_t* __restrict__ truncated,
                         double* __restrict__ aux, const int n_sub, const int epilogue, const int pack) {
    const int rod = blockIdx.x;
    const int lane = threadIdx.x;
    const size_t N = (size_t)P.n_envs;
    if (epilogue && S.skip && S.skip[rod]) {   // reset by the auto-reset pass of this env.step
        if (lane == 0) S.skip[rod] = 0;
        return;
    }

    LaneN<1> L;
    load_lane<1, kRuntimeFeatures>(S, N, rod, lane, L);
    BcTargets B;
    load_bc(S, N, rod, B);
    load_suckers<kRuntimeFeatures>(P, S, N, rod, B);
    EnvAction A;
    set_action_n<kRuntimeFeatures, kRuntimeEnv, 1>(P, S, N, rod, lane, actions, A, B, L);
    if (epilogue) push_store_prev_com<kRuntimeFeatures, kRuntimeEnv, 1>(P, S, N, rod, lane, L, n_sub);
    ConstN<1> C;
    if (S.mat) build_const<kRuntimeFeatures, 1, true>(P, lane, A, C, S.mat);
    else build_const<kRuntimeFeatures, 1>(P, lane, A, C);
    build_muscle_const<kRuntimeFeatures, 1, false>(P, S, N, rod, lane, A, C);
    LibmMat M;
    libm_material(P, S.mat, lane, M);

    double time = S.time[rod];
    const double mass = C.mass[0];

    for (int s = 0; s < n_sub; ++s) {
        libm_kinematic_step(P, P.half_dt, L);
        if (P.time_two_half_adds) time += P.half_dt;
        constrain_values_n<kRuntimeFeatures, 1>(P, B, lane, L);
        libm_dynamic_step(P, M, C, B, lane, A.force, mass, L);
        libm_kinematic_step(P, P.half_dt, L);
        time += P.time_two_half_adds ? P.half_dt : P.dt;
        constrain_values_n<kRuntimeFeatures, 1>(P, B, lane, L);
    }

    store_lane<1, kRuntimeFeatures>(S, N, rod, lane, L);
    if (lane == 0) S.time[rod] = time;
    if (epilogue)
        env_epilogue_n<kRuntimeEnv, 1>(P, S, N, rod, lane, C, L, time, A, obs, reward, terminated, truncated,
                                       aux, pack);
}

// Reset: expand the host-computed straight-rod description of each masked rod
// (CosseratRod.straight_rod, build.py:54-61) into the SoA rows.
//   init[rod][0..2] start, [3..5] step = (end-start)/n, [6..8] end, [9..17] Q rows
struct ResetArgs {
    const double* init;   // [N][18]
    const uint8_t* mask;  // [N] or nullptr
};

template <int EPL>
__global__ void __launch_bounds__(kLanes)
softrod_observe_kernel(const RodParams P, const StatePtrs S, const float* __restrict__ prev_action,
                            float* __restrict__ obs) {
    const int rod = blockIdx.x;
    const int lane = threadIdx.x;
    const size_t N = (size_t)P.n_envs;
    LaneN<EPL> L;
    load_lane<EPL, kRuntimeFeatures>(S, N, rod, lane, L);
    EnvAction A;
#pragma unroll
    for (int i = 0; i < 7; ++i) A.a[i] = 0.0f;
    A.force = 0.0;
    A.mu_set = false;
    ConstN<EPL> C;
    if (S.mat) build_const<kRuntimeFeatures, EPL, true>(P, lane, A, C, S.mat);
    else build_const<kRuntimeFeatures, EPL>(P, lane, A, C);
    const int adim = (P.env_kind == SOFTROD_ENV_SOFTPENDULUM3D) ? 2
                   : (P.env_kind == SOFTROD_ENV_ARM_SINGLE) ? 7 : (P.env_kind == SOFTROD_ENV_SOFT_ARM) ? 0
                   : is_push_env(P.env_kind) ? (P.push_mode == 0 ? 1 : 2) : 1;
    float pa[7] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    for (int i = 0; i < adim; ++i)
        pa[i] = prev_action ? prev_action[adim * (size_t)rod + i] : S.prev_action[7 * (size_t)rod + i];
    const int od = env_obs_dim(P);
    env_observe_n<kRuntimeEnv, EPL>(P, S, N, rod, lane, C, L, pa, obs + (size_t)od * rod);
}

// One rod's reset from its 18-double record; leaves the fresh state in L as well.
template <int EPL>
__device__ __forceinline__ void reset_rod(const RodParams& P, const StatePtrs& S, size_t N, int rod, int lane,
                                          const double* __restrict__ in, LaneN<EPL>& L) {
    constexpr size_t W = (size_t)kLanes * EPL;
    const int n = P.n_elem;
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        const int idx = lane * EPL + s;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double xv = in[c] + (double)idx * in[3 + c];
            if (idx == n) xv = in[6 + c];
            L.x[s][c] = xv;
            L.v[s][c] = 0.0; L.w[s][c] = 0.0; L.kap[s][c] = 0.0; L.rk[s][c] = 0.0;
        }
#pragma unroll
        for (int c = 0; c < 9; ++c) L.Q[s][c] = in[9 + c];
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double a[EPL], o[EPL];
#pragma unroll
        for (int s = 0; s < EPL; ++s) a[s] = L.x[s][c];
        shift_next<EPL>(a, o);
#pragma unroll
        for (int s = 0; s < EPL; ++s) L.t[s][c] = o[s] - L.x[s][c];
    }
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        const double len = sqrt(L.t[s][0] * L.t[s][0] + L.t[s][1] * L.t[s][1] + L.t[s][2] * L.t[s][2]) +
                           P.eps_length;
#pragma unroll
        for (int c = 0; c < 3; ++c) L.t[s][c] /= len;
    }
    store_lane<EPL, kRuntimeFeatures>(S, N, rod, lane, L);
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        const size_t m = (size_t)rod * W + (size_t)lane * EPL + s;
#pragma unroll
        for (int c = 0; c < 3; ++c) S.rkap[c * N * W + m] = 0.0;
        S.envmem[m] = 0.0;
    }
    double com[2] = {0.0, 0.0};
    if (P.env_kind == SOFTROD_ENV_ARM_SINGLE) {
        EnvAction A0;
#pragma unroll
        for (int i = 0; i < 7; ++i) A0.a[i] = 0.0f;
        A0.force = 0.0;
        ConstN<EPL> C;
        if (S.mat) build_const<kRuntimeFeatures, EPL, true>(P, lane, A0, C, S.mat);
        else build_const<kRuntimeFeatures, EPL>(P, lane, A0, C);
        com_xy_n<EPL>(P, C, lane, L, com);
    }
    if (P.features & SOFTROD_FEAT_COOMM_MUSCLES) {
        // a reset builds fresh muscle objects (activation zero) and, for ArmPush, a fresh SuckerController(index=0)
        // that is switched on after finalize (arm_push_env.py:187-195,205,222)
        if (S.mact) {
#pragma unroll
            for (int m = 0; m < SOFTROD_MAX_MUSCLES; ++m)
#pragma unroll
                for (int s = 0; s < EPL; ++s) S.mact[((size_t)m * N + rod) * W + (size_t)lane * EPL + s] = 0.0;
        }
    }
    if (lane == 0 && is_push_env(P.env_kind)) {
        S.sucker_idx[rod] = P.sucker_index[0];
        S.sucker[rod] = P.sucker_ratio0;
    }
    if (lane == 0) {
        S.time[rod] = 0.0;
        if (S.needs_reset) S.needs_reset[rod] = 0;
        if (P.env_kind == SOFTROD_ENV_SOFTPENDULUM3D) {   // _prev_action.fill(0), soft_pendulum_3d.py:68
#pragma unroll
            for (int i = 0; i < 7; ++i) S.prev_action[7 * (size_t)rod + i] = 0.0f;
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) S.bc[(size_t)i * N + rod] = in[i];
#pragma unroll
        for (int i = 0; i < 9; ++i) S.bc[(size_t)(3 + i) * N + rod] = in[9 + i];
#pragma unroll
        for (int i = 0; i < 4; ++i) S.ctrl[(size_t)i * N + rod] = (i < 2) ? com[i] : 0.0;
        if (P.env_kind == SOFTROD_ENV_SOFT_ARM) {   // tick = 0, wsol[0] (soft_arm_tracking.py:386,407-412)
#pragma unroll
            for (int i = 0; i < 3; ++i) S.ctrl[(size_t)(1 + i) * N + rod] = P.arm_target[i];
        }
    }
}



template <int EPL>
__global__ void __launch_bounds__(kLanes)
softrod_reset_kernel(const RodParams P, const StatePtrs S, const ResetArgs A) {
    const int rod = blockIdx.x;
    if (A.mask && !A.mask[rod]) return;
    LaneN<EPL> L;
    reset_rod<EPL>(P, S, (size_t)P.n_envs, rod, threadIdx.x, A.init + (size_t)rod * 18, L);
}

// softrod_queue_push*: the records staged by one push, compacted on the host, scattered into the
// resident ring [depth][N][record] (only what changed crosses PCIe: a push used to upload the
// whole ring, 9.4 MB at 4096 envs x 16 records, and stall the stream for 0.4 ms).
__global__ void __launch_bounds__(kLanes)
softrod_queue_scatter_kernel(double* __restrict__ queue, const double* __restrict__ staged,
                             const int2* __restrict__ where, const int n_envs, const int record) {
    const int m = blockIdx.x;
    const int2 w = where[m];                      // (env, slot)
    double* dst = queue + ((size_t)w.y * (size_t)n_envs + (size_t)w.x) * (size_t)record;
    const double* src = staged + (size_t)m * (size_t)record;
    for (int i = threadIdx.x; i < record; i += kLanes) dst[i] = src[i];
}

// Device-side auto-reset pass, launched before the step kernel when softrod_autoreset_enable
// is on (Gymnasium-1.0 VectorEnv NEXT_STEP semantics): an env whose
// previous step ended its episode consumes its next pre-drawn reset record instead of
// stepping; this call reports its reset observation, reward 0 and both flags clear, and the
// step kernel skips it.  No host round trip: the records were drawn from the env's own
// NumPy stream ahead of time (softrod_queue_push*).
template <int EPL>
__global__ void __launch_bounds__(kLanes)
softrod_autoreset_kernel(const RodParams P, const StatePtrs S, float* __restrict__ obs,
                         double* __restrict__ reward, uint8_t* __restrict__ terminated,
                         uint8_t* __restrict__ truncated, double* __restrict__ aux, const int pack) {
    const int rod = blockIdx.x;
    const int lane = threadIdx.x;
    if (!S.needs_reset[rod]) return;
    const size_t N = (size_t)P.n_envs;
    const int k = S.q_consumed[rod];
    if (k >= S.q_produced[rod]) {          // nothing staged: the env stays finished, the host is told
        if (lane == 0) atomicAdd(S.q_underflow, 1);
        return;
    }
    const double* in = S.queue + ((size_t)(k % S.q_depth) * N + rod) * (size_t)S.q_record;
    LaneN<EPL> L;
    reset_rod<EPL>(P, S, N, rod, lane, in, L);
    EnvAction A;
#pragma unroll
    for (int i = 0; i < 7; ++i) A.a[i] = 0.0f;
    A.force = 0.0;
    A.mu_set = false;
    ConstN<EPL> C;
    if (S.mat) build_const<kRuntimeFeatures, EPL, true>(P, lane, A, C, S.mat);
    else build_const<kRuntimeFeatures, EPL>(P, lane, A, C);
    float pa[7];
#pragma unroll
    for (int i = 0; i < 7; ++i) pa[i] = S.prev_action[7 * (size_t)rod + i];
    const int od = env_obs_dim(P);
    float* o = out_row(obs, rod, od, pack);
    env_observe_n<kRuntimeEnv, EPL>(P, S, N, rod, lane, C, L, pa, o);
    if (lane == 0) {
        emit_scalars(o, od, pack, rod, 0.0, false, false, reward, terminated, truncated, S.needs_reset);
        if (aux && P.env_kind == SOFTROD_ENV_SOFTPENDULUM3D) aux[rod] = (double)o[8];
        S.skip[rod] = 1;
        S.q_consumed[rod] = k + 1;
    }
}

}  // namespace softrod

#include "softrod_fast.hpp"
#include "softrod_octo.hpp"
#include "softrod_octo1w.hpp"
#include "softrod_window.hpp"
