// softrod_capi.hip — host side of the C-ABI declared in include/softrod.h.
//
// Owns the resident device state of one batch shard and launches the kernels of
// softrod_kernels.hpp.  No CPU implementation of the physics lives here: the only
// host arithmetic is the rod *allocation* (CosseratRod.straight_rod constants and the
// initial frame of each rod, build.py:46-61), which the reference also performs once
// per reset on the host.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "softrod_kernels.hpp"

using namespace softrod;

struct softrod_handle {
    softrod_config cfg;
    int device = 0;
    int epl = 1;            // nodes/elements per lane: 1 (n_elem <= 63) or 2 (<= 126)
    int nw = 1;             // wavefronts per env: 1, or ceil(n_arm*seg/64) for OctoFlat
    size_t init_stride = 18;  // doubles of reset staging per env
    int window_refresh = 0;   // > 0: the rod runs on two overlapping wave windows (softrod_window.hpp),
                              // halo refreshed every so many substeps
    bool octo_one_env_per_block = false;  // A/B switch SOFTROD_OCTO_ONE_ENV_PER_BLOCK, read once in softrod_create
    bool window_paired = true;            // A/B switch SOFTROD_WINDOW_PAIRED=0: one rod per workgroup with s_barrier
    bool octo_one_wave = false;           // A/B switch SOFTROD_OCTO_ONE_WAVE: softrod_octo1w.hpp (one wave per env, two slots per lane)
    std::string tier;                     // softrod_kernel_tier
    RodParams P{};
    StatePtrs S{};
    double* d_init = nullptr;     // [N][18] staging for reset
    uint8_t* d_mask = nullptr;    // [N]
    double* d_basis = nullptr;    // [(n_elem-1)][7] action basis (zero until set)
    double* d_spline = nullptr;   // breaks[MAX_PIECES + 1], coef[pieces][n_ctrl][4] (softrod_set_spline_table)
    bool spline_set = false;
    RodParams* d_params = nullptr;  // device copy of P
    StatePtrs* d_state = nullptr;   // device copy of S (re-uploaded whenever S changes)
    double* d_time_tab = nullptr;   // clock after k env.steps from a reset (clock_after, softrod_fast.hpp)
    double* d_mat = nullptr;        // [kMatRows][64] material table of a tapered rod
    double* d_sucker = nullptr;     // [SOFTROD_MAX_SUCKERS][N]
    int* d_sucker_idx = nullptr;    // [SOFTROD_MAX_SUCKERS][N]
    double* d_aux = nullptr;        // [8][N] the muscle octopus envs' target and xposbefore
    float* d_prev_kappa = nullptr;  // [N][n_arm * (n_elem - 1)] ArmTwoEnv._prev_kappa
    double* d_mact = nullptr;       // [SOFTROD_MAX_MUSCLES][N][64] muscle activations (SOFTROD_FEAT_COOMM_MUSCLES)
    double* d_mtab = nullptr;       // [SOFTROD_MAX_MUSCLES][4][64] ratio_position x, y, z, strength
    bool muscles_set = false;
    unsigned* d_ticket = nullptr;   // softrod_scatter_rows: blocks that have finished storing (tagged form)
    bool tapered = false;
    bool was_reset = false;
    bool basis_set = false;
    double* h_init = nullptr;     // pinned
    uint8_t* h_mask = nullptr;    // pinned
    std::vector<hipEvent_t> ev_start, ev_stop;  // timing ring (softrod_set_timing)
    int timed = 0;                  // launches recorded since set_timing
    hipEvent_t ev_reset = nullptr;  // guards reuse of the pinned staging buffers
    // device-side auto-reset (softrod_autoreset_enable)
    int q_depth = 0;
    double* d_queue = nullptr;      // [depth][N][init_stride]
    double* h_queue = nullptr;      // pinned mirror
    int* d_consumed = nullptr;      // [N]
    int* d_produced = nullptr;      // [N]
    int* d_underflow = nullptr;     // [1]
    uint8_t* d_flags = nullptr;     // [2][N]: needs_reset, skip
    int* h_produced = nullptr;      // pinned [N]
    std::vector<int> seen_consumed; // as of the last softrod_queue_status
    std::vector<int2> pending;      // (env, slot) written into h_queue since the last commit
    double* h_stage = nullptr;      // pinned: the pending records, compacted
    int2* h_where = nullptr;        // pinned
    double* d_stage = nullptr;
    int2* d_where = nullptr;
    size_t stage_cap = 0;           // records the staging buffers hold
    int* h_status = nullptr;        // pinned [N + 1]: consumed, underflow (softrod_queue_status_begin)
    hipEvent_t ev_status = nullptr;
    bool status_pending = false;
    hipEvent_t ev_queue = nullptr;  // guards reuse of h_queue / h_produced
    std::string err;
};

namespace {

thread_local std::string g_err;  // errors raised before a handle exists

int fail(softrod_handle* h, int code, const std::string& msg) {
    if (h) h->err = msg; else g_err = msg;
    return code;
}

// Every entry point runs on the handle's device and leaves the caller's current device as it
// found it (a process may hold handles on several GPUs, or have torch's current device elsewhere).
struct DeviceGuard {
    int prev = -1, want = -1;
    hipError_t err = hipSuccess;
    explicit DeviceGuard(int device) : want(device) {
        err = hipGetDevice(&prev);
        if (err == hipSuccess && prev != want) err = hipSetDevice(want);
    }
    ~DeviceGuard() {
        if (prev >= 0 && prev != want) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define SR_ON_DEVICE(h)                                                                  \
    DeviceGuard guard_((h)->device);                                                     \
    if (guard_.err != hipSuccess)                                                        \
        return fail(h, SOFTROD_EHIP, std::string("hipSetDevice: ") + hipGetErrorString(guard_.err))

#define SR_HIP(h, call)                                                                 \
    do {                                                                                \
        hipError_t e_ = (call);                                                         \
        if (e_ != hipSuccess)                                                           \
            return fail(h, SOFTROD_EHIP, std::string(#call) + ": " + hipGetErrorString(e_)); \
    } while (0)

// Constants of a uniform straight rod — CosseratRod.straight_rod
// (elastica/rod/factory_function.py allocate(), called at build.py:54-61) and
// AnalyticalLinearDamper.__init__ (build.py:108-113).
void fill_params(const softrod_config& c, RodParams& P) {
    std::memset(&P, 0, sizeof(P));
    P.n_envs = c.n_envs;
    P.n_elem = c.n_elem;
    P.time_two_half_adds = c.time_two_half_adds;
    P.features = c.features;
    P.env_kind = c.env_kind;
    P.filter_order = c.filter_order;
    P.damp_before_constrain = c.damp_before_constrain;
    P.dt = c.dt;
    P.half_dt = 0.5 * c.dt;
    P.final_time = c.final_time;
    const int n = c.n_elem;
    const double rest_len = c.base_length / (double)n;
    P.rest_len = rest_len;
    P.inv_rest_len = 1.0 / rest_len;
    P.rest_vor = 0.5 * (rest_len + rest_len);
    P.inv_rest_vor = 1.0 / P.rest_vor;
    const double r = c.base_radius;
    const double A0 = M_PI * r * r;
    const double I1 = A0 * A0 / (4.0 * M_PI);
    const double I[3] = {I1, I1, 2.0 * I1};
    for (int i = 0; i < 3; ++i) {
        P.J[i] = I[i] * (c.density * rest_len);
        P.invJ[i] = 1.0 / P.J[i];
    }
    P.shear[0] = c.alpha_c * c.shear_modulus * A0;
    P.shear[1] = c.alpha_c * c.shear_modulus * A0;
    P.shear[2] = c.youngs_modulus * A0;
    P.bend[0] = c.youngs_modulus * I[0];
    P.bend[1] = c.youngs_modulus * I[1];
    P.bend[2] = c.shear_modulus * I[2];
    const double volume = M_PI * (r * r) * rest_len;
    P.mass_node = c.density * volume;  // two half-element contributions
    for (int i = 0; i < 3; ++i) { P.gravity[i] = c.gravity[i]; P.tip_force[i] = c.tip_force[i]; }
    // AnalyticalLinearDamper(time_step=...): the stepper's dt unless the build hands it another one
    // (build_muscle_octopus.py:101-106: 7e-5 under a stepper run at 5e-5)
    const double ddt = c.damper_time_step > 0.0 ? c.damper_time_step : c.dt;
    P.damp_t = std::exp(-c.damping_constant * ddt);
    // element mass seen by the damper: 0.5(m_k+m_{k+1}), ends augmented by half their
    // outer node -> rho*V for every element of a uniform rod
    const double me = P.mass_node;
    for (int i = 0; i < 3; ++i) {      // (damper_protocol 1: the uniform protocol, the same exp(-nu dt) on every rate)
        P.damp_logr[i] = c.damper_protocol == 1 ? -c.damping_constant * ddt : -c.damping_constant * ddt * me * P.invJ[i];
        P.damp_r[i] = std::exp(P.damp_logr[i]);
    }
    P.eps_length = c.eps_length;
    P.eps_rot_axis = c.eps_rot_axis;
    P.acos_shift = c.acos_shift;
    P.eps_sin = c.eps_sin;
    P.two_acos_shift = 2.0 * c.acos_shift + 1.0e-200;
    P.neg_eps_sin = -c.eps_sin;
    P.base_limit = c.base_limit;
    P.step_time = (double)c.n_substeps * c.dt;  // step_skip * time_step, soft_pendulum_3d.py:110-112
    P.base_step = (float)c.base_step;
    // OctoArmSingle-v0
    P.control_penalty_coeff = (float)c.control_penalty_coeff;
    P.contact_before_forcing = c.contact_before_forcing;
    P.n_action = 7;
    {   // mass.sum() in the order compute_position_center_of_mass adds it
        double ms = 0.0;
        for (int k = 0; k <= n; ++k) ms += (k == 0 || k == n) ? 0.5 * P.mass_node : P.mass_node;
        P.mass_total = ms;
    }
    for (int i = 0; i < 2; ++i) {
        P.target[i] = c.target[i];
        P.kappa_range[i] = c.kappa_range[i];
        P.kappa_rate_range[i] = c.kappa_rate_range[i];
    }
    for (int i = 0; i < 3; ++i) {
        P.plane_origin[i] = c.plane_origin[i];
        P.plane_normal[i] = c.plane_normal[i];
        P.kin_mu[i] = c.kinetic_mu[i];
        P.stat_mu[i] = c.static_mu[i];
    }
    P.contact_k = c.contact_k;
    P.contact_nu = c.contact_nu;
    P.slip_tol = c.slip_velocity_tol;
    P.surface_tol = c.surface_tol;
    P.r0_sqrt_rest_len = r * std::sqrt(rest_len);   // radius = sqrt(V/(pi l)) = r0 sqrt(l_rest/l)
    if ((c.features & SOFTROD_FEAT_PLANE_CONTACT_ANISO) && c.plane_normal[0] == 0.0 &&
        c.plane_normal[1] == 0.0 && c.plane_normal[2] == 1.0)
        P.features |= kFeatPlaneZup;   // what both reference builds use (octopus/build.py:233)
    if (c.features & SOFTROD_FEAT_OCTO_HEAD) {
        // arms `seg` slots apart; Cylinder(length = 2 r0, radius = head_radius, density)
        // (octopus/build.py:95-105; elastica/rigidbody/cylinder.py)
        P.seg_shift = n <= 15 ? 4 : (n <= 31 ? 5 : 6);
        P.seg = 1 << P.seg_shift;
        P.n_arm = c.n_arm;
        P.n_action = c.n_knots;
        if (c.env_kind == SOFTROD_ENV_ARM_PULL_WEIGHT) P.seg_shift = 6, P.seg = 64;    // the muscle arm with a weight: one arm, one wave
        P.head_fixed = c.head_fixed;
        const double hl = c.head_length > 0.0 ? c.head_length : 2.0 * c.base_radius, hr = c.head_radius;
        for (int i = 0; i < 3; ++i) P.head_center[i] = c.head_length > 0.0 ? c.head_center[i] : 0.0;
        P.joint_angle0 = c.head_length > 0.0 ? c.joint_angle0 : 0.0;
        P.joint_angle_step = c.head_length > 0.0 ? c.joint_angle_step : 360 / (double)c.n_arm;
        const double harea = M_PI * hr * hr;
        P.head_mass = (M_PI * hr * hr * hl) * c.head_density;
        const double s1 = harea * harea / (4.0 * M_PI);
        const double smoa[3] = {s1, s1, 2.0 * s1};
        for (int i = 0; i < 3; ++i) {
            P.head_invJ[i] = 1.0 / (smoa[i] * c.head_density * hl);
        }
        P.head_radius = hr;
        P.joint_k = c.joint_k;
        P.joint_nu = c.joint_nu;
        P.joint_kt = c.joint_kt;
    }
    // SoftArmTracking-v0
    P.n_ctrl = c.n_ctrl;
    P.n_pieces = c.n_spline_pieces;
    P.muscle_scale = c.muscle_torque_scale;
    P.max_rate = c.max_activation_rate;
    P.base_length = c.base_length;
    for (int i = 0; i < 3; ++i) P.arm_target[i] = c.arm_target[i];
    P.n_suckers = c.n_suckers;
    for (int j = 0; j < SOFTROD_MAX_SUCKERS; ++j) P.sucker_index[j] = c.sucker_index[j];
    P.sucker_ratio0 = c.sucker_reduction_ratio;
    // COOMM muscle layers
    P.n_muscles = (c.features & SOFTROD_FEAT_COOMM_MUSCLES) ? c.n_muscles : 0;
    for (int m = 0; m < SOFTROD_MAX_MUSCLES; ++m) P.muscle_kind[m] = c.muscle_kind[m];
    P.fl_degree = c.muscle_fl_degree;
    for (int k = 0; k < SOFTROD_MAX_FL_COEF; ++k) P.fl_coef[k] = c.muscle_fl_coef[k];
    P.muscle_form = c.muscle_equiv_load_form;
    P.muscle_cur_radius = c.muscle_position_current_radius;
    P.muscle_tm_law = c.muscle_tm_length_law;
    P.push_mode = c.arm_push_mode;
}

bool is_octo(const softrod_handle* h) { return (h->cfg.features & SOFTROD_FEAT_OCTO_HEAD) != 0; }
// the rigid-body set whose host API is FlatEnv's (softrod_reset_octo, softrod_queue_push_octo); the muscle arm with a
// weight (SOFTROD_ENV_ARM_PULL_WEIGHT) runs the same kernels but resets like any single rod (softrod_reset_straight)
bool is_flat(const softrod_handle* h) { return h->cfg.env_kind == SOFTROD_ENV_OCTO_FLAT; }
bool is_pull(const softrod_handle* h) { return h->cfg.env_kind == SOFTROD_ENV_ARM_PULL_WEIGHT; }
// the muscle octopus envs (softrod_mocto.hpp): FlatEnv's host API (arm frames + target), the muscle arm's tables
bool mocto_kind(int e) { return e == SOFTROD_ENV_CRAWL || e == SOFTROD_ENV_ARM_TWO || e == SOFTROD_ENV_REACH; }
bool is_mocto(const softrod_handle* h) { return mocto_kind(h->cfg.env_kind); }

int launch_step(softrod_handle* h, const float* actions, float* obs, double* reward,
                uint8_t* term, uint8_t* trunc, double* aux, int n_sub, int epilogue, int pack,
                hipStream_t st) {
    const dim3 grid((unsigned)h->cfg.n_envs), block(kLanes * h->nw);
    if (h->q_depth > 0 && epilogue) {   // NEXT_STEP auto-reset pass: finished envs restart instead of stepping
        if (is_octo(h))
            hipLaunchKernelGGL(softrod_octo_autoreset_kernel, grid, block, 0, st, h->P, h->S, obs, reward, term,
                               trunc, pack);
        else if (h->epl == 2)
            hipLaunchKernelGGL(softrod_autoreset_kernel<2>, grid, block, 0, st, h->P, h->S, obs, reward, term,
                               trunc, aux, pack);
        else
            hipLaunchKernelGGL(softrod_autoreset_kernel<1>, grid, block, 0, st, h->P, h->S, obs, reward, term,
                               trunc, aux, pack);
        SR_HIP(h, hipGetLastError());
    }
    if ((h->cfg.features & SOFTROD_FEAT_COOMM_MUSCLES) && h->cfg.math_mode == SOFTROD_MATH_FAST) {
        // the fast kernel carries the muscle layers in two instantiations only (kMusclesCompiled): the tapered
        // ArmPush arm and the uniform muscle rod; never let another one run a muscle handle without its muscles
        const bool push = h->cfg.env_kind == SOFTROD_ENV_ARM_PUSH || is_pull(h) || is_mocto(h);
        if (is_mocto(h) && (!h->tapered || !h->muscles_set || (h->cfg.env_kind == SOFTROD_ENV_ARM_TWO && !h->basis_set)))
            return fail(h, SOFTROD_EINVAL, "the muscle octopus needs softrod_set_radius_profile and softrod_set_muscle_layers "
                                           "(and SOFTROD_ENV_ARM_TWO softrod_set_action_basis) before it steps");
        if (push && !h->tapered)
            return fail(h, SOFTROD_EINVAL, "SOFTROD_ENV_ARM_PUSH (SOFTROD_MATH_FAST): call softrod_set_radius_profile first "
                                           "(the reference's arm is tapered, arm_push_env.py:160-179)");
        if (!push && h->tapered)
            return fail(h, SOFTROD_EINVAL, "a tapered muscle rod outside SOFTROD_ENV_ARM_PUSH runs under SOFTROD_MATH_LIBM only");
    }
    const bool timing = h->timed < (int)h->ev_start.size();
    if (timing) SR_HIP(h, hipEventRecord(h->ev_start[h->timed], st));
    const bool zup = (h->P.features & kFeatPlaneZup) != 0;
    if (is_octo(h) && is_pull(h)) {
        hipLaunchKernelGGL((softrod_octo_step_kernel<SOFTROD_FEATURES_ARM_PULL_WEIGHT, 2, 1>), grid, block, 0, st, h->P, h->S,
                           actions, obs, reward, term, trunc, n_sub, epilogue, pack);
    } else if (is_mocto(h)) {
        // set_action | the body's substeps | get_state + reward (softrod_mocto.hpp); the timing events bracket all three
        if (epilogue && actions)
            hipLaunchKernelGGL(softrod_mocto_action_kernel, grid, block, 0, st, h->P, h->S, actions, n_sub);
        hipLaunchKernelGGL((softrod_octo_step_kernel<SOFTROD_FEATURES_ARM_PULL_WEIGHT, 4, 1>), grid, block, 0, st, h->P, h->S,
                           actions, obs, reward, term, trunc, n_sub, epilogue, pack);
        if (epilogue)
            hipLaunchKernelGGL(softrod_mocto_epilogue_kernel, grid, block, 0, st, h->P, h->S, obs, reward, term, trunc, 1, pack);
    } else if (is_octo(h)) {
#define SR_OCTO(FEATS, MAXW)                                                                        \
        hipLaunchKernelGGL((softrod_octo_step_kernel<FEATS, MAXW>), grid, block, 0, st, h->P, h->S,     \
                           actions, obs, reward, term, trunc, n_sub, epilogue, pack)
        // the reference shape (two waves per env): four envs per workgroup, partner waves on one SIMD
        if (zup && h->nw == 2 && h->octo_one_wave && h->P.n_arm * h->P.seg == 2 * kLanes && !(h->P.seg & 1)) {
            hipLaunchKernelGGL((softrod_octo1w_step_kernel<SOFTROD_FEATURES_OCTO_FLAT | kFeatPlaneZup>),
                               dim3((unsigned)h->cfg.n_envs), dim3(kLanes), 0, st, h->P, h->S,
                               actions, obs, reward, term, trunc, n_sub, epilogue, pack);
        } else
        if (zup && h->nw == 2 && !h->octo_one_env_per_block) {
            hipLaunchKernelGGL((softrod_octo_step_kernel<SOFTROD_FEATURES_OCTO_FLAT | kFeatPlaneZup, 2, 4>),
                               dim3((unsigned)((h->cfg.n_envs + 3) / 4)), dim3(kLanes * 8), 0, st, h->P, h->S,
                               actions, obs, reward, term, trunc, n_sub, epilogue, pack);
        } else
        if (zup) { if (h->nw <= 2) SR_OCTO(SOFTROD_FEATURES_OCTO_FLAT | kFeatPlaneZup, 2);
                   else SR_OCTO(SOFTROD_FEATURES_OCTO_FLAT | kFeatPlaneZup, 8); }
        else     { if (h->nw <= 2) SR_OCTO(SOFTROD_FEATURES_OCTO_FLAT, 2);
                   else SR_OCTO(SOFTROD_FEATURES_OCTO_FLAT, 8); }
#undef SR_OCTO
    } else if (h->window_refresh > 0 && epilogue) {
        // substeps on two overlapping one-node-per-lane windows, then reward / observation by the
        // two-slot kernel with n_sub = 0 on the same rows (the timing events bracket both)
        if (h->window_paired)     // four rods per workgroup, a rod's two windows on one SIMD (softrod_window.hpp)
            hipLaunchKernelGGL((softrod_step_window_kernel<SOFTROD_FEATURES_ARM_SINGLE | kFeatPlaneZup, 4>),
                               dim3((unsigned)((h->cfg.n_envs + 3) / 4)), dim3(8 * kLanes), 0, st, h->P, h->S, actions,
                               n_sub, h->window_refresh);
        else
        hipLaunchKernelGGL((softrod_step_window_kernel<SOFTROD_FEATURES_ARM_SINGLE | kFeatPlaneZup>), grid,
                           dim3(2 * kLanes), 0, st, h->P, h->S, actions, n_sub, h->window_refresh);
        hipLaunchKernelGGL((softrod_step_fast_kernel<SOFTROD_FEATURES_ARM_SINGLE | kFeatPlaneZup,
                                                     SOFTROD_ENV_ARM_SINGLE, 2>),
                           grid, dim3(kLanes), 0, st, h->P, h->S, actions, obs, reward, term, trunc, aux, 0, 1, pack);
    } else if (h->cfg.math_mode == SOFTROD_MATH_FAST) {
        // instantiations specialised for the registered envs' feature sets (one or two
        // slots per lane); anything else (known-answer tests, custom feature mixes) takes
        // the run-time-mask instantiation
        const unsigned f = h->cfg.features;
        const int e = h->cfg.env_kind;
#define SR_LAUNCH(FEATS, ENV, EPL)                                                                  \
        hipLaunchKernelGGL((softrod_step_fast_kernel<FEATS, ENV, EPL>), grid, block, 0, st, h->P, h->S, \
                           actions, obs, reward, term, trunc, aux, n_sub, epilogue, pack)
#define SR_DISPATCH(EPL)                                                                            \
        do {                                                                                        \
            if (f == SOFTROD_FEATURES_SOFTPENDULUM && e == SOFTROD_ENV_SOFTPENDULUM)                \
                SR_LAUNCH(SOFTROD_FEATURES_SOFTPENDULUM, SOFTROD_ENV_SOFTPENDULUM, EPL);            \
            else if (f == SOFTROD_FEATURES_SOFTPENDULUM3D && e == SOFTROD_ENV_SOFTPENDULUM3D)       \
                SR_LAUNCH(SOFTROD_FEATURES_SOFTPENDULUM3D, SOFTROD_ENV_SOFTPENDULUM3D, EPL);        \
            else if (f == SOFTROD_FEATURES_ARM_SINGLE && e == SOFTROD_ENV_ARM_SINGLE && zup)        \
                SR_LAUNCH(SOFTROD_FEATURES_ARM_SINGLE | kFeatPlaneZup, SOFTROD_ENV_ARM_SINGLE, EPL); \
            else if (f == SOFTROD_FEATURES_SOFT_ARM && e == SOFTROD_ENV_SOFT_ARM)                   \
                SR_LAUNCH(SOFTROD_FEATURES_SOFT_ARM, SOFTROD_ENV_SOFT_ARM, EPL);                    \
            else if (f == kFeaturesMuscleRod && e == SOFTROD_ENV_NONE && EPL == 1)                  \
                SR_LAUNCH(kFeaturesMuscleRod, SOFTROD_ENV_NONE, 1);                                 \
            else                                                                                    \
                SR_LAUNCH(kRuntimeFeatures, kRuntimeEnv, EPL);                                      \
        } while (0)
        if (h->tapered) {    // per-lane material constants (TAPER = true), one slot per lane
#define SR_LAUNCH_TAPER(FEATS, ENV)                                                                 \
            hipLaunchKernelGGL((softrod_step_fast_kernel<FEATS, ENV, 1, true>), grid, block, 0, st, h->P, h->S, \
                               actions, obs, reward, term, trunc, aux, n_sub, epilogue, pack)
            // the two tapered feature sets the reference holds on disk get their own instantiation: the
            // OctoArmSingle set (a tapered arm on the plane, `bench.py --taper`) and the damped arm with
            // ControllableFixConstraint suckers of arm_push_env.py:160-196 (its COOMM muscles are not on
            // disk); any other mix takes the run-time mask
            if (f == SOFTROD_FEATURES_ARM_SINGLE && e == SOFTROD_ENV_ARM_SINGLE && zup)
                SR_LAUNCH_TAPER(SOFTROD_FEATURES_ARM_SINGLE | kFeatPlaneZup, SOFTROD_ENV_ARM_SINGLE);
            else if (f == kFeaturesTaperedSuckerArm && e == SOFTROD_ENV_NONE)
                SR_LAUNCH_TAPER(kFeaturesTaperedSuckerArm, SOFTROD_ENV_NONE);
            else if (f == SOFTROD_FEATURES_ARM_PUSH && e == SOFTROD_ENV_ARM_PUSH)     // OctoArmPush-v0 / -v1
                SR_LAUNCH_TAPER(SOFTROD_FEATURES_ARM_PUSH, SOFTROD_ENV_ARM_PUSH);
            else
                SR_LAUNCH_TAPER(kRuntimeFeatures, kRuntimeEnv);
#undef SR_LAUNCH_TAPER
        } else if (h->epl == 2) SR_DISPATCH(2); else SR_DISPATCH(1);
#undef SR_DISPATCH
#undef SR_LAUNCH
    } else
        hipLaunchKernelGGL(softrod_step_libm_kernel, grid, block, 0, st, h->P, h->S,
                           actions, obs, reward, term, trunc, aux, n_sub, epilogue, pack);
    SR_HIP(h, hipGetLastError());
    if (timing) {
        SR_HIP(h, hipEventRecord(h->ev_stop[h->timed], st));
        ++h->timed;
    }
    return SOFTROD_OK;
}

int upload_and_reset(softrod_handle* h, hipStream_t st, bool use_mask) {
    const size_t N = (size_t)h->cfg.n_envs;
    SR_HIP(h, hipMemcpyAsync(h->d_init, h->h_init, N * h->init_stride * sizeof(double), hipMemcpyHostToDevice, st));
    if (use_mask)
        SR_HIP(h, hipMemcpyAsync(h->d_mask, h->h_mask, N, hipMemcpyHostToDevice, st));
    ResetArgs A{h->d_init, use_mask ? h->d_mask : nullptr};
    if (is_octo(h)) {
        OctoResetArgs OA{h->d_init, h->d_init + N * (size_t)h->cfg.n_arm * 18, use_mask ? h->d_mask : nullptr};   // (targets: 2 numbers per env, 4 for the muscle octopus)
        hipLaunchKernelGGL(softrod_octo_reset_kernel, dim3((unsigned)N), dim3(kLanes * h->nw), 0, st, h->P, h->S, OA);
    } else if (h->epl == 2)
        hipLaunchKernelGGL(softrod_reset_kernel<2>, dim3((unsigned)N), dim3(kLanes), 0, st, h->P, h->S, A);
    else
        hipLaunchKernelGGL(softrod_reset_kernel<1>, dim3((unsigned)N), dim3(kLanes), 0, st, h->P, h->S, A);
    SR_HIP(h, hipGetLastError());
    SR_HIP(h, hipEventRecord(h->ev_reset, st));
    h->was_reset = true;
    return SOFTROD_OK;
}

// host part of straight_rod for one rod: start, step, end, Q rows
void straight_init(const softrod_config& c, const double start[3], const double direction[3],
                   const double normal_in[3], double out[18]) {
    const int n = c.n_elem;
    double end[3], normal[3], t[3], d[3];
    for (int i = 0; i < 3; ++i) end[i] = start[i] + direction[i] * c.base_length;
    const double nn = std::sqrt(normal_in[0] * normal_in[0] + normal_in[1] * normal_in[1] +
                                normal_in[2] * normal_in[2]);
    for (int i = 0; i < 3; ++i) normal[i] = normal_in[i] / nn;
    for (int i = 0; i < 3; ++i) {
        out[i] = start[i];
        out[3 + i] = (end[i] - start[i]) / (double)n;
        out[6 + i] = end[i];
        d[i] = (start[i] + 1.0 * out[3 + i]) - start[i];  // x[1]-x[0] as linspace produces it
    }
    const double l = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
    for (int i = 0; i < 3; ++i) t[i] = d[i] / l;
    for (int i = 0; i < 3; ++i) out[9 + i] = normal[i];
    out[12] = t[1] * normal[2] - t[2] * normal[1];
    out[13] = t[2] * normal[0] - t[0] * normal[2];
    out[14] = t[0] * normal[1] - t[1] * normal[0];
    for (int i = 0; i < 3; ++i) out[15 + i] = t[i];
}

void config_common(softrod_config* cfg, int n_envs) {
    std::memset(cfg, 0, sizeof(*cfg));
    cfg->struct_size = (uint32_t)sizeof(softrod_config);
    cfg->n_envs = n_envs;
    cfg->n_elem = 50;                                   // soft_pendulum.py:64
    cfg->dt = 1.0e-4;                                   // :62
    cfg->n_substeps = (int)(1.0 / (25 * cfg->dt));      // :78 (recording_fps = 25, :63)
    cfg->math_mode = SOFTROD_MATH_FAST;
    cfg->final_time = 5.0;                              // :61
    cfg->alpha_c = 27.0 / 28.0;
    cfg->eps_length = 1e-14;
    cfg->eps_rot_axis = 1e-14;
    cfg->acos_shift = 1e-10;
    cfg->eps_sin = 1e-14;
    cfg->time_two_half_adds = 1;
    cfg->damp_before_constrain = 0;   // constrain() is registered before dampen() in both builds
}

}  // namespace

extern "C" {

int softrod_abi_version(void) { return SOFTROD_ABI_VERSION; }

#ifndef SOFTROD_SOURCE_HASH
#define SOFTROD_SOURCE_HASH "unhashed"
#endif
const char* softrod_source_hash(void) { return SOFTROD_SOURCE_HASH; }

int softrod_action_dim(int env_kind) {
    return env_kind == SOFTROD_ENV_SOFTPENDULUM3D ? 2 : env_kind == SOFTROD_ENV_ARM_SINGLE ? 7
         : env_kind == SOFTROD_ENV_OCTO_FLAT ? 24 : env_kind == SOFTROD_ENV_SOFT_ARM ? 8
         : (env_kind == SOFTROD_ENV_ARM_PUSH || env_kind == SOFTROD_ENV_ARM_PULL_WEIGHT) ? 2
         : env_kind == SOFTROD_ENV_CRAWL ? 24 : env_kind == SOFTROD_ENV_ARM_TWO ? 18 : env_kind == SOFTROD_ENV_REACH ? 480 : 1;
}
int softrod_obs_dim(int env_kind) {
    return env_kind == SOFTROD_ENV_SOFTPENDULUM3D ? 9 : env_kind == SOFTROD_ENV_ARM_SINGLE ? 25
         : env_kind == SOFTROD_ENV_OCTO_FLAT ? 8 * 56 + 13 : env_kind == SOFTROD_ENV_SOFT_ARM ? 14
         : (env_kind == SOFTROD_ENV_ARM_PUSH || env_kind == SOFTROD_ENV_ARM_PULL_WEIGHT) ? 84
         : env_kind == SOFTROD_ENV_CRAWL ? 8 * 131 : env_kind == SOFTROD_ENV_ARM_TWO ? 2 * 52 : env_kind == SOFTROD_ENV_REACH ? 8 * 189 : 4;
}
int softrod_config_action_dim(const softrod_config* cfg) {
    if (!cfg) return 0;
    if (cfg->env_kind == SOFTROD_ENV_SOFT_ARM) return 2 * cfg->n_ctrl;     // soft_arm_tracking.py:152-157
    if (cfg->env_kind == SOFTROD_ENV_ARM_PUSH || cfg->env_kind == SOFTROD_ENV_ARM_PULL_WEIGHT)
        return cfg->arm_push_mode == 0 ? 1 : 2;                             // arm_push_env.py:100-118
    if (mocto_kind(cfg->env_kind)) return cfg->n_arm * cfg->n_knots;        // crawl_env.py:84-88, arm_two_env.py:84-89, reach_env.py:79-84
    return cfg->env_kind == SOFTROD_ENV_OCTO_FLAT ? cfg->n_arm * cfg->n_knots : softrod_action_dim(cfg->env_kind);
}
int softrod_config_obs_dim(const softrod_config* cfg) {
    if (!cfg) return 0;
    if (cfg->env_kind == SOFTROD_ENV_SOFT_ARM) return 2 * cfg->n_ctrl + 6;  // :158-163
    if (cfg->env_kind == SOFTROD_ENV_ARM_PUSH || cfg->env_kind == SOFTROD_ENV_ARM_PULL_WEIGHT)
        return 2 * (cfg->n_elem + 1) + 2;                                   // arm_push_env.py:104,118-120
    if (mocto_kind(cfg->env_kind)) {
        const int n = cfg->n_elem, na = cfg->n_arm, nk = cfg->n_knots;
        if (cfg->env_kind == SOFTROD_ENV_ARM_TWO) return na * ((n - 1) * 2 + nk + na + 3);          // arm_two_env.py:92-95
        return na * ((n - 1) + (n + 1) * 4 + nk + na + (cfg->env_kind == SOFTROD_ENV_CRAWL ? 17 : 18));   // crawl_env.py:91-101, reach_env.py:87-91
    }
    if (cfg->env_kind != SOFTROD_ENV_OCTO_FLAT) return softrod_obs_dim(cfg->env_kind);
    return cfg->n_arm * ((cfg->n_elem - 1) + 4 * (cfg->n_elem + 1) + cfg->n_knots) + 13;
}
int softrod_aux_dim(int env_kind) { return env_kind == SOFTROD_ENV_SOFTPENDULUM3D ? 1 : 0; }

int softrod_config_softpendulum(softrod_config* cfg, int n_envs) {
    if (!cfg || n_envs < 1) return SOFTROD_EINVAL;
    config_common(cfg, n_envs);
    cfg->features = SOFTROD_FEATURES_SOFTPENDULUM;
    cfg->env_kind = SOFTROD_ENV_SOFTPENDULUM;
    cfg->base_length = 1.0;                             // build.py:23-26
    cfg->base_radius = 0.05;
    cfg->density = 1000.0;                              // build.py:18-22
    cfg->youngs_modulus = 1e6;
    cfg->shear_modulus = 1e6 / (2.0 * (1.0 + 0.5));     // PyElastica default (none passed)
    cfg->gravity[0] = 0.0; cfg->gravity[1] = -9.80665; cfg->gravity[2] = 0.0;  // build.py:87-91
    cfg->damping_constant = 2e-3;                       // build.py:108
    return SOFTROD_OK;
}

int softrod_config_softpendulum3d(softrod_config* cfg, int n_envs) {
    if (!cfg || n_envs < 1) return SOFTROD_EINVAL;
    config_common(cfg, n_envs);
    cfg->features = SOFTROD_FEATURES_SOFTPENDULUM3D;
    cfg->env_kind = SOFTROD_ENV_SOFTPENDULUM3D;
    cfg->filter_order = 7;                              // soft_pendulum_3d/build.py:82-85
    cfg->base_length = 1.0;                             // soft_pendulum_3d/build.py:55-64
    cfg->base_radius = 0.1;
    cfg->density = 4000.0;
    cfg->youngs_modulus = 1e6;
    cfg->shear_modulus = 1e6 / (2.0 * (1.0 + 0.5));
    cfg->gravity[0] = 0.0; cfg->gravity[1] = 0.0; cfg->gravity[2] = -9.80665;  // :73-76
    cfg->damping_constant = 1.0;                        // :77-81
    cfg->base_step = 1e-3;                              // soft_pendulum_3d.py:57
    cfg->base_limit = 0.5;                              // :58
    return SOFTROD_OK;
}

int softrod_config_arm_single(softrod_config* cfg, int n_envs) {
    if (!cfg || n_envs < 1) return SOFTROD_EINVAL;
    config_common(cfg, n_envs);
    cfg->features = SOFTROD_FEATURES_ARM_SINGLE;
    cfg->env_kind = SOFTROD_ENV_ARM_SINGLE;
    cfg->dt = 7.0e-5;                                   // octopus/arm_single_env.py:58
    cfg->n_substeps = (int)(1.0 / (20 * cfg->dt));      // recording_fps = 20 (:59) -> 714 (:77)
    cfg->final_time = 10.0;                             // :57
    const double L0 = 0.35, r0 = 0.35 * 0.02;           // octopus/build.py:46-49
    cfg->base_length = L0;
    cfg->base_radius = r0;
    cfg->density = 1000.0;                              // :30-33
    cfg->youngs_modulus = 1e6;
    cfg->shear_modulus = 1e6 / (2.0 * (1.0 + 0.5));
    const double g = -9.81;                             // :237
    cfg->gravity[0] = 0.0; cfg->gravity[1] = 0.0; cfg->gravity[2] = g;
    cfg->damping_constant = 1e-2;                       // :285
    cfg->plane_origin[0] = 0.0; cfg->plane_origin[1] = 0.0; cfg->plane_origin[2] = -r0;  // :244
    cfg->plane_normal[0] = 0.0; cfg->plane_normal[1] = 0.0; cfg->plane_normal[2] = 1.0;  // :233
    cfg->contact_k = 1e2;                               // :241
    cfg->contact_nu = 1e1;                              // :242
    cfg->slip_velocity_tol = 1e-8;                      // :245
    cfg->surface_tol = 1e-4;
    const double period = 2.0, froude = 0.1;
    const double mu = L0 / (period * period * std::fabs(g) * froude);   // :247
    const double fac[3] = {1.0, 1.5, 2.0};               // forward, backward, sideways (:253-256)
    for (int i = 0; i < 3; ++i) {
        cfg->kinetic_mu[i] = mu * fac[i];
        cfg->static_mu[i] = 2 * (mu * fac[i]);          // :257
    }
    cfg->control_penalty_coeff = 0.001;                 // arm_single_env.py:62
    cfg->target[0] = 1.0; cfg->target[1] = 0.0;         // :165
    cfg->kappa_range[0] = -49.33508476187419; cfg->kappa_range[1] = 49.33545827754751;          // :111
    cfg->kappa_rate_range[0] = -21.063520620377012; cfg->kappa_rate_range[1] = 24.664591289161944;  // :113
    return SOFTROD_OK;
}

int softrod_config_octo_flat(softrod_config* cfg, int n_envs) {
    const int rc = softrod_config_arm_single(cfg, n_envs);   // per-arm constants: build.py:30-49,134-200
    if (rc != SOFTROD_OK) return rc;
    cfg->features = SOFTROD_FEATURES_OCTO_FLAT;
    cfg->env_kind = SOFTROD_ENV_OCTO_FLAT;
    cfg->n_elem = 10;                                    // octopus/flat_env.py:62
    cfg->final_time = 5.0;                               // :58
    cfg->n_substeps = (int)(1.0 / (5 * cfg->dt));        // recording_fps = 5 (:60) -> 2857
    cfg->n_arm = 8;                                      // :61
    cfg->n_knots = 3;                                    // n_action (:63)
    cfg->head_radius = 0.04;                             // octopus/build.py:95-105
    cfg->head_density = 700.0;
    cfg->joint_k = 1e6;                                  // :117-132
    cfg->joint_nu = 1e-3;
    cfg->joint_kt = 1e0;
    cfg->head_center[0] = 0.0; cfg->head_center[1] = 0.0;          // Cylinder(start = (0, 0, -r0), e_z, e_y, 2 r0), :95-105
    cfg->head_center[2] = -cfg->base_radius + 2.0 * cfg->base_radius / 2;
    cfg->head_length = 2.0 * cfg->base_radius;
    cfg->joint_angle0 = 0.0;
    cfg->joint_angle_step = 360 / (double)cfg->n_arm;    // :73-74
    return SOFTROD_OK;
}

int softrod_config_soft_arm(softrod_config* cfg, int n_envs) {
    if (!cfg || n_envs < 1) return fail(nullptr, SOFTROD_EINVAL, "bad argument");
    config_common(cfg, n_envs);
    cfg->features = SOFTROD_FEATURES_SOFT_ARM;
    cfg->env_kind = SOFTROD_ENV_SOFT_ARM;
    cfg->n_elem = 40;                                    // soft_arm/soft_arm_tracking.py:116
    cfg->dt = 2.0e-4;                                    // sim_dt, :117
    cfg->n_substeps = (int)std::rint(0.01 / cfg->dt);    // num_steps_per_update, :118-121
    cfg->final_time = 5.0;                               // max_episode_final_time, :127
    cfg->base_length = 1000.0;                           // :129 (millimetres)
    cfg->base_radius = 50.0;                             // :130
    cfg->density = 1000 * 1e-6;                          // :280
    cfg->youngs_modulus = 2e6;                           // :122
    cfg->shear_modulus = 2e6 / (2.0 * (1.0 + 0.5));      // not passed at :272-283: PyElastica's default
    cfg->damping_constant = 2e6 * 1e-7 * 1;              // :269
    cfg->n_ctrl = 4;                                     // :132
    cfg->n_spline_pieces = 3;                            // 4 + 2 points, not-a-knot
    cfg->muscle_torque_scale = 10 * 50.0 * 2e6;          // alpha, :350
    cfg->max_activation_rate = INFINITY;                 // :143
    cfg->arm_target[0] = cfg->arm_target[1] = cfg->arm_target[2] = 500.0;   // :147
    return SOFTROD_OK;
}

int softrod_config_arm_push(softrod_config* cfg, int n_envs, int mode) {
    if (!cfg || n_envs < 1 || (mode != 0 && mode != 1)) return fail(nullptr, SOFTROD_EINVAL, "bad argument");
    config_common(cfg, n_envs);
    cfg->features = SOFTROD_FEATURES_ARM_PUSH;
    cfg->env_kind = SOFTROD_ENV_ARM_PUSH;
    cfg->arm_push_mode = mode;                           // octopus/arm_push_env.py:90-97
    cfg->n_elem = 40;                                    // :88
    cfg->dt = 5.0e-5;                                    // :67
    cfg->n_substeps = (int)(1.0 / (40 * cfg->dt));       // recording_fps = 40 (:68) -> 500 (:85)
    cfg->final_time = 2.5;                               // :66
    cfg->base_length = 0.2;                              // L0, :160
    cfg->base_radius = 0.012;                            // radius_base, :161 (the taper: softrod_set_radius_profile)
    cfg->density = 700.0;                                // :174
    cfg->youngs_modulus = 1e4;                           // :175
    cfg->shear_modulus = 1e4 / 1.5;                      // :176
    cfg->damping_constant = 0.05 * 2 * 1e2;              // damp_coefficient * 1e2, :166,183
    cfg->n_suckers = 1;                                  // :187-195
    cfg->sucker_index[0] = 0;
    cfg->sucker_reduction_ratio = 1.0;                   // controllable_constraint.py:11
    cfg->damp_before_constrain = 1;                      // _build registers dampen() before constrain() (:180-195)
    cfg->n_muscles = 3;                                  // create_es_muscle_layers, octopus/build.py:295-338
    cfg->muscle_kind[0] = SOFTROD_MUSCLE_LONGITUDINAL;
    cfg->muscle_kind[1] = SOFTROD_MUSCLE_LONGITUDINAL;
    cfg->muscle_kind[2] = SOFTROD_MUSCLE_TRANSVERSE;
    cfg->muscle_fl_degree = 3;                           // Chang et al. 2023: max{3.06 l^3 - 13.64 l^2 + 18.01 l - 6.44, 0}
    cfg->muscle_fl_coef[0] = -6.44; cfg->muscle_fl_coef[1] = 18.01; cfg->muscle_fl_coef[2] = -13.64; cfg->muscle_fl_coef[3] = 3.06;
    cfg->muscle_equiv_load_form = 0;
    cfg->muscle_position_current_radius = 1;
    cfg->muscle_tm_length_law = 0;
    return SOFTROD_OK;
}

int softrod_config_arm_pull_weight(softrod_config* cfg, int n_envs) {
    const int rc = softrod_config_arm_push(cfg, n_envs, 1);      // registered with mode "continuous" (gym_softrobot/__init__.py:48-52)
    if (rc != SOFTROD_OK) return rc;
    cfg->features = SOFTROD_FEATURES_ARM_PULL_WEIGHT;
    cfg->env_kind = SOFTROD_ENV_ARM_PULL_WEIGHT;
    cfg->dt = 2.5e-5;                                    // octopus/arm_push_env.py:518
    cfg->n_substeps = (int)(1.0 / (40 * cfg->dt));       // 1000
    cfg->damping_constant = 0.05 * 2 * 5e2;              // :549
    cfg->sucker_reduction_ratio = 0.9;                   // :591-599
    cfg->n_arm = 1; cfg->n_knots = 1;
    const double rigid_rod_radius = 0.015, radius_base = 0.012;     // :524,554
    cfg->head_radius = rigid_rod_radius;
    cfg->head_density = 700 * 1.0;                       // :565
    cfg->head_length = radius_base * 2;                  // :553
    cfg->head_center[0] = -rigid_rod_radius * 0.9;       // start (:555-558) + direction * length / 2
    cfg->head_center[1] = 0.0;
    cfg->head_center[2] = -2 * radius_base + radius_base * 2 / 2;
    cfg->joint_k = 1e6; cfg->joint_nu = 1e-2; cfg->joint_kt = 1e0;   // :577-589
    cfg->joint_angle0 = 0.0; cfg->joint_angle_step = 0.0;
    return SOFTROD_OK;
}

int softrod_config_muscle_octopus(softrod_config* cfg, int n_envs, int env_kind) {
    if (!cfg || n_envs < 1 || !mocto_kind(env_kind)) return fail(nullptr, SOFTROD_EINVAL, "bad argument");
    const int rc = softrod_config_arm_push(cfg, n_envs, 1);      // the three layers and the recalled COOMM switches
    if (rc != SOFTROD_OK) return rc;
    cfg->features = SOFTROD_FEATURES_ARM_PULL_WEIGHT;
    cfg->env_kind = env_kind;
    cfg->n_elem = 20;                                    // crawl_env.py:65, arm_two_env.py:59, reach_env.py:57
    cfg->dt = 5.0e-5;                                    // :63 / :57 / :55
    cfg->n_substeps = (int)(1.0 / (25 * cfg->dt));       // recording_fps 25 -> 800
    cfg->final_time = env_kind == SOFTROD_ENV_CRAWL ? 10.0 : 5.0;
    // build_muscle_octopus.py:26-47 ARM_MATERIAL / DEFAULT_SCALE_LENGTH / HEAD_PROPERTIES
    cfg->base_length = 0.25;
    cfg->base_radius = 0.013;                            // r0; the arms' radii: linspace(0.013, 0.0042, n) through softrod_set_radius_profile
    cfg->density = 1000.0;
    cfg->youngs_modulus = 1.5e4;
    cfg->shear_modulus = 1.5e4 / (1.0 + 0.5);
    cfg->damping_constant = 0.20 * 1e-2;                 // damping_constant * nu_scale (:101-106)
    cfg->damper_time_step = 7e-5;                        // the dampers' own time_step (:105)
    cfg->n_arm = env_kind == SOFTROD_ENV_ARM_TWO ? 2 : 8;
    cfg->n_knots = env_kind == SOFTROD_ENV_CRAWL ? 3 : (env_kind == SOFTROD_ENV_ARM_TWO ? 9 : 3 * cfg->n_elem);
    cfg->head_radius = 0.04;
    cfg->head_density = 50.0;
    cfg->head_length = 0.013 * 2;                        // Cylinder(start (0, 0, -2 r0), e_z, e_y, 2 r0, head_radius, density) (:108-114)
    cfg->head_center[0] = 0.0; cfg->head_center[1] = 0.0; cfg->head_center[2] = -0.013 * 2 + 0.013 * 2 / 2;
    cfg->head_fixed = env_kind == SOFTROD_ENV_REACH ? 1 : 0;      // OneEndFixedBC on the head (reach_env.py:126-130)
    cfg->joint_k = 1e6; cfg->joint_kt = 1e2; cfg->joint_nu = 1e-3;
    if (env_kind == SOFTROD_ENV_ARM_TWO) {               // build_two_arms (:200-203); three suckers per arm (arm_two_env.py:76-80,126-139)
        cfg->joint_angle0 = 90.0; cfg->joint_angle_step = 180.0;
        cfg->n_suckers = 3;
        for (int j = 0; j < 3; ++j) cfg->sucker_index[j] = cfg->n_elem / (3 * 2) * (2 * j + 1);
    } else {                                             // build_octopus_muscles (:83-86); CrawlEnv: one sucker per arm (crawl_env.py:146-155)
        cfg->joint_angle0 = 45.0 / 2; cfg->joint_angle_step = 45.0;
        cfg->n_suckers = env_kind == SOFTROD_ENV_CRAWL ? 1 : 0;
        cfg->sucker_index[0] = 0;
    }
    cfg->sucker_reduction_ratio = 1.0;
    cfg->damp_before_constrain = 1;                      // the builds dampen(), the envs constrain() afterwards
    return SOFTROD_OK;
}

int softrod_create(const softrod_config* cfg, int device, softrod_handle** out) {
    if (!cfg || !out) return fail(nullptr, SOFTROD_EINVAL, "null argument");
    *out = nullptr;
    if (cfg->struct_size != sizeof(softrod_config))
        return fail(nullptr, SOFTROD_EINVAL, "softrod_config.struct_size mismatch");
    if (cfg->n_envs < 1 || cfg->n_elem < 2 || cfg->n_elem > 2 * kLanes - 2)
        return fail(nullptr, SOFTROD_EINVAL, "need n_envs >= 1 and 2 <= n_elem <= 126");
    if (cfg->n_elem > kLanes - 1 && cfg->math_mode != SOFTROD_MATH_FAST)
        return fail(nullptr, SOFTROD_EINVAL,
                    "rods longer than 63 elements (two per lane) exist for SOFTROD_MATH_FAST only");
    if (cfg->n_substeps < 0 || !(cfg->dt > 0.0))
        return fail(nullptr, SOFTROD_EINVAL, "need n_substeps >= 0 and dt > 0");
    if (cfg->math_mode != SOFTROD_MATH_LIBM && cfg->math_mode != SOFTROD_MATH_FAST)
        return fail(nullptr, SOFTROD_EINVAL, "unknown math_mode");
    if (cfg->env_kind < SOFTROD_ENV_NONE || cfg->env_kind > SOFTROD_ENV_REACH)
        return fail(nullptr, SOFTROD_EINVAL, "unknown env_kind");
    if (cfg->features & SOFTROD_FEAT_COOMM_MUSCLES) {
        if (cfg->n_muscles < 1 || cfg->n_muscles > SOFTROD_MAX_MUSCLES || cfg->muscle_fl_degree < 0 ||
            cfg->muscle_fl_degree >= SOFTROD_MAX_FL_COEF || cfg->n_elem > kLanes - 1 ||
            ((cfg->features & SOFTROD_FEAT_OCTO_HEAD) && cfg->env_kind != SOFTROD_ENV_ARM_PULL_WEIGHT && !mocto_kind(cfg->env_kind)))
            return fail(nullptr, SOFTROD_EINVAL,
                        "COOMM muscles: 1 <= n_muscles <= 4, 0 <= muscle_fl_degree <= 7, one rod of up to 63 elements per env");
        for (int m = 0; m < cfg->n_muscles; ++m)
            if (cfg->muscle_kind[m] != SOFTROD_MUSCLE_LONGITUDINAL && cfg->muscle_kind[m] != SOFTROD_MUSCLE_TRANSVERSE)
                return fail(nullptr, SOFTROD_EINVAL, "muscle_kind: SOFTROD_MUSCLE_LONGITUDINAL or SOFTROD_MUSCLE_TRANSVERSE");
        if ((cfg->muscle_equiv_load_form | 1) != 1 || (cfg->muscle_position_current_radius | 1) != 1 ||
            (cfg->muscle_tm_length_law | 1) != 1)
            return fail(nullptr, SOFTROD_EINVAL, "muscle_equiv_load_form, muscle_position_current_radius, muscle_tm_length_law: 0 or 1");
    }
    if ((cfg->features & SOFTROD_FEAT_COOMM_MUSCLES) && cfg->math_mode == SOFTROD_MATH_FAST &&
        !((cfg->features == SOFTROD_FEATURES_ARM_PUSH && cfg->env_kind == SOFTROD_ENV_ARM_PUSH) ||
          (cfg->features == SOFTROD_FEATURES_ARM_PULL_WEIGHT && (cfg->env_kind == SOFTROD_ENV_ARM_PULL_WEIGHT || mocto_kind(cfg->env_kind))) ||
          (cfg->features == kFeaturesMuscleRod && cfg->env_kind == SOFTROD_ENV_NONE)))
        return fail(nullptr, SOFTROD_EINVAL,
                    "SOFTROD_MATH_FAST compiles the COOMM muscles for SOFTROD_FEATURES_ARM_PUSH with SOFTROD_ENV_ARM_PUSH "
                    "(tapered) and for FIXED_BC | ANALYTICAL_DAMPER | COOMM_MUSCLES with SOFTROD_ENV_NONE (uniform rod); "
                    "use SOFTROD_MATH_LIBM for any other mix");
    if (cfg->env_kind == SOFTROD_ENV_ARM_PULL_WEIGHT &&
        (cfg->features != SOFTROD_FEATURES_ARM_PULL_WEIGHT || cfg->math_mode != SOFTROD_MATH_FAST || cfg->n_arm != 1 ||
         !(cfg->head_length > 0.0) || !(cfg->head_radius > 0.0) || !(cfg->head_density > 0.0)))
        return fail(nullptr, SOFTROD_EINVAL, "SOFTROD_ENV_ARM_PULL_WEIGHT: SOFTROD_FEATURES_ARM_PULL_WEIGHT, SOFTROD_MATH_FAST, n_arm = 1, "
                                             "head_length / head_radius / head_density > 0");
    if (cfg->env_kind == SOFTROD_ENV_ARM_PUSH || cfg->env_kind == SOFTROD_ENV_ARM_PULL_WEIGHT) {
        const unsigned need = SOFTROD_FEAT_COOMM_MUSCLES | SOFTROD_FEAT_SUCKER_CONSTRAINT;
        if ((cfg->features & need) != need || cfg->n_muscles < 3 || (cfg->arm_push_mode != 0 && cfg->arm_push_mode != 1))
            return fail(nullptr, SOFTROD_EINVAL,
                        "SOFTROD_ENV_ARM_PUSH needs the sucker constraint, three muscle layers and arm_push_mode 0 or 1");
    }
    if (cfg->damper_protocol != 0 && cfg->damper_protocol != 1)
        return fail(nullptr, SOFTROD_EINVAL, "damper_protocol: 0 (per unit mass) or 1 (uniform)");
    // The fast kernels expand theta / sin(theta + eps_sin) as (theta / sin theta)(1 - eps_sin cot theta)
    // (eps_sin_factor, softrod_fast.hpp; the bke * rsq(D^2 + two_shift) term of softrod_planar.hpp), which
    // holds while eps_sin << theta_min = sqrt(2 acos_shift), the smallest angle acos(.. - acos_shift)
    // returns.  The reference's values (1e-14 against 1.4e-5) sit nine orders inside; a config outside
    // — acos_shift = 0 with a straight joint sends cot theta to 1e150 and flips the sign of the bending
    // stiffness — is refused here rather than integrated wrongly (the libm kernel evaluates the
    // quotient as written and takes any values).
    if (cfg->math_mode == SOFTROD_MATH_FAST &&
        !(cfg->acos_shift > 0.0 && cfg->eps_sin >= 0.0 && cfg->eps_sin <= 1.0e-3 * std::sqrt(2.0 * cfg->acos_shift)))
        return fail(nullptr, SOFTROD_EINVAL,
                    "SOFTROD_MATH_FAST needs acos_shift > 0 and 0 <= eps_sin <= 1e-3 sqrt(2 acos_shift); "
                    "use SOFTROD_MATH_LIBM for other values");
    {
        const bool muscles = (cfg->features & SOFTROD_FEAT_SPLINE_MUSCLE_TORQUES) != 0;
        if (muscles != (cfg->env_kind == SOFTROD_ENV_SOFT_ARM))
            return fail(nullptr, SOFTROD_EINVAL,
                        "SOFTROD_FEAT_SPLINE_MUSCLE_TORQUES and SOFTROD_ENV_SOFT_ARM go together");
        if (muscles && (cfg->math_mode != SOFTROD_MATH_FAST || cfg->n_ctrl < 1 || cfg->n_ctrl > 4 ||
                        cfg->n_spline_pieces < 1 || cfg->n_spline_pieces > SOFTROD_MAX_SPLINE_PIECES ||
                        cfg->n_elem - 1 < cfg->n_ctrl || !(cfg->max_activation_rate > 0.0)))
            return fail(nullptr, SOFTROD_EINVAL,
                        "spline muscles need SOFTROD_MATH_FAST, 1 <= n_ctrl <= 4, 1 <= n_spline_pieces <= 8, "
                        "max_activation_rate > 0");
    }
    const bool octo = (cfg->features & SOFTROD_FEAT_OCTO_HEAD) != 0;
    const bool pull = cfg->env_kind == SOFTROD_ENV_ARM_PULL_WEIGHT;
    const bool mocto = mocto_kind(cfg->env_kind);
    if (octo != (cfg->env_kind == SOFTROD_ENV_OCTO_FLAT || pull || mocto))
        return fail(nullptr, SOFTROD_EINVAL, "SOFTROD_FEAT_OCTO_HEAD goes with SOFTROD_ENV_OCTO_FLAT, SOFTROD_ENV_ARM_PULL_WEIGHT "
                                             "or the muscle octopus envs");
    if (mocto) {
        // n_arm * 32 slots = 1 or 4 whole waves, so that every slot of the block belongs to an arm
        if (cfg->features != SOFTROD_FEATURES_ARM_PULL_WEIGHT || cfg->math_mode != SOFTROD_MATH_FAST)
            return fail(nullptr, SOFTROD_EINVAL, "the muscle octopus exists for SOFTROD_FEATURES_ARM_PULL_WEIGHT and SOFTROD_MATH_FAST only");
        if (cfg->n_elem < 16 || cfg->n_elem > 31 || (cfg->n_arm != 2 && cfg->n_arm != 8) || cfg->n_muscles != 3)
            return fail(nullptr, SOFTROD_EINVAL, "the muscle octopus needs 16 <= n_elem <= 31, n_arm = 2 or 8, three muscle layers");
        const int nk = cfg->env_kind == SOFTROD_ENV_CRAWL ? 3 : (cfg->env_kind == SOFTROD_ENV_ARM_TWO ? 9 : 3 * cfg->n_elem);
        if (cfg->n_knots != nk || (cfg->env_kind == SOFTROD_ENV_ARM_TWO && cfg->n_suckers != 3) ||
            (cfg->env_kind == SOFTROD_ENV_CRAWL && cfg->n_suckers != 1) || (cfg->env_kind == SOFTROD_ENV_REACH && cfg->n_suckers != 0))
            return fail(nullptr, SOFTROD_EINVAL, "the muscle octopus: n_knots (actions per arm) 3 / 9 / 3 n_elem and n_suckers 1 / 3 / 0 "
                                                 "for CRAWL / ARM_TWO / REACH");
        if (!(cfg->head_radius > 0.0) || !(cfg->head_density > 0.0) || !(cfg->head_length > 0.0))
            return fail(nullptr, SOFTROD_EINVAL, "the muscle octopus needs head_radius, head_density, head_length > 0");
    }
    if (octo && !pull && !mocto) {
        if (cfg->features != SOFTROD_FEATURES_OCTO_FLAT || cfg->math_mode != SOFTROD_MATH_FAST)
            return fail(nullptr, SOFTROD_EINVAL,
                        "OctoFlat exists for SOFTROD_FEATURES_OCTO_FLAT and SOFTROD_MATH_FAST only");
        if (cfg->n_elem > kLanes - 1 || cfg->n_arm < 1 || cfg->n_knots < 1 || cfg->n_knots > cfg->n_elem ||
            (cfg->n_elem - 1) * cfg->n_knots > 2 * kLanes * 7)
            return fail(nullptr, SOFTROD_EINVAL, "OctoFlat needs n_elem <= 63, n_arm >= 1, 1 <= n_knots <= n_elem");
        const int seg = cfg->n_elem <= 15 ? 16 : (cfg->n_elem <= 31 ? 32 : 64);
        if (cfg->n_arm * seg > 8 * kLanes)
            return fail(nullptr, SOFTROD_EINVAL, "OctoFlat: n_arm * slots-per-arm must not exceed 512");
        if (!(cfg->head_radius > 0.0) || !(cfg->head_density > 0.0))
            return fail(nullptr, SOFTROD_EINVAL, "OctoFlat needs head_radius > 0 and head_density > 0");
    }
    if (cfg->features & SOFTROD_FEAT_SUCKER_CONSTRAINT) {
        if ((octo && !pull && !mocto) || cfg->n_suckers < (mocto ? 0 : 1) || cfg->n_suckers > SOFTROD_MAX_SUCKERS)
            return fail(nullptr, SOFTROD_EINVAL, "ControllableFixConstraint: 1 <= n_suckers <= 4, not with OctoFlat");
        for (int j = 0; j < cfg->n_suckers; ++j)
            if (cfg->sucker_index[j] < 0 || cfg->sucker_index[j] >= cfg->n_elem)
                return fail(nullptr, SOFTROD_EINVAL, "ControllableFixConstraint: 0 <= sucker_index < n_elem");
    }
    if ((cfg->features & SOFTROD_FEAT_LAPLACE_FILTER) && (cfg->filter_order < 1 || cfg->n_elem < 3))
        return fail(nullptr, SOFTROD_EINVAL, "LaplaceDissipationFilter needs filter_order >= 1");
    {
        const unsigned bcs = cfg->features & (SOFTROD_FEAT_PENDULUM_BC | SOFTROD_FEAT_FIXED_BC |
                                              SOFTROD_FEAT_MOVING_BASE_BC);
        if (bcs & (bcs - 1)) return fail(nullptr, SOFTROD_EINVAL, "at most one boundary condition");
    }
    int ndev = 0;
    const hipError_t cnt_err = hipGetDeviceCount(&ndev);
    if (cnt_err != hipSuccess || ndev < 1)
        return fail(nullptr, SOFTROD_ENODEV,
                    std::string("no HIP device visible: hipGetDeviceCount -> ") +
                        hipGetErrorString(cnt_err) + ", count " + std::to_string(ndev));
    if (device < 0 || device >= ndev) return fail(nullptr, SOFTROD_EINVAL, "device out of range");
    softrod_handle* h = new (std::nothrow) softrod_handle;
    if (!h) return fail(nullptr, SOFTROD_ENOMEM, "host allocation failed");
    h->cfg = *cfg;
    h->device = device;
    h->epl = cfg->n_elem > kLanes - 1 ? 2 : 1;
    fill_params(h->cfg, h->P);
    const size_t N = (size_t)cfg->n_envs;
    if (octo) {
        h->nw = (cfg->n_arm * h->P.seg + kLanes - 1) / kLanes;
        h->init_stride = (size_t)cfg->n_arm * 18 + (mocto ? 4 : 2);
    }
    // A/B switches for profiling and tests.  A product library must not change its kernel tier because of a
    // stray environment variable: they are read only when SOFTROD_DEBUG_SWITCHES=1 is set as well
    // (tests/test_gpu_debug_switches.py), and softrod_kernel_tier() reports what was selected.
    const char* dbg = std::getenv("SOFTROD_DEBUG_SWITCHES");
    const bool debug_switches = dbg && dbg[0] == '1';
    auto debug_env = [&](const char* name) -> const char* { return debug_switches ? std::getenv(name) : nullptr; };
    if (const char* one = debug_env("SOFTROD_OCTO_ONE_ENV_PER_BLOCK"))
        h->octo_one_env_per_block = one[0] == '1';
    if (const char* one = debug_env("SOFTROD_WINDOW_PAIRED"))              // softrod_window.hpp
        h->window_paired = one[0] != '0';
    if (const char* one = debug_env("SOFTROD_OCTO_ONE_WAVE"))              // softrod_octo1w.hpp
        h->octo_one_wave = one[0] == '1';
    {   // two-window form: ArmSingle with the e_z contact, 64..102 elements
        const int halo = kLanes - (cfg->n_elem + 2) / 2;     // the narrower of the two halos
        const char* off = debug_env("SOFTROD_NO_WINDOW");
        if (h->epl == 2 && cfg->features == SOFTROD_FEATURES_ARM_SINGLE && cfg->env_kind == SOFTROD_ENV_ARM_SINGLE &&
            cfg->math_mode == SOFTROD_MATH_FAST && (h->P.features & kFeatPlaneZup) && !(off && off[0] == '1') &&
            halo >= 3 * kWindowRho)
            h->window_refresh = halo / kWindowRho;    // the front (< 3.25 nodes per substep) stays in the halo
        if (const char* r = debug_env("SOFTROD_WINDOW_REFRESH")) if (h->window_refresh > 0) h->window_refresh = std::atoi(r);
    }
    const size_t adim = (size_t)softrod_config_action_dim(cfg);
    const size_t rowb = N * kLanes * h->epl * h->nw * sizeof(double);
    int rc = SOFTROD_OK;
    auto alloc = [&](void** p, size_t bytes) {
        if (rc != SOFTROD_OK) return;
        if (hipMalloc(p, bytes) != hipSuccess) { rc = SOFTROD_ENOMEM; return; }
        if (hipMemset(*p, 0, bytes) != hipSuccess) rc = SOFTROD_EHIP;
    };
    DeviceGuard guard_(device);
    if (guard_.err != hipSuccess) rc = SOFTROD_EHIP;
    alloc((void**)&h->S.pos, 3 * rowb);
    alloc((void**)&h->S.vel, 3 * rowb);
    alloc((void**)&h->S.dir, 9 * rowb);
    alloc((void**)&h->S.omg, 3 * rowb);
    alloc((void**)&h->S.tan, 3 * rowb);
    alloc((void**)&h->S.time, N * sizeof(double));
    alloc((void**)&h->S.bc, 12 * N * sizeof(double));
    alloc((void**)&h->S.ctrl, 4 * N * sizeof(double));
    alloc((void**)&h->S.kap, 3 * rowb);
    alloc((void**)&h->S.rkap, 3 * rowb);
    alloc((void**)&h->S.envmem, rowb);
    alloc((void**)&h->S.prev_action, N * (adim > 7 ? adim : 7) * sizeof(float));
    alloc((void**)&h->S.head, 20 * N * sizeof(double));
    alloc((void**)&h->d_spline, (size_t)(SOFTROD_MAX_SPLINE_PIECES + 1 + SOFTROD_MAX_SPLINE_PIECES * 4 * 4) * sizeof(double));
    h->P.spline = h->d_spline;
    alloc((void**)&h->d_ticket, sizeof(unsigned));    // softrod_scatter_rows' arrival counter, zero
    alloc((void**)&h->d_params, sizeof(RodParams));
    if (rc == SOFTROD_OK && hipMemcpy(h->d_params, &h->P, sizeof(RodParams), hipMemcpyHostToDevice) != hipSuccess)
        rc = SOFTROD_EHIP;
    h->S.params = h->d_params;
    alloc((void**)&h->d_state, sizeof(StatePtrs));
    h->S.self = h->d_state;
    if (rc == SOFTROD_OK && cfg->n_substeps > 0) {
        // the clock as `self.time = self.do_step(self.simulator, self.time, self.time_step)` accumulates it
        // (soft_pendulum.py:183-184): same additions, same order, IEEE doubles -> bit-identical
        constexpr int kTab = 1024;
        std::vector<double> tab((size_t)kTab);
        double t = 0.0;
        const double half = 0.5 * cfg->dt;
        for (int k = 0; k < kTab; ++k) {
            tab[(size_t)k] = t;
            for (int s = 0; s < cfg->n_substeps; ++s) {
                if (cfg->time_two_half_adds) { t = t + half; t = t + half; }
                else t = t + cfg->dt;
            }
        }
        alloc((void**)&h->d_time_tab, tab.size() * sizeof(double));
        if (rc == SOFTROD_OK && hipMemcpy(h->d_time_tab, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
            rc = SOFTROD_EHIP;
        h->S.time_tab = h->d_time_tab;
        h->P.tab_len = kTab;
        h->P.tab_n_sub = cfg->n_substeps;
        h->P.inv_step_time = 1.0 / ((double)cfg->n_substeps * cfg->dt);
        if (rc == SOFTROD_OK && hipMemcpy(h->d_params, &h->P, sizeof(RodParams), hipMemcpyHostToDevice) != hipSuccess)
            rc = SOFTROD_EHIP;
    }
    const size_t NS = N * (size_t)(mocto ? cfg->n_arm : 1);      // SuckerControllers: per env, per ARM of the muscle octopus
    alloc((void**)&h->d_sucker, (size_t)SOFTROD_MAX_SUCKERS * NS * sizeof(double));
    h->S.sucker = h->d_sucker;
    alloc((void**)&h->d_sucker_idx, (size_t)SOFTROD_MAX_SUCKERS * NS * sizeof(int));
    h->S.sucker_idx = h->d_sucker_idx;
    if (mocto) {
        alloc((void**)&h->d_aux, (size_t)8 * N * sizeof(double));
        alloc((void**)&h->d_prev_kappa, N * (size_t)cfg->n_arm * (size_t)(cfg->n_elem - 1) * sizeof(float));
        h->S.aux = h->d_aux;
        h->S.prev_kappa = h->d_prev_kappa;
    }
    if (rc == SOFTROD_OK) {
        std::vector<int> idx((size_t)SOFTROD_MAX_SUCKERS * NS, 0);
        for (int j = 0; j < SOFTROD_MAX_SUCKERS; ++j)
            for (size_t e = 0; e < NS; ++e) idx[(size_t)j * NS + e] = cfg->sucker_index[j];
        if (hipMemcpy(h->d_sucker_idx, idx.data(), idx.size() * sizeof(int), hipMemcpyHostToDevice) != hipSuccess)
            rc = SOFTROD_EHIP;
    }
    if (cfg->features & SOFTROD_FEAT_COOMM_MUSCLES) {
        alloc((void**)&h->d_mact, (size_t)SOFTROD_MAX_MUSCLES * rowb);
        alloc((void**)&h->d_mtab, (size_t)SOFTROD_MAX_MUSCLES * 4 * kLanes * sizeof(double));
        h->S.mact = h->d_mact;
        h->S.mtab = h->d_mtab;
    }
    if (rc == SOFTROD_OK && (cfg->features & SOFTROD_FEAT_SUCKER_CONSTRAINT)) {
        // the controllers are switched on after finalize (arm_push_env.py:222): effective ratio = the configured one
        std::vector<double> init((size_t)SOFTROD_MAX_SUCKERS * NS, 0.0);
        for (int j = 0; j < cfg->n_suckers; ++j)
            for (size_t e = 0; e < NS; ++e) init[(size_t)j * NS + e] = cfg->sucker_reduction_ratio;
        if (hipMemcpy(h->d_sucker, init.data(), init.size() * sizeof(double), hipMemcpyHostToDevice) != hipSuccess)
            rc = SOFTROD_EHIP;
    }
    alloc((void**)&h->d_basis, (size_t)2 * kLanes * 7 * sizeof(double));
    h->S.basis = h->d_basis;
    alloc((void**)&h->d_init, N * h->init_stride * sizeof(double));
    alloc((void**)&h->d_mask, N);
    if (rc == SOFTROD_OK && hipHostMalloc((void**)&h->h_init, N * h->init_stride * sizeof(double)) != hipSuccess) rc = SOFTROD_ENOMEM;
    if (rc == SOFTROD_OK && hipHostMalloc((void**)&h->h_mask, N) != hipSuccess) rc = SOFTROD_ENOMEM;
    if (rc == SOFTROD_OK && hipEventCreateWithFlags(&h->ev_reset, hipEventDisableTiming) != hipSuccess)
        rc = SOFTROD_EHIP;
    if (rc == SOFTROD_OK && hipMemcpy(h->d_state, &h->S, sizeof(StatePtrs), hipMemcpyHostToDevice) != hipSuccess)
        rc = SOFTROD_EHIP;
    if (rc != SOFTROD_OK) {
        softrod_destroy(h);
        return fail(nullptr, rc, "device allocation failed");
    }
    *out = h;
    return SOFTROD_OK;
}

int softrod_reset_octo(softrod_handle* h, const double* arm_start, const double* arm_direction,
                       const double* target, const uint8_t* mask, void* stream) {
    if (!h || !arm_start || !arm_direction || !target) return fail(h, SOFTROD_EINVAL, "null argument");
    if (!is_flat(h) && !is_mocto(h)) return fail(h, SOFTROD_EINVAL, "softrod_reset_octo is for SOFTROD_ENV_OCTO_FLAT and the muscle octopus envs");
    SR_ON_DEVICE(h);
    SR_HIP(h, hipEventSynchronize(h->ev_reset));
    const int N = h->cfg.n_envs, na = h->cfg.n_arm;
    const double normal[3] = {0.0, 0.0, 1.0};             // octopus/build.py:82, build_muscle_octopus.py:92
    double* tgt = h->h_init + (size_t)N * na * 18;
    const size_t td = is_mocto(h) ? 4 : 2;                // the muscle octopus: target x, y, z and the episode's final_time (0: the config's)
    for (int e = 0; e < N; ++e) {
        if (mask) h->h_mask[e] = mask[e];
        if (mask && !mask[e]) continue;
        for (int a = 0; a < na; ++a) {
            const size_t k = (size_t)e * na + a;
            straight_init(h->cfg, arm_start + 3 * k, arm_direction + 3 * k, normal, h->h_init + k * 18);
        }
        for (size_t i = 0; i < td; ++i) tgt[td * (size_t)e + i] = target[td * (size_t)e + i];
    }
    return upload_and_reset(h, (hipStream_t)stream, mask != nullptr);
}

// ---- device-side auto-reset ------------------------------------------------------------------
namespace {
// Everything softrod_autoreset_enable allocates, released (idempotent; also softrod_destroy's path).
void autoreset_release(softrod_handle* h) {
    void* dbufs[] = {h->d_queue, h->d_consumed, h->d_produced, h->d_underflow, h->d_flags, h->d_stage, h->d_where};
    for (void* p : dbufs) if (p) (void)hipFree(p);
    void* hbufs[] = {h->h_queue, h->h_produced, h->h_stage, h->h_where, h->h_status};
    for (void* p : hbufs) if (p) (void)hipHostFree(p);
    if (h->ev_queue) (void)hipEventDestroy(h->ev_queue);
    if (h->ev_status) (void)hipEventDestroy(h->ev_status);
    h->d_queue = nullptr; h->d_consumed = nullptr; h->d_produced = nullptr; h->d_underflow = nullptr;
    h->d_flags = nullptr; h->d_stage = nullptr; h->d_where = nullptr;
    h->h_queue = nullptr; h->h_produced = nullptr; h->h_stage = nullptr; h->h_where = nullptr; h->h_status = nullptr;
    h->ev_queue = nullptr; h->ev_status = nullptr;
    h->stage_cap = 0;
    h->q_depth = 0;
}

int autoreset_allocate(softrod_handle* h, int depth, int fail_at) {
    int calls = 0;
    // fail_at > 0: the fail_at-th HIP call of this function "fails" without being made (failure injection
    // of tests/test_gpu_debug_switches.py; 0 in production)
#define SR_Q(call)                                                                                  \
    do {                                                                                            \
        if (++calls == fail_at) return fail(h, SOFTROD_EHIP, "softrod_autoreset_enable: injected failure at call " + std::to_string(calls)); \
        SR_HIP(h, call);                                                                            \
    } while (0)
    const size_t N = (size_t)h->cfg.n_envs;
    const size_t qb = (size_t)depth * N * h->init_stride * sizeof(double);
    SR_Q(hipMalloc((void**)&h->d_queue, qb));
    SR_Q(hipMemset(h->d_queue, 0, qb));
    SR_Q(hipHostMalloc((void**)&h->h_queue, qb));
    SR_Q(hipMalloc((void**)&h->d_consumed, N * sizeof(int)));
    SR_Q(hipMalloc((void**)&h->d_produced, N * sizeof(int)));
    SR_Q(hipMalloc((void**)&h->d_underflow, sizeof(int)));
    SR_Q(hipMalloc((void**)&h->d_flags, 2 * N));
    SR_Q(hipMemset(h->d_consumed, 0, N * sizeof(int)));
    SR_Q(hipMemset(h->d_produced, 0, N * sizeof(int)));
    SR_Q(hipMemset(h->d_underflow, 0, sizeof(int)));
    SR_Q(hipMemset(h->d_flags, 0, 2 * N));
    SR_Q(hipHostMalloc((void**)&h->h_produced, N * sizeof(int)));
    std::memset(h->h_produced, 0, N * sizeof(int));
    SR_Q(hipEventCreateWithFlags(&h->ev_queue, hipEventDisableTiming));
    h->stage_cap = 2 * N;     // pushes of more records than this upload the whole ring instead
    SR_Q(hipHostMalloc((void**)&h->h_stage, h->stage_cap * h->init_stride * sizeof(double)));
    SR_Q(hipHostMalloc((void**)&h->h_where, h->stage_cap * sizeof(int2)));
    SR_Q(hipMalloc((void**)&h->d_stage, h->stage_cap * h->init_stride * sizeof(double)));
    SR_Q(hipMalloc((void**)&h->d_where, h->stage_cap * sizeof(int2)));
    SR_Q(hipHostMalloc((void**)&h->h_status, (N + 1) * sizeof(int)));
    SR_Q(hipEventCreateWithFlags(&h->ev_status, hipEventDisableTiming));
    // the device copy of the state pointers is the LAST thing to change: a failure above leaves it
    // (and h->S) exactly as it was, so the handle keeps stepping without auto-reset
    StatePtrs S = h->S;
    S.needs_reset = h->d_flags;
    S.skip = h->d_flags + N;
    S.queue = h->d_queue;
    S.q_consumed = h->d_consumed;
    S.q_produced = h->d_produced;
    S.q_underflow = h->d_underflow;
    S.q_depth = depth;
    S.q_record = (int)h->init_stride;
    SR_Q(hipMemcpy(h->d_state, &S, sizeof(StatePtrs), hipMemcpyHostToDevice));
    h->S = S;
    h->seen_consumed.assign(N, 0);
    h->q_depth = depth;
    return SOFTROD_OK;
#undef SR_Q
}
}  // namespace

int softrod_autoreset_enable(softrod_handle* h, int depth) {
    if (!h || depth < 1 || depth > 4096) return fail(h, SOFTROD_EINVAL, "need 1 <= depth <= 4096");
    if (h->q_depth) return fail(h, SOFTROD_EINVAL, "auto-reset is already enabled");
    if (h->cfg.env_kind == SOFTROD_ENV_NONE) return fail(h, SOFTROD_EINVAL, "env_kind NONE has no episodes");
    SR_ON_DEVICE(h);
    // SOFTROD_DEBUG_FAIL_AUTORESET_CALL=k (honoured only with SOFTROD_DEBUG_SWITCHES=1): the k-th HIP call of
    // the set-up fails, for the test that a failed enable leaves nothing allocated and can be retried
    int fail_at = 0;
    {
        const char* dbg = std::getenv("SOFTROD_DEBUG_SWITCHES");
        const char* k = (dbg && dbg[0] == '1') ? std::getenv("SOFTROD_DEBUG_FAIL_AUTORESET_CALL") : nullptr;
        if (k) fail_at = std::atoi(k);
    }
    const int rc = autoreset_allocate(h, depth, fail_at);
    if (rc != SOFTROD_OK) {
        const std::string why = h->err;
        autoreset_release(h);      // nothing of the failed attempt stays allocated; a retry starts clean
        h->err = why;
    }
    return rc;
}

namespace {
// Next free record of env e, or nullptr if staging `count` more would overwrite records the
// device may not have consumed yet (as far as the last softrod_queue_status knows).
int queue_begin(softrod_handle* h, const int32_t* counts, int max_count) {
    if (!h->q_depth) return fail(h, SOFTROD_EINVAL, "call softrod_autoreset_enable first");
    if (!counts || max_count < 0) return fail(h, SOFTROD_EINVAL, "bad argument");
    const int N = h->cfg.n_envs;
    for (int e = 0; e < N; ++e) {
        if (counts[e] < 0 || counts[e] > max_count) return fail(h, SOFTROD_EINVAL, "counts[e] out of range");
        if (h->h_produced[e] + counts[e] - h->seen_consumed[e] > h->q_depth)
            return fail(h, SOFTROD_EINVAL, "reset queue overflow: staged + new records exceed depth");
    }
    SR_HIP(h, hipEventSynchronize(h->ev_queue));   // previous upload out of the pinned mirror
    return SOFTROD_OK;
}
double* queue_slot(softrod_handle* h, int e, int k) {
    h->pending.push_back(make_int2(e, k % h->q_depth));
    return h->h_queue + ((size_t)(k % h->q_depth) * (size_t)h->cfg.n_envs + (size_t)e) * h->init_stride;
}
int queue_commit(softrod_handle* h, hipStream_t st) {
    const size_t N = (size_t)h->cfg.n_envs;
    const size_t M = h->pending.size(), R = h->init_stride;
    if (M > 0 && M <= h->stage_cap) {
        // only the records of this push cross PCIe: compacted copy + a scatter kernel
        for (size_t m = 0; m < M; ++m) {
            const int2 w = h->pending[m];
            std::memcpy(h->h_stage + m * R, h->h_queue + ((size_t)w.y * N + (size_t)w.x) * R, R * sizeof(double));
            h->h_where[m] = w;
        }
        SR_HIP(h, hipMemcpyAsync(h->d_stage, h->h_stage, M * R * sizeof(double), hipMemcpyHostToDevice, st));
        SR_HIP(h, hipMemcpyAsync(h->d_where, h->h_where, M * sizeof(int2), hipMemcpyHostToDevice, st));
        hipLaunchKernelGGL(softrod_queue_scatter_kernel, dim3((unsigned)M), dim3(kLanes), 0, st, h->d_queue, h->d_stage,
                           h->d_where, (int)N, (int)R);
        SR_HIP(h, hipGetLastError());
    } else if (M > 0) {
        SR_HIP(h, hipMemcpyAsync(h->d_queue, h->h_queue, (size_t)h->q_depth * N * R * sizeof(double),
                                 hipMemcpyHostToDevice, st));
    }
    h->pending.clear();
    SR_HIP(h, hipMemcpyAsync(h->d_produced, h->h_produced, N * sizeof(int), hipMemcpyHostToDevice, st));
    SR_HIP(h, hipEventRecord(h->ev_queue, st));
    return SOFTROD_OK;
}
}  // namespace

int softrod_queue_push(softrod_handle* h, const double* theta0, const int32_t* counts, int max_count,
                       void* stream) {
    if (!h || !theta0) return fail(h, SOFTROD_EINVAL, "null argument");
    if (is_octo(h)) return fail(h, SOFTROD_EINVAL, "rigid-body envs stage resets through softrod_queue_push_octo / _straight");
    SR_ON_DEVICE(h);
    const int rc = queue_begin(h, counts, max_count);
    if (rc != SOFTROD_OK) return rc;
    for (int e = 0; e < h->cfg.n_envs; ++e)
        for (int j = 0; j < counts[e]; ++j) {
            const double th = theta0[(size_t)e * max_count + j];
            const double start[3] = {0.0, 0.0, 0.0};
            const double direction[3] = {1.0 * std::cos(th), 1.0 * std::sin(th), 0.0};
            const double normal[3] = {1.0 * std::sin(th), -1.0 * std::cos(th), 0.0};
            straight_init(h->cfg, start, direction, normal, queue_slot(h, e, h->h_produced[e]++));
        }
    return queue_commit(h, (hipStream_t)stream);
}

int softrod_queue_push_straight(softrod_handle* h, const double* start, const double* direction,
                                const double* normal, const int32_t* counts, int max_count, void* stream) {
    if (!h || !start || !direction || !normal) return fail(h, SOFTROD_EINVAL, "null argument");
    if (is_flat(h)) return fail(h, SOFTROD_EINVAL, "OctoFlat stages resets through softrod_queue_push_octo");
    SR_ON_DEVICE(h);
    const int rc = queue_begin(h, counts, max_count);
    if (rc != SOFTROD_OK) return rc;
    for (int e = 0; e < h->cfg.n_envs; ++e)
        for (int j = 0; j < counts[e]; ++j) {
            const size_t k = ((size_t)e * max_count + j) * 3;
            double* rec = queue_slot(h, e, h->h_produced[e]++);
            straight_init(h->cfg, start + k, direction + k, normal + k, rec);
            if (is_pull(h)) { rec[18] = 0.0; rec[19] = 0.0; }      // the rigid-body record's target slot: unused
        }
    return queue_commit(h, (hipStream_t)stream);
}

int softrod_queue_push_octo(softrod_handle* h, const double* arm_start, const double* arm_direction,
                            const double* target, const int32_t* counts, int max_count, void* stream) {
    if (!h || !arm_start || !arm_direction || !target) return fail(h, SOFTROD_EINVAL, "null argument");
    if (!is_flat(h) && !is_mocto(h)) return fail(h, SOFTROD_EINVAL, "softrod_queue_push_octo is for SOFTROD_ENV_OCTO_FLAT and the muscle octopus envs");
    SR_ON_DEVICE(h);
    const int rc = queue_begin(h, counts, max_count);
    if (rc != SOFTROD_OK) return rc;
    const int na = h->cfg.n_arm;
    const double normal[3] = {0.0, 0.0, 1.0};
    for (int e = 0; e < h->cfg.n_envs; ++e)
        for (int j = 0; j < counts[e]; ++j) {
            double* rec = queue_slot(h, e, h->h_produced[e]++);
            const size_t k = (size_t)e * max_count + j;
            for (int a = 0; a < na; ++a)
                straight_init(h->cfg, arm_start + (k * na + a) * 3, arm_direction + (k * na + a) * 3, normal,
                              rec + (size_t)a * 18);
            const size_t td = is_mocto(h) ? 4 : 2;
            for (size_t i = 0; i < td; ++i) rec[(size_t)na * 18 + i] = target[td * k + i];
        }
    return queue_commit(h, (hipStream_t)stream);
}

int softrod_queue_status(softrod_handle* h, int32_t* consumed, int32_t* underflow, void* stream) {
    if (!h || !consumed) return fail(h, SOFTROD_EINVAL, "null argument");
    if (!h->q_depth) return fail(h, SOFTROD_EINVAL, "call softrod_autoreset_enable first");
    SR_ON_DEVICE(h);
    const size_t N = (size_t)h->cfg.n_envs;
    int uf = 0;
    SR_HIP(h, hipMemcpyAsync(consumed, h->d_consumed, N * sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    SR_HIP(h, hipMemcpyAsync(&uf, h->d_underflow, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    SR_HIP(h, hipStreamSynchronize((hipStream_t)stream));
    for (size_t e = 0; e < N; ++e) h->seen_consumed[e] = consumed[e];
    if (underflow) *underflow = uf;
    return SOFTROD_OK;
}

int softrod_queue_status_begin(softrod_handle* h, void* stream) {
    if (!h) return fail(h, SOFTROD_EINVAL, "null argument");
    if (!h->q_depth) return fail(h, SOFTROD_EINVAL, "call softrod_autoreset_enable first");
    SR_ON_DEVICE(h);
    const size_t N = (size_t)h->cfg.n_envs;
    SR_HIP(h, hipMemcpyAsync(h->h_status, h->d_consumed, N * sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    SR_HIP(h, hipMemcpyAsync(h->h_status + N, h->d_underflow, sizeof(int), hipMemcpyDeviceToHost, (hipStream_t)stream));
    SR_HIP(h, hipEventRecord(h->ev_status, (hipStream_t)stream));
    h->status_pending = true;
    return SOFTROD_OK;
}

int softrod_queue_status_poll(softrod_handle* h, int wait, int32_t* consumed, int32_t* underflow) {
    if (!h || !consumed) return fail(h, SOFTROD_EINVAL, "null argument");
    if (!h->status_pending) return fail(h, SOFTROD_EINVAL, "no softrod_queue_status_begin outstanding");
    SR_ON_DEVICE(h);
    const hipError_t q = wait ? hipEventSynchronize(h->ev_status) : hipEventQuery(h->ev_status);
    if (q == hipErrorNotReady) return 0;
    if (q != hipSuccess) return fail(h, SOFTROD_EHIP, std::string("queue status event: ") + hipGetErrorString(q));
    const size_t N = (size_t)h->cfg.n_envs;
    for (size_t e = 0; e < N; ++e) { consumed[e] = h->h_status[e]; h->seen_consumed[e] = h->h_status[e]; }
    if (underflow) *underflow = h->h_status[N];
    h->status_pending = false;
    return 1;
}

int softrod_queue_advance(softrod_handle* h, const int32_t* by, void* stream) {
    if (!h || !by) return fail(h, SOFTROD_EINVAL, "null argument");
    if (!h->q_depth) return fail(h, SOFTROD_EINVAL, "call softrod_autoreset_enable first");
    SR_ON_DEVICE(h);
    const int N = h->cfg.n_envs;
    // consumed += by (by < 0: := produced).  The device counters are only ever written by the
    // auto-reset pass, which is stream-ordered with these copies.
    std::vector<int> cons((size_t)N);
    SR_HIP(h, hipMemcpyAsync(cons.data(), h->d_consumed, (size_t)N * sizeof(int), hipMemcpyDeviceToHost,
                             (hipStream_t)stream));
    SR_HIP(h, hipStreamSynchronize((hipStream_t)stream));
    for (int e = 0; e < N; ++e) {
        int c = by[e] < 0 ? h->h_produced[e] : cons[(size_t)e] + by[e];
        if (c > h->h_produced[e]) c = h->h_produced[e];
        cons[(size_t)e] = c;
        h->seen_consumed[(size_t)e] = c;
    }
    SR_HIP(h, hipMemcpyAsync(h->d_consumed, cons.data(), (size_t)N * sizeof(int), hipMemcpyHostToDevice,
                             (hipStream_t)stream));
    SR_HIP(h, hipStreamSynchronize((hipStream_t)stream));
    return SOFTROD_OK;
}

int softrod_reset(softrod_handle* h, const double* theta0, const uint8_t* mask, void* stream) {
    if (!h || !theta0) return fail(h, SOFTROD_EINVAL, "null argument");
    if (is_octo(h)) return fail(h, SOFTROD_EINVAL, "rigid-body envs reset through softrod_reset_octo / softrod_reset_straight");
    SR_ON_DEVICE(h);
    SR_HIP(h, hipEventSynchronize(h->ev_reset));  // previous upload out of the pinned buffers
    const int N = h->cfg.n_envs;
    for (int e = 0; e < N; ++e) {
        if (mask) h->h_mask[e] = mask[e];
        if (mask && !mask[e]) continue;
        const double th = theta0[e];
        // build.py:46-52
        const double start[3] = {0.0, 0.0, 0.0};
        const double direction[3] = {1.0 * std::cos(th), 1.0 * std::sin(th), 0.0};
        const double normal[3] = {1.0 * std::sin(th), -1.0 * std::cos(th), 0.0};
        straight_init(h->cfg, start, direction, normal, h->h_init + (size_t)e * 18);
    }
    return upload_and_reset(h, (hipStream_t)stream, mask != nullptr);
}

int softrod_reset_straight(softrod_handle* h, const double* start, const double* direction,
                           const double* normal, const uint8_t* mask, void* stream) {
    if (!h || !start || !direction || !normal) return fail(h, SOFTROD_EINVAL, "null argument");
    if (is_flat(h)) return fail(h, SOFTROD_EINVAL, "OctoFlat resets through softrod_reset_octo");
    SR_ON_DEVICE(h);
    SR_HIP(h, hipEventSynchronize(h->ev_reset));
    const int N = h->cfg.n_envs;
    // ArmPullWeightEnv: the one arm's record where the rigid-body reset kernel expects the arms', a zero target behind them
    double* tgt = h->h_init + (size_t)N * 18;
    for (int e = 0; e < N; ++e) {
        if (mask) h->h_mask[e] = mask[e];
        if (mask && !mask[e]) continue;
        straight_init(h->cfg, start + 3 * e, direction + 3 * e, normal + 3 * e, h->h_init + (size_t)e * 18);
        if (is_pull(h)) { tgt[2 * (size_t)e] = 0.0; tgt[2 * (size_t)e + 1] = 0.0; }
    }
    return upload_and_reset(h, (hipStream_t)stream, mask != nullptr);
}

int softrod_set_spline_table(softrod_handle* h, const double* breaks, const double* coef) {
    if (!h || !breaks || !coef) return fail(h, SOFTROD_EINVAL, "null argument");
    if (!(h->cfg.features & SOFTROD_FEAT_SPLINE_MUSCLE_TORQUES))
        return fail(h, SOFTROD_EINVAL, "this handle has no spline muscles");
    const int np = h->cfg.n_spline_pieces, nc = h->cfg.n_ctrl;
    for (int p = 0; p < np; ++p)
        if (!(breaks[p + 1] > breaks[p])) return fail(h, SOFTROD_EINVAL, "breaks must ascend");
    std::vector<double> buf((size_t)SOFTROD_MAX_SPLINE_PIECES + 1 + (size_t)np * nc * 4, 0.0);
    for (int p = 0; p <= np; ++p) buf[p] = breaks[p];
    for (int i = 0; i < np * nc * 4; ++i) buf[SOFTROD_MAX_SPLINE_PIECES + 1 + i] = coef[i];
    SR_ON_DEVICE(h);
    SR_HIP(h, hipMemcpy(h->d_spline, buf.data(), buf.size() * sizeof(double), hipMemcpyHostToDevice));
    h->spline_set = true;
    return SOFTROD_OK;
}

// CosseratRod.straight_rod's allocation (elastica/rod/factory_function.py allocate(), recalled)
// with base_radius an ARRAY of n_elements radii (octopus/arm_push_env.py:160-179): the rows of
// kernels' per-lane material table.  Mirrors straight_rod() of oracle/softrod_oracle.c.
int softrod_set_radius_profile(softrod_handle* h, const double* radius) {
    if (!h || !radius) return fail(h, SOFTROD_EINVAL, "null argument");
    if ((is_octo(h) && !is_pull(h) && !is_mocto(h)) || h->epl != 1 || h->window_refresh > 0)
        return fail(h, SOFTROD_EINVAL, "tapered rods: one rod of up to 63 elements per env, or the muscle octopus' arms");
    if (h->was_reset) return fail(h, SOFTROD_EINVAL, "softrod_set_radius_profile must precede the first reset");
    const softrod_config& c = h->cfg;
    const int n = c.n_elem;
    for (int k = 0; k < n; ++k)
        if (!(radius[k] > 0.0)) return fail(h, SOFTROD_EINVAL, "radii must be positive");
    SR_ON_DEVICE(h);
    constexpr int W = kLanes;
    // the muscle octopus: every arm is this rod, `seg` slots apart — the table is built for one arm's slots and repeated
    const int period = is_mocto(h) ? h->P.seg : W;
    const double ddt = c.damper_time_step > 0.0 ? c.damper_time_step : c.dt;
    std::vector<double> T((size_t)kMatRows * W, 0.0);
    const double rest_len = c.base_length / (double)n;
    std::vector<double> mass((size_t)n + 1, 0.0), bend01((size_t)n), bend2((size_t)n);
    for (int k = 0; k < n; ++k) {
        const double r = radius[k];
        const double A0 = M_PI * r * r;
        const double I1 = A0 * A0 / (4.0 * M_PI), I3 = 2.0 * I1;
        const double J0 = I1 * (c.density * rest_len), J2 = I3 * (c.density * rest_len);
        T[(size_t)kMatJ0 * W + k] = J0; T[(size_t)kMatJ2 * W + k] = J2;
        T[(size_t)kMatInvJ0 * W + k] = 1.0 / J0; T[(size_t)kMatInvJ2 * W + k] = 1.0 / J2;
        T[(size_t)kMatShear01 * W + k] = c.alpha_c * c.shear_modulus * A0;
        T[(size_t)kMatShear2 * W + k] = c.youngs_modulus * A0;
        bend01[(size_t)k] = c.youngs_modulus * I1;
        bend2[(size_t)k] = c.shear_modulus * I3;
        const double volume = M_PI * (r * r) * rest_len;
        mass[(size_t)k] += 0.5 * c.density * volume;
        mass[(size_t)k + 1] += 0.5 * c.density * volume;
        T[(size_t)kMatR0s * W + k] = r * std::sqrt(rest_len);
        T[(size_t)kMatInvR0s * W + k] = 1.0 / (r * std::sqrt(rest_len));
    }
    for (int k = 0; k < n - 1; ++k) {      // rest-length-weighted average onto the Voronoi vertices
        T[(size_t)kMatBend01 * W + k] = (bend01[(size_t)k + 1] * rest_len + bend01[(size_t)k] * rest_len) / (rest_len + rest_len);
        T[(size_t)kMatBend2 * W + k] = (bend2[(size_t)k + 1] * rest_len + bend2[(size_t)k] * rest_len) / (rest_len + rest_len);
    }
    double ms = 0.0;
    for (int k = 0; k <= n; ++k) { T[(size_t)kMatMass * W + k] = mass[(size_t)k]; ms += mass[(size_t)k]; }
    for (int k = n + 1; k < period; ++k) T[(size_t)kMatMass * W + k] = 1.0;     // finite filler past the rod
    for (int k = 0; k < n; ++k) {          // AnalyticalLinearDamper's per-element coefficients
        double me = 0.5 * (mass[(size_t)k + 1] + mass[(size_t)k]);
        if (k == 0) me += 0.5 * mass[0];
        if (k == n - 1) me += 0.5 * mass[(size_t)n];
        const double l0 = c.damper_protocol == 1 ? -c.damping_constant * ddt : -c.damping_constant * ddt * me * T[(size_t)kMatInvJ0 * W + k];
        const double l2 = c.damper_protocol == 1 ? -c.damping_constant * ddt : -c.damping_constant * ddt * me * T[(size_t)kMatInvJ2 * W + k];
        T[(size_t)kMatDampLog0 * W + k] = l0; T[(size_t)kMatDampLog2 * W + k] = l2;
        T[(size_t)kMatDampR0 * W + k] = std::exp(l0); T[(size_t)kMatDampR2 * W + k] = std::exp(l2);
    }
    for (int k = n; k < period; ++k) {      // slots past the last element: harmless finite values
        T[(size_t)kMatInvJ0 * W + k] = 0.0; T[(size_t)kMatInvJ2 * W + k] = 0.0;
        T[(size_t)kMatDampR0 * W + k] = 1.0; T[(size_t)kMatDampR2 * W + k] = 1.0;
        T[(size_t)kMatR0s * W + k] = 1.0; T[(size_t)kMatInvR0s * W + k] = 1.0;
    }
    for (int row = 0; row < kMatRows; ++row)
        for (int k = period; k < W; ++k) T[(size_t)row * W + k] = T[(size_t)row * W + (k % period)];
    if (!h->d_mat) SR_HIP(h, hipMalloc((void**)&h->d_mat, T.size() * sizeof(double)));
    SR_HIP(h, hipMemcpy(h->d_mat, T.data(), T.size() * sizeof(double), hipMemcpyHostToDevice));
    h->S.mat = h->d_mat;
    h->P.mass_total = ms;
    h->tapered = true;
    SR_HIP(h, hipMemcpy(h->d_params, &h->P, sizeof(RodParams), hipMemcpyHostToDevice));
    SR_HIP(h, hipMemcpy(h->d_state, &h->S, sizeof(StatePtrs), hipMemcpyHostToDevice));
    return SOFTROD_OK;
}

int softrod_set_muscle_layers(softrod_handle* h, const double* ratio_position, const double* strength) {
    if (!h || !ratio_position || !strength) return fail(h, SOFTROD_EINVAL, "null argument");
    if (!(h->cfg.features & SOFTROD_FEAT_COOMM_MUSCLES)) return fail(h, SOFTROD_EINVAL, "this handle has no COOMM muscles");
    const int n = h->cfg.n_elem, M = h->cfg.n_muscles;
    constexpr int W = kLanes;
    std::vector<double> T((size_t)SOFTROD_MAX_MUSCLES * 4 * W, 0.0);
    for (int m = 0; m < M; ++m)
        for (int k = 0; k < n; ++k) {
            for (int i = 0; i < 3; ++i) {
                const double v = ratio_position[((size_t)m * 3 + i) * n + k];
                if (!std::isfinite(v)) return fail(h, SOFTROD_EINVAL, "ratio_position must be finite");
                T[((size_t)m * 4 + i) * W + k] = v;
            }
            const double st = strength[(size_t)m * n + k];
            if (!std::isfinite(st)) return fail(h, SOFTROD_EINVAL, "strength must be finite");
            T[((size_t)m * 4 + 3) * W + k] = st;
        }
    if (is_mocto(h))                        // every arm carries the same layers, `seg` slots apart
        for (int row = 0; row < SOFTROD_MAX_MUSCLES * 4; ++row)
            for (int k = h->P.seg; k < W; ++k) T[(size_t)row * W + k] = T[(size_t)row * W + (k % h->P.seg)];
    SR_ON_DEVICE(h);
    SR_HIP(h, hipMemcpy(h->d_mtab, T.data(), T.size() * sizeof(double), hipMemcpyHostToDevice));
    h->muscles_set = true;
    return SOFTROD_OK;
}

int softrod_set_action_basis(softrod_handle* h, const double* basis) {
    if (!h || !basis) return fail(h, SOFTROD_EINVAL, "null argument");
    SR_ON_DEVICE(h);
    // SOFTROD_ENV_ARM_TWO: [n_elem][3], apply_X = basis @ activation (arm_two_env.py:237-245)
    if (is_mocto(h) && h->cfg.env_kind != SOFTROD_ENV_ARM_TWO) return fail(h, SOFTROD_EINVAL, "this env interpolates nothing");
    const size_t bytes = is_mocto(h) ? (size_t)h->cfg.n_elem * 3 * sizeof(double)
                                     : (size_t)(h->cfg.n_elem - 1) * (is_octo(h) ? h->cfg.n_knots : 7) * sizeof(double);
    SR_HIP(h, hipMemcpy(h->d_basis, basis, bytes, hipMemcpyHostToDevice));
    h->basis_set = true;
    return SOFTROD_OK;
}

int softrod_step(softrod_handle* h, const float* actions, float* obs, double* reward,
                 uint8_t* terminated, uint8_t* truncated, double* aux, void* stream) {
    if (!h || !actions || !obs || !reward || !terminated || !truncated)
        return fail(h, SOFTROD_EINVAL, "null argument");
    if ((h->cfg.features & SOFTROD_FEAT_REST_KAPPA_ACTION) && !h->basis_set)
        return fail(h, SOFTROD_EINVAL, "softrod_set_action_basis must be called before softrod_step");
    if ((h->cfg.features & SOFTROD_FEAT_SPLINE_MUSCLE_TORQUES) && !h->spline_set)
        return fail(h, SOFTROD_EINVAL, "softrod_set_spline_table must be called before softrod_step");
    if (h->cfg.env_kind == SOFTROD_ENV_NONE)
        return fail(h, SOFTROD_EINVAL, "env_kind NONE has no step epilogue; use softrod_substeps");
    if ((h->cfg.features & SOFTROD_FEAT_COOMM_MUSCLES) && !h->muscles_set)
        return fail(h, SOFTROD_EINVAL, "softrod_set_muscle_layers must be called before softrod_step");
    SR_ON_DEVICE(h);
    return launch_step(h, actions, obs, reward, terminated, truncated, aux, h->cfg.n_substeps, 1, 0,
                       (hipStream_t)stream);
}

int softrod_step_packed(softrod_handle* h, const float* actions, float* packed, double* aux, void* stream) {
    if (!h || !actions || !packed) return fail(h, SOFTROD_EINVAL, "null argument");
    if (h->cfg.env_kind == SOFTROD_ENV_NONE)
        return fail(h, SOFTROD_EINVAL, "env_kind NONE has no step epilogue; use softrod_substeps");
    if ((h->cfg.features & SOFTROD_FEAT_REST_KAPPA_ACTION) && !h->basis_set)
        return fail(h, SOFTROD_EINVAL, "softrod_set_action_basis must be called before softrod_step");
    if ((h->cfg.features & SOFTROD_FEAT_SPLINE_MUSCLE_TORQUES) && !h->spline_set)
        return fail(h, SOFTROD_EINVAL, "softrod_set_spline_table must be called before softrod_step");
    if ((h->cfg.features & SOFTROD_FEAT_COOMM_MUSCLES) && !h->muscles_set)
        return fail(h, SOFTROD_EINVAL, "softrod_set_muscle_layers must be called before softrod_step");
    SR_ON_DEVICE(h);
    return launch_step(h, actions, packed, nullptr, nullptr, nullptr, aux, h->cfg.n_substeps, 1, 1,
                       (hipStream_t)stream);
}

namespace {
struct PeerTable { float* p[SOFTROD_MAX_PEERS]; };
// One thread per 32-bit word of the local rows; every peer's copy of the row block is written from it
// with SYSTEM-scope write-through stores (sc0 sc1: nothing of them stays behind in this GPU's L2).
// tag_word >= 0: when every block's stores have been acknowledged, the LAST block to finish stores
// `tag` into word `tag_word` of every peer buffer (system-scope release) — a reader that finds the tag
// finds the rows (the buffers are uncached / fine-grained on the reader's side, softrod_exchange_alloc).
__global__ void __launch_bounds__(256) softrod_scatter_rows_kernel(const float* __restrict__ packed, PeerTable peers,
                                                                   int n_peers, size_t n_words, size_t offset,
                                                                   long long tag_word, unsigned tag,
                                                                   unsigned* __restrict__ ticket) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_words) {
        const float v = packed[i];
        for (int q = 0; q < n_peers; ++q)
            __hip_atomic_store(&peers.p[q][offset + i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    if (tag_word < 0) return;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // this wave's stores have been acknowledged
    __syncthreads();
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");      // system scope
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const unsigned arrived = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (arrived == gridDim.x - 1) {
            __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // next launch (stream-ordered)
            for (int q = 0; q < n_peers; ++q)
                __hip_atomic_store(reinterpret_cast<unsigned*>(peers.p[q]) + tag_word, tag, __ATOMIC_RELEASE,
                                   __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
}  // namespace

int softrod_scatter_rows(softrod_handle* h, const float* packed, const uint64_t* peer_buffers, int n_peers,
                         int row_words, int64_t first_row, int64_t tag_word, uint32_t tag, void* stream) {
    if (!h || !packed || !peer_buffers) return fail(h, SOFTROD_EINVAL, "null argument");
    if (n_peers < 1 || n_peers > SOFTROD_MAX_PEERS || row_words < 1 || first_row < 0)
        return fail(h, SOFTROD_EINVAL, "need 1 <= n_peers <= SOFTROD_MAX_PEERS, row_words >= 1, first_row >= 0");
    SR_ON_DEVICE(h);
    PeerTable t{};
    for (int q = 0; q < n_peers; ++q) {
        if (!peer_buffers[q]) return fail(h, SOFTROD_EINVAL, "null peer buffer");
        t.p[q] = reinterpret_cast<float*>(static_cast<uintptr_t>(peer_buffers[q]));
    }
    // h->d_ticket was allocated and zeroed in softrod_create (a lazy hipMemset on the null stream is not
    // ordered before a launch on a non-blocking stream); ONE ticket per handle: tagged scatters of one
    // handle must be stream-ordered with each other (include/softrod.h)
    const size_t n_words = (size_t)h->cfg.n_envs * (size_t)row_words;
    const unsigned blocks = (unsigned)((n_words + 255) / 256);
    hipLaunchKernelGGL(softrod_scatter_rows_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, packed, t,
                       n_peers, n_words, (size_t)first_row * (size_t)row_words, (long long)tag_word, (unsigned)tag,
                       h->d_ticket);
    SR_HIP(h, hipGetLastError());
    return SOFTROD_OK;
}

// -- exchange buffers of transport "p2p": memory the OTHER GPUs store into -----------------------
// A GPU's L2 does not snoop a peer's stores into its HBM: rows a peer has written are only certain to
// be what a local kernel reads if the lines cannot sit in the local L2 — so these buffers are UNCACHED
// (hipDeviceMallocUncached; fine-grained where that is refused), not ordinary coarse-grained
// allocations, which are coherent across devices at kernel boundaries of the WRITER only.
int softrod_exchange_alloc(int device, uint64_t bytes, void** dev_ptr, uint8_t* ipc_handle, int* memory_kind) {
    if (!dev_ptr || bytes == 0) return fail(nullptr, SOFTROD_EINVAL, "null argument");
    DeviceGuard guard_(device);
    if (guard_.err != hipSuccess) return fail(nullptr, SOFTROD_EHIP, std::string("hipSetDevice: ") + hipGetErrorString(guard_.err));
    void* p = nullptr;
    int kind = SOFTROD_EXCHANGE_UNCACHED;
    hipError_t e = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocUncached);
    if (e != hipSuccess) {
        (void)hipGetLastError();
        kind = SOFTROD_EXCHANGE_FINEGRAINED;
        e = hipExtMallocWithFlags(&p, (size_t)bytes, hipDeviceMallocFinegrained);
    }
    if (e != hipSuccess) return fail(nullptr, SOFTROD_EHIP, std::string("hipExtMallocWithFlags: ") + hipGetErrorString(e));
    e = hipMemset(p, 0, (size_t)bytes);
    if (e == hipSuccess) e = hipDeviceSynchronize();
    if (e == hipSuccess && ipc_handle) {
        hipIpcMemHandle_t hd;
        static_assert(sizeof(hd) == SOFTROD_IPC_HANDLE_BYTES, "hipIpcMemHandle_t size");
        e = hipIpcGetMemHandle(&hd, p);
        if (e == hipSuccess) std::memcpy(ipc_handle, &hd, sizeof(hd));
    }
    if (e != hipSuccess) {
        (void)hipFree(p);
        return fail(nullptr, SOFTROD_EHIP, std::string("exchange buffer set-up: ") + hipGetErrorString(e));
    }
    *dev_ptr = p;
    if (memory_kind) *memory_kind = kind;
    return SOFTROD_OK;
}

int softrod_exchange_open(int device, const uint8_t* ipc_handle, int owner_device, void** dev_ptr) {
    if (!ipc_handle || !dev_ptr) return fail(nullptr, SOFTROD_EINVAL, "null argument");
    DeviceGuard guard_(device);
    if (guard_.err != hipSuccess) return fail(nullptr, SOFTROD_EHIP, std::string("hipSetDevice: ") + hipGetErrorString(guard_.err));
    if (owner_device >= 0 && owner_device != device) {
        int can = 0;
        hipError_t e = hipDeviceCanAccessPeer(&can, device, owner_device);
        if (e != hipSuccess || !can)
            return fail(nullptr, SOFTROD_EHIP, "device " + std::to_string(device) + " cannot access peer " + std::to_string(owner_device));
        e = hipDeviceEnablePeerAccess(owner_device, 0);
        if (e != hipSuccess && e != hipErrorPeerAccessAlreadyEnabled)
            return fail(nullptr, SOFTROD_EHIP, std::string("hipDeviceEnablePeerAccess: ") + hipGetErrorString(e));
        (void)hipGetLastError();
    }
    hipIpcMemHandle_t hd;
    std::memcpy(&hd, ipc_handle, sizeof(hd));
    void* p = nullptr;
    const hipError_t e = hipIpcOpenMemHandle(&p, hd, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess) return fail(nullptr, SOFTROD_EHIP, std::string("hipIpcOpenMemHandle: ") + hipGetErrorString(e));
    *dev_ptr = p;
    return SOFTROD_OK;
}

int softrod_exchange_close(int device, void* dev_ptr) {
    if (!dev_ptr) return SOFTROD_OK;
    DeviceGuard guard_(device);
    const hipError_t e = hipIpcCloseMemHandle(dev_ptr);
    return e == hipSuccess ? SOFTROD_OK : fail(nullptr, SOFTROD_EHIP, std::string("hipIpcCloseMemHandle: ") + hipGetErrorString(e));
}

int softrod_exchange_free(int device, void* dev_ptr) {
    if (!dev_ptr) return SOFTROD_OK;
    DeviceGuard guard_(device);
    (void)hipDeviceSynchronize();
    const hipError_t e = hipFree(dev_ptr);
    return e == hipSuccess ? SOFTROD_OK : fail(nullptr, SOFTROD_EHIP, std::string("hipFree: ") + hipGetErrorString(e));
}

int softrod_substeps(softrod_handle* h, const float* actions, int n, void* stream) {
    if (!h || n < 0) return fail(h, SOFTROD_EINVAL, "bad argument");
    if (actions && h->cfg.env_kind != SOFTROD_ENV_SOFTPENDULUM && h->cfg.env_kind != SOFTROD_ENV_NONE)
        return fail(h, SOFTROD_EINVAL, "softrod_substeps takes no actions for this env_kind");
    if ((h->cfg.features & SOFTROD_FEAT_COOMM_MUSCLES) && !h->muscles_set)
        return fail(h, SOFTROD_EINVAL, "softrod_set_muscle_layers must be called before softrod_substeps");
    SR_ON_DEVICE(h);
    return launch_step(h, actions, nullptr, nullptr, nullptr, nullptr, nullptr, n, 0, 0, (hipStream_t)stream);
}

int softrod_observe(softrod_handle* h, const float* prev_action, float* obs, void* stream) {
    if (!h || !obs) return fail(h, SOFTROD_EINVAL, "null argument");
    if (h->cfg.env_kind == SOFTROD_ENV_NONE) return fail(h, SOFTROD_EINVAL, "env_kind NONE has no observation");
    SR_ON_DEVICE(h);
    const dim3 grid((unsigned)h->cfg.n_envs), block(kLanes);
    if (is_mocto(h)) {
        // get_state on the resident state (prev_action: the resident copy); like the reference's it moves ArmTwoEnv's _prev_kappa
        if (prev_action) return fail(h, SOFTROD_EINVAL, "the muscle octopus envs observe with their resident prev_action (pass NULL)");
        hipLaunchKernelGGL(softrod_mocto_epilogue_kernel, grid, dim3(kLanes * h->nw), 0, (hipStream_t)stream, h->P, h->S, obs,
                           nullptr, nullptr, nullptr, 0, 0);
    } else if (is_octo(h) && !is_pull(h))
        hipLaunchKernelGGL(softrod_octo_observe_kernel, grid, dim3(kLanes * h->nw), 0, (hipStream_t)stream,
                           h->P, h->S, prev_action, obs);
    else if (h->epl == 2)
        hipLaunchKernelGGL(softrod_observe_kernel<2>, grid, block, 0, (hipStream_t)stream, h->P, h->S,
                           prev_action, obs);
    else
        hipLaunchKernelGGL(softrod_observe_kernel<1>, grid, block, 0, (hipStream_t)stream, h->P, h->S,
                           prev_action, obs);
    SR_HIP(h, hipGetLastError());
    return SOFTROD_OK;
}

int softrod_state_view_get(softrod_handle* h, softrod_state_view* out) {
    if (!h || !out) return fail(h, SOFTROD_EINVAL, "null argument");
    out->n_envs = h->cfg.n_envs;
    out->n_elem = h->cfg.n_elem;
    out->lane_stride = kLanes * h->epl * h->nw;
    out->arm_stride = (is_flat(h) || is_mocto(h)) ? h->P.seg : 0;
    out->position = h->S.pos;
    out->velocity = h->S.vel;
    out->director = h->S.dir;
    out->omega = h->S.omg;
    out->tangents = h->S.tan;
    out->time = h->S.time;
    out->control = h->S.ctrl;
    out->kappa = h->S.kap;
    out->rest_kappa = h->S.rkap;
    out->env_memory = h->S.envmem;
    out->prev_action = h->S.prev_action;
    out->head = h->S.head;
    out->bc_targets = h->S.bc;
    out->sucker_ratio = h->S.sucker;
    out->muscle_activation = h->d_mact;
    out->sucker_index = h->d_sucker_idx;
    out->material = h->d_mat;
    out->env_aux = h->d_aux;
    out->prev_kappa = h->d_prev_kappa;
    return SOFTROD_OK;
}

int softrod_set_timing(softrod_handle* h, int n_launches) {
    if (!h || n_launches < 0 || n_launches > (1 << 20)) return fail(h, SOFTROD_EINVAL, "bad argument");
    SR_ON_DEVICE(h);
    while ((int)h->ev_start.size() < n_launches) {
        hipEvent_t a = nullptr, b = nullptr;
        SR_HIP(h, hipEventCreate(&a));
        SR_HIP(h, hipEventCreate(&b));
        h->ev_start.push_back(a);
        h->ev_stop.push_back(b);
    }
    while ((int)h->ev_start.size() > n_launches) {
        (void)hipEventDestroy(h->ev_start.back());
        (void)hipEventDestroy(h->ev_stop.back());
        h->ev_start.pop_back();
        h->ev_stop.pop_back();
    }
    h->timed = 0;
    return SOFTROD_OK;
}

int softrod_kernel_times_ms(softrod_handle* h, float* out_ms, int cap, int* count) {
    if (!h || !out_ms || !count || cap < 0) return fail(h, SOFTROD_EINVAL, "bad argument");
    SR_ON_DEVICE(h);
    const int n = h->timed < cap ? h->timed : cap;
    for (int i = 0; i < n; ++i) {
        SR_HIP(h, hipEventSynchronize(h->ev_stop[i]));
        SR_HIP(h, hipEventElapsedTime(out_ms + i, h->ev_start[i], h->ev_stop[i]));
    }
    *count = n;
    return SOFTROD_OK;
}

int softrod_last_kernel_ms(softrod_handle* h, float* ms) {
    if (!h || !ms) return fail(h, SOFTROD_EINVAL, "null argument");
    if (h->timed < 1) return fail(h, SOFTROD_EINVAL, "timing not enabled or no timed launch yet");
    SR_ON_DEVICE(h);
    SR_HIP(h, hipEventSynchronize(h->ev_stop[h->timed - 1]));
    SR_HIP(h, hipEventElapsedTime(ms, h->ev_start[h->timed - 1], h->ev_stop[h->timed - 1]));
    return SOFTROD_OK;
}

const char* softrod_last_error(softrod_handle* h) { return h ? h->err.c_str() : g_err.c_str(); }

// Mirrors launch_step's choice (the env.step form: epilogue = 1).
const char* softrod_kernel_tier(softrod_handle* h) {
    if (!h) return "";
    const bool zup = (h->P.features & kFeatPlaneZup) != 0;
    const unsigned f = h->cfg.features;
    const int e = h->cfg.env_kind;
    std::string t;
    if (is_mocto(h)) {
        t = std::string("softrod_mocto_action_kernel | softrod_octo_step_kernel<muscle arms,") + std::to_string(h->nw) +
            (h->nw == 1 ? " wave" : " waves") + ",1 env/wg,taper> | softrod_mocto_epilogue_kernel";
    } else if (is_octo(h) && is_pull(h)) {
        t = "softrod_octo_step_kernel<ArmPullWeight,1 wave,1 env/wg,taper>";
    } else if (is_octo(h)) {
        if (zup && h->nw == 2 && h->octo_one_wave && h->P.n_arm * h->P.seg == 2 * kLanes && !(h->P.seg & 1))
            t = "softrod_octo1w_step_kernel<zup,1 wave,1 env/wg>";
        else if (zup && h->nw == 2 && !h->octo_one_env_per_block)
            t = "softrod_octo_step_kernel<zup,2 waves,4 envs/wg>";
        else
            t = std::string("softrod_octo_step_kernel<") + (zup ? "zup," : "general plane,") +
                (h->nw <= 2 ? "2" : "8") + " waves max,1 env/wg>";
    } else if (h->window_refresh > 0) {
        t = std::string("softrod_step_window_kernel<ArmSingle,") + (h->window_paired ? "4 rods/wg" : "1 rod/wg,s_barrier") +
            "> refresh=" + std::to_string(h->window_refresh) + " + softrod_step_fast_kernel<ArmSingle,epl=2> epilogue";
    } else if (h->cfg.math_mode == SOFTROD_MATH_FAST) {
        const char* spec = "runtime mask";
        if (h->tapered) {
            if (f == SOFTROD_FEATURES_ARM_SINGLE && e == SOFTROD_ENV_ARM_SINGLE && zup) spec = "ArmSingle";
            else if (f == kFeaturesTaperedSuckerArm && e == SOFTROD_ENV_NONE) spec = "damped sucker arm";
            else if (f == SOFTROD_FEATURES_ARM_PUSH && e == SOFTROD_ENV_ARM_PUSH) spec = "ArmPush";
        } else if (f == SOFTROD_FEATURES_SOFTPENDULUM && e == SOFTROD_ENV_SOFTPENDULUM) spec = "SoftPendulum";
        else if (f == SOFTROD_FEATURES_SOFTPENDULUM3D && e == SOFTROD_ENV_SOFTPENDULUM3D) spec = "SoftPendulum3D";
        else if (f == SOFTROD_FEATURES_ARM_SINGLE && e == SOFTROD_ENV_ARM_SINGLE && zup) spec = "ArmSingle";
        else if (f == SOFTROD_FEATURES_SOFT_ARM && e == SOFTROD_ENV_SOFT_ARM) spec = "SoftArm";
        else if (f == kFeaturesMuscleRod && e == SOFTROD_ENV_NONE) spec = "muscle rod";
        t = std::string("softrod_step_fast_kernel<") + spec + ",epl=" + std::to_string(h->tapered ? 1 : h->epl) +
            (h->tapered ? ",taper>" : ">");
    } else
        t = "softrod_step_libm_kernel";
    h->tier = t;
    return h->tier.c_str();
}

int softrod_destroy(softrod_handle* h) {
    if (!h) return SOFTROD_OK;
    DeviceGuard guard_(h->device);
    (void)hipDeviceSynchronize();
    autoreset_release(h);
    void* bufs[] = {h->S.pos, h->S.vel, h->S.dir, h->S.omg, h->S.tan, h->S.time, h->S.bc,
                    h->S.ctrl, h->S.kap, h->S.rkap, h->S.envmem, h->S.prev_action, h->S.head, h->d_params, h->d_state, h->d_time_tab, h->d_mat, h->d_sucker, h->d_sucker_idx, h->d_aux, h->d_prev_kappa, h->d_mact, h->d_mtab, h->d_basis, h->d_spline, h->d_init, h->d_mask, h->d_ticket};
    for (void* p : bufs) (void)hipFree(p);
    if (h->h_init) (void)hipHostFree(h->h_init);
    if (h->h_mask) (void)hipHostFree(h->h_mask);
    for (hipEvent_t e : h->ev_start) (void)hipEventDestroy(e);
    for (hipEvent_t e : h->ev_stop) (void)hipEventDestroy(e);
    if (h->ev_reset) (void)hipEventDestroy(h->ev_reset);
    delete h;
    return SOFTROD_OK;
}

}  // extern "C"

#ifdef SOFTROD_PHASE_CLOCKS
// diagnostic build only: the phase stamps of the last step launch (tools/phase_clocks.py)
extern "C" int softrod_debug_phase_clocks(unsigned long long* out, int n_rods) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(softrod::g_phase_clock), (size_t)n_rods * 8 * sizeof(unsigned long long));
}
#endif
