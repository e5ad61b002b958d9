// softrod_octo.hpp — OctoFlat-v0: n_arm Cosserat rods joined to one rigid head, one env per
// workgroup (BASELINE.json configs[4]).
//
// What it restates, and from where (each mirrored by oracle/octoflat_oracle.inc.c):
//   build_octopus                      gym_softrobot/envs/octopus/build.py:52-217
//   FixedJoint2Rigid                   gym_softrobot/utils/custom_elastica/joint.py:20-225
//   BodyBoundaryCondition              gym_softrobot/utils/custom_elastica/constraint.py:8-85
//   intersection (arm crossing)        gym_softrobot/utils/intersection.py:12-77
//   FlatEnv get_state/set_action/step  gym_softrobot/envs/octopus/flat_env.py:231-408
//   Cylinder / RigidBodyBase           pyelastica==1.0.0 (recalled, not on disk)
//
// Mapping.  The arms of one env sit `seg` slots apart (seg = 16 for the reference's 10
// elements per arm) in a block of nw = ceil(n_arm*seg/64) wavefronts, ONE NODE PER LANE, so
// the whole rod machinery of softrod_fast.hpp (DPP stencils, fused damper, contact) runs
// unchanged: an arm never straddles a wavefront and the ghost slots between arms carry no
// stiffness, no mass coefficient and no contact.  The rigid head (x, v, Q, w: 18 doubles)
// is replicated in registers of every lane.  Per substep the only cross-arm traffic is the
// head's net joint load: three doubles per arm through LDS (double-buffered, ONE s_barrier
// per substep, consumed after the arms' contact and rate update so that the round trip is
// hidden), then every lane integrates the same head with the same operands in the same
// order, so the replicas stay bit-identical.
//
// State rows: the env's slots are rows [env*nw + wave][64] of the SoA arrays (the generic
// one-rod-per-wave layout with N*nw rows); arm a = slots a*seg .. a*seg+n_elem.
#pragma once

namespace softrod {

struct HeadState {
    double x[3], v[3], Q[9], w[3];
};

__device__ __forceinline__ void load_head(const StatePtrs& S, size_t N, int env, HeadState& H, double tgt[2]) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        H.x[i] = S.head[(size_t)i * N + env];
        H.v[i] = S.head[(size_t)(3 + i) * N + env];
        H.w[i] = S.head[(size_t)(15 + i) * N + env];
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) H.Q[i] = S.head[(size_t)(6 + i) * N + env];
    tgt[0] = S.head[(size_t)18 * N + env];
    tgt[1] = S.head[(size_t)19 * N + env];
}

__device__ __forceinline__ void store_head(const StatePtrs& S, size_t N, int env, const HeadState& H) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        S.head[(size_t)i * N + env] = H.x[i];
        S.head[(size_t)(3 + i) * N + env] = H.v[i];
        S.head[(size_t)(15 + i) * N + env] = H.w[i];
    }
#pragma unroll
    for (int i = 0; i < 9; ++i) S.head[(size_t)(6 + i) * N + env] = H.Q[i];
}

}  // namespace softrod
#include "softrod_mocto.hpp"
namespace softrod {

// Kinematic half/full step of the rigid body followed by
// BodyBoundaryCondition.compute_contrain_values (constraint.py:41-58): z held, d3 = e_z,
// d1 and d2 renormalised in the plane.  compute_constrain_rates (constraint.py:60-85) keeps
// v_z = 0 and omega = (0, 0, w_z) — established at kernel entry and after every rate update —
// so the step is a planar one: x, y advance, z stays, and the transposed Rodrigues matrix
// R(h w) reduces to a rotation about e_z acting on the x, y components of d1 and d2.
// d1 and d2 enter every step normalised to rounding (head_normalize at kernel entry) and a
// rotation keeps |d|^2 = 1 + O(1e-16), where 1/sqrt(n2) and its first Newton iterate
// (3 - n2)/2 agree to O(1e-32): one FMA instead of the quarter-rate v_rsq_f64 and its
// refinement, on the head's serial critical path.
__device__ __forceinline__ void head_kinematic(double h, HeadState& H) {
    H.x[0] = fma(h, H.v[0], H.x[0]);
    H.x[1] = fma(h, H.v[1], H.x[1]);
    const double a = h * H.w[2];
    const double t = a * a;
    double sc, cc;
    sinc_cosc(t, sc, cc);
    const double sn = sc * a, cs = fma(-cc, t, 1.0);
    const double q00 = fma(sn, H.Q[3], cs * H.Q[0]), q01 = fma(sn, H.Q[4], cs * H.Q[1]);
    const double q10 = fma(-sn, H.Q[0], cs * H.Q[3]), q11 = fma(-sn, H.Q[1], cs * H.Q[4]);
    const double i0 = fma(-0.5, fma(q00, q00, q01 * q01), 1.5), i1 = fma(-0.5, fma(q10, q10, q11 * q11), 1.5);
    H.Q[0] = q00 * i0; H.Q[1] = q01 * i0; H.Q[2] = 0.0;
    H.Q[3] = q10 * i1; H.Q[4] = q11 * i1; H.Q[5] = 0.0;
    H.Q[6] = 0.0; H.Q[7] = 0.0; H.Q[8] = 1.0;
}
// BodyBoundaryCondition.compute_contrain_values on whatever the state rows hold (kernel entry)
__device__ __forceinline__ void head_normalize(HeadState& H) {
    const double i0 = fast_rsqrt(fma(H.Q[0], H.Q[0], H.Q[1] * H.Q[1]));
    const double i1 = fast_rsqrt(fma(H.Q[3], H.Q[3], H.Q[4] * H.Q[4]));
    H.Q[0] *= i0; H.Q[1] *= i0; H.Q[2] = 0.0;
    H.Q[3] *= i1; H.Q[4] *= i1; H.Q[5] = 0.0;
    H.Q[6] = 0.0; H.Q[7] = 0.0; H.Q[8] = 1.0;
}

// np.linalg.solve on the 4x4 system of utils/intersection.py:60-66 — LU with partial
// pivoting, fully unrolled so that every index is a compile-time constant (registers, no
// scratch).  Returns false for a singular matrix (numpy raises LinAlgError; the reference
// would crash there, the oracle skips the pair).
__device__ __forceinline__ bool solve4_t01(double (&A)[4][4], double (&b)[4], double& t0, double& t1) {
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        int m = k;
        double best = fabs(A[k][k]);
#pragma unroll
        for (int i = k + 1; i < 4; ++i) {
            const double v = fabs(A[i][k]);
            if (v > best) { best = v; m = i; }
        }
        if (best == 0.0) return false;
#pragma unroll
        for (int i = k + 1; i < 4; ++i) {
            if (m == i) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { const double t = A[k][j]; A[k][j] = A[i][j]; A[i][j] = t; }
                const double t = b[k]; b[k] = b[i]; b[i] = t;
            }
        }
#pragma unroll
        for (int i = k + 1; i < 4; ++i) {
            const double f = A[i][k] / A[k][k];
#pragma unroll
            for (int j = k; j < 4; ++j) A[i][j] -= f * A[k][j];
            b[i] -= f * b[k];
        }
    }
    double x[4];
#pragma unroll
    for (int i = 3; i >= 0; --i) {
        double s = b[i];
#pragma unroll
        for (int j = i + 1; j < 4; ++j) s -= A[i][j] * x[j];
        x[i] = s / A[i][i];
    }
    t0 = x[0]; t1 = x[1];
    return true;
}

// FlatEnv.get_state (centralized), flat_env.py:231-286.
//   individual[arm] = [kappa0 (n-1) | x - cx (n+1) | y - cy (n+1) | vx (n+1) | vy (n+1) | prev_action (nk)]
//   shared          = [target - head_xy | head v_xy | head directors (9)]
__device__ __forceinline__ void octo_write_obs(const RodParams& P, int tid, const LaneN<1>& L,
                                               const HeadState& H, const double tgt[2],
                                               const float* __restrict__ prev_action,
                                               float* __restrict__ o) {
    const int n = P.n_elem, nk = P.n_action;
    const int width = (n - 1) + 4 * (n + 1) + nk;
    const int r = tid & (P.seg - 1), arm = tid >> P.seg_shift;
    if (arm < P.n_arm) {
        float* row = o + (size_t)arm * width;
        if (r < n - 1) row[r] = (float)L.kap[0][0];
        if (r <= n) {
            row[(n - 1) + r] = (float)(L.x[0][0] - H.x[0]);
            row[(n - 1) + (n + 1) + r] = (float)(L.x[0][1] - H.x[1]);
            row[(n - 1) + 2 * (n + 1) + r] = (float)L.v[0][0];
            row[(n - 1) + 3 * (n + 1) + r] = (float)L.v[0][1];
        }
        if (r < nk) row[(n - 1) + 4 * (n + 1) + r] = prev_action[arm * nk + r];
    }
    if (tid == 0) {
        float* sh = o + (size_t)P.n_arm * width;
        sh[0] = (float)(tgt[0] - H.x[0]);
        sh[1] = (float)(tgt[1] - H.x[1]);
        sh[2] = (float)H.v[0];
        sh[3] = (float)H.v[1];
#pragma unroll
        for (int i = 0; i < 9; ++i) sh[4 + i] = (float)H.Q[i];
    }
}

__device__ __forceinline__ int octo_obs_dim(const RodParams& P) {
    return P.n_arm * ((P.n_elem - 1) + 4 * (P.n_elem + 1) + P.n_action) + 13;
}

// ---------------------------------------------------------------------------------
// One env.step (or n_sub bare substeps) of every env of the shard.
// grid = n_envs, block = 64*nw threads.  MAXW bounds nw for the register allocator.
// ---------------------------------------------------------------------------------
#ifndef SOFTROD_OCTO_WAVES
#define SOFTROD_OCTO_WAVES 2
#endif
// EPB = envs per workgroup.  EPB = 1: one env per workgroup of nw waves, the waves meet at
// s_barrier once per substep.  EPB = 4 (the reference shape, nw = 2): FOUR envs per workgroup of
// eight waves, env e on waves e and e + 4.  The hardware places waves w and w + 4 of a workgroup
// on the SAME SIMD (probed with s_getreg HW_ID on gfx950: 4096 of 4096 pairs; partners of a
// two-wave workgroup always sit on different SIMDs), and a workgroup of 8 x 256 VGPRs is exactly
// one CU.  The two waves of an env then share one SIMD's issue slots: while one waits for the
// other's joint loads the other is the one running, so the wait costs no SIMD time, and the
// rendezvous becomes a flag in LDS (release / acquire at workgroup scope) instead of a
// workgroup-wide s_barrier that would couple four unrelated envs.
// SOFTROD_OCTO_PRIO: a wave's issue priority until it has posted its joint loads of the substep; 0
// after.  The partner that has NOT posted is the one the SIMD serves first, so the wave that runs
// ahead (the arbiter favours the older wave) finds the partner's flag set when it gets to the
// rendezvous instead of sleeping on it: 11.72 -> 11.57 ms (profiles/README.md, r2k).
#ifndef SOFTROD_OCTO_PRIO
#define SOFTROD_OCTO_PRIO 1
#endif
// The muscle-arm instantiations (taper table, three muscle layers, suckers on top of the rod, joint and head code) want ~330
// registers: at two waves per SIMD (256) they park 217 of them in scratch and the loop waits on it — 12 GB of HBM traffic per
// env.step of 4096 OctoArmPullWeight envs, VALU busy 0.37.  ONE wave per SIMD (512 registers, no spills in the loop) is
// faster in spite of half the occupancy: OctoArmPullWeight 13.7 -> 10.6 ms, OctoCrawl 12.6 -> 9.9, OctoArmTwo 5.18 -> 2.96,
// OctoReach 21.4 -> 12.3 (profiles/README.md "Round 6").  OctoFlat's own instantiations are indifferent (8.94 / 8.97 ms) and stay.
#ifndef SOFTROD_MUSCLE_OCTO_WAVES
#define SOFTROD_MUSCLE_OCTO_WAVES 1
#endif
template <unsigned F>
constexpr int octo_kernel_waves() { return kMusclesCompiled<F> ? SOFTROD_MUSCLE_OCTO_WAVES : SOFTROD_OCTO_WAVES; }

template <unsigned F, int MAXW, int EPB = 1>
__global__ void __launch_bounds__(kLanes * MAXW * EPB, (octo_kernel_waves<F>()))
softrod_octo_step_kernel(const RodParams P, const StatePtrs S, const float* __restrict__ actions,
                         float* __restrict__ obs, double* __restrict__ reward,
                         uint8_t* __restrict__ terminated, uint8_t* __restrict__ truncated,
                         const int n_sub, const int epilogue, const int pack) {
    static_assert(EPB == 1 || MAXW == 2, "several envs per workgroup: the two-wave shape only");
    __shared__ double xch_[EPB][2][MAXW][4];  // [env][buffer][wave][Fx, Fy, Tz of the wave's arms, pad]
    __shared__ double sxy_[EPB][kLanes * MAXW][2];
    __shared__ int scount_[EPB];
    __shared__ int flag_[EPB][2][MAXW];       // EPB > 1: substeps posted by each wave, per buffer
    __shared__ int sany_[2][EPB];             // EPB > 1: per-env "any" of the two collectives

    const int wib = threadIdx.x >> 6;         // wave in the block
    const int es = (EPB == 1) ? 0 : (wib & (EPB - 1));           // env slot in the block
    const int wave = (EPB == 1) ? wib : (wib / EPB);             // wave of the env
    const int lane = threadIdx.x & 63;
    const int tid = wave * kLanes + lane;     // thread of the env
    const int nw = (EPB == 1) ? (int)(blockDim.x >> 6) : MAXW;
    const int nthr = nw * kLanes;
    const bool active = (EPB == 1) || ((int)blockIdx.x * EPB + es < P.n_envs);
    const int env = active ? (int)blockIdx.x * EPB + es : P.n_envs - 1;   // idle slots shadow a real env, read-only
    auto& xch = xch_[es];
    auto& sxy = sxy_[es];
    int& scount = scount_[es];
    const size_t N = (size_t)P.n_envs, NR = N * (size_t)nw;
    const int row = env * nw + wave;
    const int n = P.n_elem, nk = P.n_action;
    const int r = tid & (P.seg - 1), arm = tid >> P.seg_shift;
    const bool arm_ok = arm < P.n_arm;
    if constexpr (SOFTROD_OCTO_CONTACT_LDS && (F & kFeatPlaneZup) != 0) stage_contact_params(P);   // (barriers follow)
    if (tid < 2 * MAXW * 4) (&xch[0][0][0])[tid] = 0.0;   // rows of absent waves read as zero loads
    if constexpr (EPB == 1) __syncthreads();     // the staged tables are there before any wave reads them (EPB > 1: below)
    if (EPB > 1 && tid < 2 * MAXW) (&flag_[es][0][0])[tid] = 0;
    // per-env "any thread of the env": the workgroup barrier when the env IS the workgroup; with
    // several envs per workgroup every wave takes the same barriers and the answer is per slot
    auto env_any = [&](bool pred, int which) -> bool {
        if constexpr (EPB == 1) return __syncthreads_or(pred ? 1 : 0) != 0;
        else {
            if (tid == 0) sany_[which][es] = 0;
            __syncthreads();
            if (__any(pred) && lane == 0) atomicOr(&sany_[which][es], 1);
            __syncthreads();
            return sany_[which][es] != 0;
        }
    };
    bool live = active;
    // the muscle octopus envs (softrod_mocto.hpp): this kernel steps the body only; set_action has run, get_state / reward follow
    const bool mocto = kMusclesCompiled<F> && is_mocto_env(P.env_kind);
    if (epilogue && S.skip && S.skip[env] && active) {   // reset by the auto-reset pass of this env.step
        if constexpr (EPB == 1) {
            __syncthreads();                   // every thread has read the flag
            if (tid == 0 && !mocto) S.skip[env] = 0;       // (mocto: the epilogue kernel clears it)
            return;
        }
        live = false;
    }
    if constexpr (EPB > 1) {
        __syncthreads();                       // flags cleared; every thread has read its skip flag
        if (!live && active && tid == 0) S.skip[env] = 0;
    }

    LaneN<1> L;
    load_lane<1, F>(S, NR, row, lane, L);
    sanitize_unused_slots<1>(P, tid, L);        // ghost slots between the arms (softrod_kernels.hpp)
    HeadState H;
    double tgt[2];
    load_head(S, N, env, H, tgt);
    H.v[2] = 0.0; H.w[0] = 0.0; H.w[1] = 0.0;   // compute_constrain_rates holds these at zero
    const double before[2] = {H.x[0], H.x[1]};

    // set_action (flat_env.py:288-311): rest_kappa[0,:] = zero-padded cubic interp1d of the
    // arm's knots = basis @ knots
    if (has<F>(P, SOFTROD_FEAT_REST_KAPPA_ACTION) && actions && live) {
        double rk0 = 0.0;
        if (arm_ok && r < n - 1) {
            const float* a = actions + (size_t)env * (P.n_arm * nk) + arm * nk;
            for (int j = 0; j < nk; ++j) rk0 += S.basis[r * nk + j] * (double)a[j];
        }
        L.rk[0][0] = rk0;
        S.rkap[(size_t)row * kLanes + lane] = rk0;
    }

    // joint frame of this arm: z_rotation(head d2, 360/n_arm * arm degrees), joint.py:66-71
    const bool base = arm_ok && r == 0;
    // FixedJoint2Rigid(angle = ...): 360 / n_arm * arm_i in FlatEnv (octopus/build.py:73-74,117-132), 0 for the weight
    // of ArmPullWeightEnv (arm_push_env.py:585-587); the host's product, so that it is the reference's float64
    const double ang = (P.joint_angle0 + P.joint_angle_step * (double)arm) / 180.0 * M_PI;
    const double ct = cos(ang), st = sin(ang);

    // The muscle arm joined to a rigid body (ArmPullWeightEnv): tapered, COOMM layers, a sucker; ArmPushEnv's
    // set_action / step / get_state (softrod_kernels.hpp).  One wave per env (n_arm * seg <= 64).
    constexpr bool kMuscleArm = kMusclesCompiled<F>;
    EnvAction A;
#pragma unroll
    for (int i = 0; i < 8; ++i) A.a[i] = 0.0f;
    A.force = 0.0;
    A.mu_set = false;
    BcTargets B;
#pragma unroll
    for (int i = 0; i < 3; ++i) { B.pos[i] = 0.0; B.vel[i] = 0.0; }
#pragma unroll
    for (int i = 0; i < 9; ++i) B.Q[i] = 0.0;
#pragma unroll
    for (int j = 0; j < SOFTROD_MAX_SUCKERS; ++j) { B.keep[j] = 1.0; B.snode[j] = -1; B.selem[j] = -1; }
    if constexpr (kMuscleArm) {
        if (mocto) {
            // every arm has its own SuckerControllers: rows [j][env * n_arm + arm]; the targets as slots of the env's block
            const size_t NA = N * (size_t)P.n_arm, ai = (size_t)env * P.n_arm + (arm_ok ? arm : 0);
#pragma unroll
            for (int j = 0; j < SOFTROD_MAX_SUCKERS; ++j) {
                if (j < P.n_suckers && arm_ok) {
                    B.keep[j] = 1.0 - S.sucker[(size_t)j * NA + ai];
                    sucker_targets(P, S.sucker_idx[(size_t)j * NA + ai], B.snode[j], B.selem[j]);
                    B.snode[j] += arm * P.seg;
                    B.selem[j] += arm * P.seg;
                }
            }
        } else {
            load_suckers<F>(P, S, N, env, B);
            if (live) {
                set_action_n<F, kRuntimeEnv, 1>(P, S, NR, row, lane, actions, A, B, L);
                if (epilogue) push_store_prev_com<F, kRuntimeEnv, 1>(P, S, NR, row, lane, L, n_sub);
            }
        }
    }
    ConstN<1> C;
    build_const<F, 1, kMuscleArm>(P, tid, A, C, S.mat);
    if constexpr (kMuscleArm) build_muscle_const<F, 1, true>(P, S, NR, row, lane, A, C);
    RodParams Pk = P;
    if (!has<F>(P, SOFTROD_FEAT_ANALYTICAL_DAMPER)) Pk.damp_t = 1.0;
    const double head_inv_mass = 1.0 / P.head_mass;

    double time = S.time[env];
    int parity = 0;
    int posted = 0;          // EPB > 1: substeps whose joint loads this wave has posted per buffer pair

    // SOFTROD_OCTO_DIAG (diagnostic builds of tools/octo_budget.sh / octo_ab.sh ONLY; results are wrong with any
    // bit set): 1 = no joint evaluation (zero loads are posted), 2 = no head step, 4 = no exchange (no rendezvous,
    // no LDS reads), 8 = no in-wave reduction (lane 0 posts its own arm's load).  The instruction budget of the
    // substep is the difference of their hot-path counts (tools/hot_path_isa.py), the cost of a block in time the
    // difference of their timings (profiles/README.md r5).
#ifndef SOFTROD_OCTO_DIAG
#define SOFTROD_OCTO_DIAG 0
#endif
// SOFTROD_OCTO_BASE_MASK (A/B switch, default 0 = off; bit 1: the joint evaluation, bit 2: the head's step run
// with EXEC restricted to the base lanes, 4 of 64 — only they consume the head's state inside the loop; the head
// is broadcast from lane 0 after the loop).  The idea: the instruction count is unchanged (a wave instruction
// costs its issue slot whatever its mask) but masked lanes do not switch, and this kernel runs against the
// board's power limit.  MEASURED AND NOT ADOPTED (round 5, profiles/README.md "OctoFlat budget"): 9.26-9.29 ms
// against 9.31-9.34 ms per launch on one box (-0.5 %, inside the box-to-box spread), and with BOTH bits set the
// one-arm shape (OctoFlatLite-v0, one wave per env) aborts on the GPU while either bit alone passes - a
// code-generation hazard nobody needs for half a per cent.
#ifndef SOFTROD_OCTO_BASE_MASK
#define SOFTROD_OCTO_BASE_MASK 0
#endif
    auto joints = [&](double (&f)[1][3], double (&tq)[1][3], const LaneN<1>& Lc, const double (&xn)[1][3]) {
        double part[3] = {0.0, 0.0, 0.0};
        if ((SOFTROD_OCTO_DIAG & 1) == 0 && (!(SOFTROD_OCTO_BASE_MASK & 1) || base)) {
        // FixedJoint2Rigid.apply_forces (joint.py:48-123): spring + normal damping between the
        // arm's node 0 and the point head_radius along the arm's direction from the head axis.
        // The head's d2 lies in the plane (constrain_values), so the direction has no z part.
        const double b0 = H.Q[3], b1 = H.Q[4];
        const double dir[2] = {-(ct * b0 - st * b1), -(st * b0 + ct * b1)};
        const double pos[3] = {fma(dir[0], P.head_radius, H.x[0]), fma(dir[1], P.head_radius, H.x[1]), 0.0};
        double dv[3], d2 = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) { dv[i] = Lc.x[0][i] - pos[i]; d2 = fma(dv[i], dv[i], d2); }
        const bool apart = d2 > (2.220446049250313e-12 * 2.220446049250313e-12);
        // (the seed on a clamped argument and a select after it: as `apart ? rsqrt : 0` the compiler
        // branches around the refinement, in every substep, for a case that never happens)
        const double ir = fast_rsqrt(fmax(d2, 1.0e-300));
        const double idist = apart ? ir : 0.0;
        double nv[3], rel = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) { nv[i] = dv[i] * idist; rel = fma(Lc.v[0][i] - H.v[i], nv[i], rel); }
        const double damp = P.joint_nu * rel;
        double fj[3], link[3], force[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            fj[i] = fma(P.joint_k, dv[i], -damp * nv[i]);
            link[i] = xn[0][i] - Lc.x[0][i];
        }
        // apply_torques (joint.py:125-219): node 1 pulled towards its rest place on the ray
        force[0] = -P.joint_kt * (xn[0][0] - fma(P.rest_len, dir[0], pos[0]));
        force[1] = -P.joint_kt * (xn[0][1] - fma(P.rest_len, dir[1], pos[1]));
        force[2] = -P.joint_kt * xn[0][2];
        const double tj[3] = {link[1] * force[2] - link[2] * force[1], link[2] * force[0] - link[0] * force[2],
                              link[0] * force[1] - link[1] * force[0]};
        // (`base` as a factor 1 / 0 instead of a condition removes the divergent branches the compiler
        // wraps these selects in — 13 instructions fewer and 9 % SLOWER, 10.56 against 9.64 ms:
        // profiles/README.md r2k)
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            f[0][i] -= base ? fj[i] : 0.0;
            const double* Q = Lc.Q[0];
            tq[0][i] += base ? fma(Q[3 * i + 2], tj[2], fma(Q[3 * i + 1], tj[1], Q[3 * i] * tj[0])) : 0.0;
        }
        // Net load on the head.  Its rates are held to (vx, vy, 0), (0, 0, wz) and its d3 to e_z,
        // so only Fx, Fy and the lab-frame torque about z (= torque along d3) reach it.  The arms
        // of a wave are summed IN the wave: a xor-butterfly over the arm stride (ds_bpermute: the
        // LDS crossbar, no memory, no VALU) leaves the sum of the wave's base lanes in every base
        // lane, in a fixed order; lane 0 posts the wave's three numbers in LDS.  They are consumed
        // after the arms' contact and rate update (head_update below), which hides the round trip
        // and lets the waves of the env reach the barrier together.
        part[0] = base ? fj[0] : 0.0; part[1] = base ? fj[1] : 0.0; part[2] = base ? tj[2] : 0.0;
        }
#pragma unroll
        for (int off = 16; off < kLanes; off <<= 1) {
            if ((SOFTROD_OCTO_DIAG & 8) == 0 && off >= P.seg) {
#pragma unroll
                for (int i = 0; i < 3; ++i) part[i] += __shfl_xor(part[i], off);
            }
        }
        if (lane == 0) {
            xch[parity][wave][0] = part[0];
            xch[parity][wave][1] = part[1];
            xch[parity][wave][2] = part[2];
            if constexpr (EPB > 1)
                __hip_atomic_store(&flag_[es][parity][wave], posted + 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
#if SOFTROD_OCTO_PRIO
        if constexpr (EPB > 1) __builtin_amdgcn_s_setprio(0);     // posted: yield to the partner that has not
#endif
    };
    // RigidBodyBase.update_accelerations + the rate update under
    // BodyBoundaryCondition.compute_constrain_rates (constraint.py:60-85): with
    // w = (0, 0, wz) the gyroscopic term J w x w vanishes identically.  Every lane adds the
    // waves' partial sums in the same order (wave 0 first), so the head replicas stay
    // bit-identical; the order differs from the reference's loop over connections (arm by arm)
    // by the rounding of a sum of eight numbers.  LDS is double-buffered: ONE barrier per substep.
    // The head's step between two force evaluations: RigidBodyBase.update_accelerations + the rate
    // update under BodyBoundaryCondition.compute_constrain_rates (constraint.py:60-85) — with
    // w = (0, 0, wz) the gyroscopic term J w x w vanishes identically — then its kinematic step.
    // `pend` holds the waves' partial sums read from LDS right after the barrier, `hk` the step
    // the head takes with them (the first half step at entry, with zero loads).  (Running this
    // chain at the top of the next dynamic_n instead, to interleave it with the arms' geometry,
    // was measured and loses: 13.1 against 12.0 ms, profiles/README.md.)  Every lane adds
    // the partials in the same order (wave 0 first), so the head replicas stay bit-identical;
    // the order differs from the reference's loop over connections (arm by arm) by the rounding
    // of a sum of eight numbers.  LDS is double-buffered: ONE barrier per substep.
    double pend[MAXW == 2 ? 2 : 1][3] = {};
    double hk = P.half_dt;
    auto head_step = [&]() {
        if constexpr ((SOFTROD_OCTO_DIAG & 2) != 0) return;
        if constexpr (kMuscleArm) { if (P.head_fixed) return; }      // OneEndFixedBC: the head stays what the reset made it
        if constexpr ((SOFTROD_OCTO_BASE_MASK & 2) != 0) { if (!base) return; }
        double tot[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) tot[i] = (MAXW == 2) ? pend[0][i] + pend[MAXW == 2 ? 1 : 0][i] : pend[0][i];
        H.v[0] = fma(P.dt, tot[0] * head_inv_mass, H.v[0]);
        H.v[1] = fma(P.dt, tot[1] * head_inv_mass, H.v[1]);
        H.w[2] = fma(P.dt, P.head_invJ[2] * (-tot[2]), H.w[2]);
        head_kinematic(hk, H);
    };
    auto exchange = [&]() {       // after the arms' kinematic step: everyone has posted
        if constexpr ((SOFTROD_OCTO_DIAG & 4) != 0) return;
        if constexpr (EPB == 1) __syncthreads();
        else {                    // the partner wave runs on this SIMD: wait for its flag, not for a barrier
            while (__hip_atomic_load(&flag_[es][parity][wave ^ 1], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) <= posted)
                __builtin_amdgcn_s_sleep(1);
            posted += parity;     // both buffers of this pair used: the next pair carries the next count
        }
        if constexpr (MAXW == 2) {
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                pend[0][i] = xch[parity][0][i];
                pend[MAXW == 2 ? 1 : 0][i] = xch[parity][MAXW == 2 ? 1 : 0][i];   // zero if nw == 1 (cleared at entry)
            }
        } else {
#pragma unroll
            for (int i = 0; i < 3; ++i) pend[0][i] = 0.0;
            for (int w = 0; w < nw; ++w) {
#pragma unroll
                for (int i = 0; i < 3; ++i) pend[0][i] += xch[parity][w][i];
            }
        }
        parity ^= 1;
    };

    // an env that already holds a NaN is not integrated (see softrod_step_fast_kernel)
    bool dead = false;
    if (n_sub > 0 && epilogue) {
        bool bad = isnan(H.x[0]) || isnan(H.x[1]) || isnan(H.v[0]) || isnan(H.v[1]) || isnan(H.w[2]) ||
                   isnan(H.Q[0]) || isnan(H.Q[1]) || isnan(H.Q[3]) || isnan(H.Q[4]);
#pragma unroll
        for (int c = 0; c < 3; ++c)
            bad = bad || (arm_ok && r <= n && (isnan(L.x[0][c]) || isnan(L.v[0][c]))) ||
                  (arm_ok && r < n && isnan(L.w[0][c]));
#pragma unroll
        for (int c = 0; c < 9; ++c) bad = bad || (arm_ok && r < n && isnan(L.Q[0][c]));
        dead = env_any(bad, 0);
        if (dead) {
            poison_rod<1>(L);
            time = clock_after(P, S, time, n_sub);
        }
    }
    // The ghost slots between the arms (5 of every 16 lanes at 10 elements) sit the loop out with EXEC
    // cleared, like the idle lanes of the one-rod kernels (softrod_fast.hpp): lane 0 of every wave is
    // a base node, so the rendezvous and the LDS posts are unaffected, and a DPP shift that would read
    // a ghost returns 0 where it used to return the ghost's zeros.  Bit-identical; worth what the
    // box's power budget makes of it: 9.47 -> 9.34 ms on one box, 9.36 / 9.355 on another that was
    // already there (profiles/README.md r3k).  The invariants of softrod_fast.hpp's general_substeps
    // apply: lane 0 of every wave stays active and reaches exchange()'s s_barrier / LDS rendezvous
    // n_sub times like every other wave, and the __shfl_xor butterflies read 0 from a masked lane.
    // A/B-checked against the unmasked build by tests/test_gpu_mask_ab.py.
#ifndef SOFTROD_OCTO_GHOST_MASK
#define SOFTROD_OCTO_GHOST_MASK 1
#endif
    if (n_sub > 0 && !dead && live && (!SOFTROD_OCTO_GHOST_MASK || (arm_ok && r <= n))) {
        kinematic_n<1>(P.half_dt, C, L);
        head_normalize(H);                       // hk = dt/2 and zero loads: the head's first half step
        head_step();
        for (int s = 0; s < n_sub; ++s) {
#if SOFTROD_OCTO_PRIO
            if constexpr (EPB > 1) __builtin_amdgcn_s_setprio(SOFTROD_OCTO_PRIO);
#endif
            dynamic_n<F, 1, kMusclesCompiled<F>>(Pk, C, B, tid, L, joints);
            const bool last = (s == n_sub - 1);
            const double h = last ? P.half_dt : P.dt;
            kinematic_n<1>(h, C, L);
            exchange();
            hk = h;
            head_step();
        }
        time = clock_after(P, S, time, n_sub);
    }
    if constexpr ((SOFTROD_OCTO_BASE_MASK & 2) != 0) {
        // only the base lanes stepped the head inside the loop: everybody gets lane 0's for the epilogue
        // (all base lanes hold the same bits: the same operands in the same order)
#pragma unroll
        for (int i = 0; i < 3; ++i) { H.x[i] = __shfl(H.x[i], 0); H.v[i] = __shfl(H.v[i], 0); H.w[i] = __shfl(H.w[i], 0); }
#pragma unroll
        for (int i = 0; i < 9; ++i) H.Q[i] = __shfl(H.Q[i], 0);
    }
    if (live) {
        store_lane<1, F>(S, NR, row, lane, L);
        if (tid == 0) {
            S.time[env] = time;
            store_head(S, N, env, H);
        }
    }
    if constexpr (kMusclesCompiled<F>) {
        // rod.kappa[0] as this launch's last force evaluation cached it: what the muscle octopus envs' get_state reads
        if (mocto && live && n_sub > 0) S.kap[(size_t)row * kLanes + lane] = L.kap[0][0];
        if (mocto) return;
    }
    if (!epilogue) return;
    if constexpr (kMusclesCompiled<F>) {
        // ArmPullWeightEnv.step = ArmPushEnv.step after the loop (arm_push_env.py:288-347) on the arm alone
        if (live)
            env_epilogue_n<kRuntimeEnv, 1>(P, S, NR, row, lane, C, L, time, A, obs, reward, terminated, truncated, nullptr, pack);
        return;
    }

    // ---- FlatEnv.step epilogue, flat_env.py:330-408 ----
    const int adim = P.n_arm * nk;
    if (tid < adim && live) S.prev_action[(size_t)env * adim + tid] = actions[(size_t)env * adim + tid];
    bool bad = false;
#pragma unroll
    for (int c = 0; c < 3; ++c) bad = bad || isnan(L.x[0][c]) || isnan(L.v[0][c]);
    bad = bad && arm_ok && r <= n;
    sxy[tid][0] = L.x[0][0];
    sxy[tid][1] = L.x[0][1];
    if (tid == 0) scount = 0;
    const bool invalid = env_any(bad, 1);
    // arm crossings: pairs (i-1, i) for i = 0..n_arm-2, index -1 wrapping to the last arm
    int cnt = 0;
    const int per = n * n, total = (P.n_arm - 1) * per;
    for (int q = tid; q < total; q += nthr) {
        const int i = q / per, rem = q - i * per;
        const int ii = rem / n, jj = rem - ii * n;
        const int a1 = (i - 1 + P.n_arm) % P.n_arm, a2 = i;
        const int s1 = a1 * P.seg + ii, s2 = a2 * P.seg + jj;
        const double x1a = sxy[s1][0], x1b = sxy[s1 + 1][0], y1a = sxy[s1][1], y1b = sxy[s1 + 1][1];
        const double x2a = sxy[s2][0], x2b = sxy[s2 + 1][0], y2a = sxy[s2][1], y2b = sxy[s2 + 1][1];
        const bool c1 = fmin(x1a, x1b) <= fmax(x2a, x2b);
        const bool c2 = fmax(x1a, x1b) >= fmin(x2a, x2b);
        const bool c3 = fmin(y1a, y1b) <= fmax(y2a, y2b);
        const bool c4 = fmax(y1a, y1b) >= fmin(y2a, y2b);
        if (!(c1 && c2 && c3 && c4)) continue;
        double M[4][4] = {{x1b - x1a, 0.0, -1.0, 0.0}, {0.0, x2b - x2a, -1.0, 0.0},
                          {y1b - y1a, 0.0, 0.0, -1.0}, {0.0, y2b - y2a, 0.0, -1.0}};
        double bv[4] = {-x1a, -x2a, -y1a, -y2a};
        double t0, t1;
        if (!solve4_t01(M, bv, t0, t1)) continue;
        if (t0 >= 0.0 && t1 >= 0.0 && t0 <= 1.0 && t1 <= 1.0) ++cnt;
    }
    if (cnt) atomicAdd(&scount, cnt);
    __syncthreads();
    const int od = octo_obs_dim(P);
    float* o = out_row(obs, env, od, pack);
    if (!live) return;                         // (no barrier below)
    if (tid == 0) {
        const double tx = tgt[0] - H.x[0], ty = tgt[1] - H.x[1];
        const double dist = sqrt(tx * tx + ty * ty);
        double survive = 0.0, forward = 0.0;
        bool term = false;
        if (invalid) { term = true; survive = -50.0; }
        else {
            survive = -0.02 * (double)scount;
            const double bx = tgt[0] - before[0], by = tgt[1] - before[1];
            forward = (dist - sqrt(bx * bx + by * by)) / P.step_time;
            if (dist < 0.1) { survive = 100.0; term = true; }
        }
        double rew = forward - 0.0 + survive - 0.0;
        if (term) rew -= dist - 0.1;
        emit_scalars(o, od, pack, env, rew, term, time > P.final_time, reward, terminated, truncated,
                     S.needs_reset);
    }
    octo_write_obs(P, tid, L, H, tgt, actions + (size_t)env * adim, o);
}

// get_state after reset (or at any time): prev_action = the resident copy unless given.
__global__ void __launch_bounds__(1024)
softrod_octo_observe_kernel(const RodParams P, const StatePtrs S, const float* __restrict__ prev_action,
                            float* __restrict__ obs) {
    const int env = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nw = blockDim.x >> 6;
    const size_t N = (size_t)P.n_envs, NR = N * (size_t)nw;
    LaneN<1> L;
    load_lane<1, kRuntimeFeatures>(S, NR, env * nw + wave, lane, L);
    HeadState H;
    double tgt[2];
    load_head(S, N, env, H, tgt);
    const int adim = P.n_arm * P.n_action;
    const float* pa = (prev_action ? prev_action : S.prev_action) + (size_t)env * adim;
    octo_write_obs(P, tid, L, H, tgt, pa, obs + (size_t)env * octo_obs_dim(P));
}

// FlatEnv.reset -> build_octopus (octopus/build.py:52-217).
//   init[env][arm][18]  straight_rod description of each arm (start, step, end, Q rows)
//   target[env][2]      (2 - 0.5) * np_random.random(2) + 0.5, flat_env.py:221
// The head is Cylinder(start=(0,0,-r0), direction=e_z, normal=e_y, length 2 r0) (:103-105):
// centre (0,0,0), directors rows e_y, e_z x e_y = -e_x, e_z; finalize() applies the head
// constraint once, which leaves these exact values unchanged.
struct OctoResetArgs {
    const double* init;
    const double* target;
    const uint8_t* mask;
};

// One env's reset from its n_arm straight-rod records and its target; the fresh per-lane
// state is left in L, the head in H.
__device__ __forceinline__ void octo_reset_env(const RodParams& P, const StatePtrs& S, int env,
                                               const double* __restrict__ arms, const double* __restrict__ target,
                                               LaneN<1>& L, HeadState& H) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nw = blockDim.x >> 6;
    const size_t N = (size_t)P.n_envs, NR = N * (size_t)nw;
    const int row = env * nw + wave;
    const int n = P.n_elem;
    const int r = tid & (P.seg - 1);
    int arm = tid >> P.seg_shift;
    if (arm >= P.n_arm) arm = P.n_arm - 1;   // slots past the last arm: finite filler
    const double* in = arms + (size_t)arm * 18;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double xv = in[c] + (double)r * in[3 + c];
        if (r == n) xv = in[6 + c];
        L.x[0][c] = xv;
        L.v[0][c] = 0.0; L.w[0][c] = 0.0; L.kap[0][c] = 0.0; L.rk[0][c] = 0.0;
    }
#pragma unroll
    for (int c = 0; c < 9; ++c) L.Q[0][c] = in[9 + c];
    double len2 = 0.0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        L.t[0][c] = from_next(L.x[0][c]) - L.x[0][c];
        len2 += L.t[0][c] * L.t[0][c];
    }
    const double len = sqrt(len2) + P.eps_length;
#pragma unroll
    for (int c = 0; c < 3; ++c) L.t[0][c] /= len;
    store_lane<1, kRuntimeFeatures>(S, NR, row, lane, L);
    const size_t m = (size_t)row * kLanes + lane;
#pragma unroll
    for (int c = 0; c < 3; ++c) S.rkap[c * NR * kLanes + m] = 0.0;
    S.envmem[m] = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) { H.x[i] = P.head_center[i]; H.v[i] = 0.0; H.w[i] = 0.0; }
    if (P.features & SOFTROD_FEAT_COOMM_MUSCLES) {     // fresh muscle objects and SuckerController (reset_rod says why)
        if (S.mact) {
#pragma unroll
            for (int mm = 0; mm < SOFTROD_MAX_MUSCLES; ++mm) S.mact[((size_t)mm * NR + row) * kLanes + lane] = 0.0;
        }
        if (tid == 0 && is_push_env(P.env_kind)) {
            S.sucker_idx[env] = P.sucker_index[0];
            S.sucker[env] = P.sucker_ratio0;
        }
        if (is_mocto_env(P.env_kind)) {
            // fresh SuckerControllers on every arm (crawl_env.py:146-155: index, reduction_ratio 1.0; turned on after
            // finalize), a straight arm's kappa
            const int a = tid >> P.seg_shift;
            if (r == 0 && a < P.n_arm) {
                const size_t NA = N * (size_t)P.n_arm, ai = (size_t)env * P.n_arm + a;
#pragma unroll
                for (int j = 0; j < SOFTROD_MAX_SUCKERS; ++j) {
                    S.sucker_idx[(size_t)j * NA + ai] = P.sucker_index[j];
                    S.sucker[(size_t)j * NA + ai] = (j < P.n_suckers) ? P.sucker_ratio0 : 0.0;
                }
            }
            S.kap[m] = 0.0;
        }
    }
    const double Q0[9] = {0.0, 1.0, 0.0, -1.0, 0.0, 0.0, 0.0, 0.0, 1.0};
#pragma unroll
    for (int i = 0; i < 9; ++i) H.Q[i] = Q0[i];
    if (tid == 0) {
        S.time[env] = 0.0;
        if (S.needs_reset) S.needs_reset[env] = 0;
        store_head(S, N, env, H);
        S.head[(size_t)18 * N + env] = target[0];
        S.head[(size_t)19 * N + env] = target[1];
        if (is_mocto_env(P.env_kind)) {      // the env's target (three numbers: ReachEnv's is a point in space) and, fourth,
            S.aux[(size_t)0 * N + env] = target[0];      // the episode's own final_time (0: the config's; crawl_env.py:135-136)
            S.aux[(size_t)1 * N + env] = target[1];
            S.aux[(size_t)2 * N + env] = target[2];
            S.aux[(size_t)5 * N + env] = target[3];
            S.aux[(size_t)3 * N + env] = H.x[0];
            S.aux[(size_t)4 * N + env] = H.x[1];
        }
    }
}

__global__ void __launch_bounds__(1024)
softrod_octo_reset_kernel(const RodParams P, const StatePtrs S, const OctoResetArgs A) {
    const int env = blockIdx.x;
    if (A.mask && !A.mask[env]) return;
    LaneN<1> L;
    HeadState H;
    octo_reset_env(P, S, env, A.init + (size_t)env * P.n_arm * 18, A.target + (is_mocto_env(P.env_kind) ? 4 : 2) * (size_t)env, L, H);
}

// Device-side auto-reset pass for OctoFlat (see softrod_autoreset_kernel): the queue record
// is [n_arm][18] arm frames followed by the target (2).
__global__ void __launch_bounds__(1024)
softrod_octo_autoreset_kernel(const RodParams P, const StatePtrs S, float* __restrict__ obs,
                              double* __restrict__ reward, uint8_t* __restrict__ terminated,
                              uint8_t* __restrict__ truncated, const int pack) {
    const int env = blockIdx.x;
    const int tid = threadIdx.x;
    if (!S.needs_reset[env]) return;
    const size_t N = (size_t)P.n_envs;
    const int k = S.q_consumed[env];
    if (k >= S.q_produced[env]) {
        if (tid == 0) atomicAdd(S.q_underflow, 1);
        return;
    }
    const double* in = S.queue + ((size_t)(k % S.q_depth) * N + env) * (size_t)S.q_record;
    const bool mocto = is_mocto_env(P.env_kind);
    const double tgt[4] = {in[P.n_arm * 18], in[P.n_arm * 18 + 1], mocto ? in[P.n_arm * 18 + 2] : 0.0,
                           mocto ? in[P.n_arm * 18 + 3] : 0.0};
    LaneN<1> L;
    HeadState H;
    octo_reset_env(P, S, env, in, tgt, L, H);
    const bool push = is_push_env(P.env_kind);
    const int od = push ? env_obs_dim(P) : (mocto ? mocto_obs_dim(P) : octo_obs_dim(P));
    float* o = out_row(obs, env, od, pack);
    if (mocto) {
        __syncthreads();                       // thread 0's target rows are there (get_state reads them)
        mocto_write_obs(P, S, env, tid, L.x[0], L.v[0], 0.0, H, o);
    } else if (push) {
        float pa[2] = {S.prev_action[7 * (size_t)env], S.prev_action[7 * (size_t)env + 1]};
        (void)push_get_state_n<1>(P, tid & 63, L, pa, o, false);
    } else
        octo_write_obs(P, tid, L, H, tgt, S.prev_action + (size_t)env * (P.n_arm * P.n_action), o);
    __syncthreads();                           // every thread has read needs_reset / q_consumed
    if (tid == 0) {
        emit_scalars(o, od, pack, env, 0.0, false, false, reward, terminated, truncated, S.needs_reset);
        S.skip[env] = 1;
        S.q_consumed[env] = k + 1;
    }
}

}  // namespace softrod
