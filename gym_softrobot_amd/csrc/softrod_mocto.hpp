// softrod_mocto.hpp — the env code of the muscle octopus: CrawlEnv, ArmTwoEnv, ReachEnv
// (SOFTROD_ENV_CRAWL / _ARM_TWO / _REACH; OctoCrawl-v0, OctoArmTwo-v0, OctoReach-v0).
//
// The body — n_arm tapered muscle arms on a rigid head (octopus/build_muscle_octopus.py) — is stepped by
// softrod_octo_step_kernel's muscle-arm instantiation (softrod_octo.hpp), which knows nothing of these envs: their
// set_action runs BEFORE it (softrod_mocto_action_kernel: the action becomes per-element activations, sucker indices
// and ratios in the resident rows), their get_state / reward / termination AFTER it (softrod_mocto_epilogue_kernel, on
// the stored state).  Two launches of a few microseconds around 800 substeps keep three envs' worth of branches and
// their registers out of a loop that already spills.
//
//   set_action   crawl_env.py:220-241   arm_two_env.py:205-251   reach_env.py:192-205
//   get_state    crawl_env.py:175-218   arm_two_env.py:166-203   reach_env.py:150-190
//   step         crawl_env.py:243-303   arm_two_env.py:253-343   reach_env.py:207-270
// mirrored by tests/oracle_mocto.py over the C oracle's body.  PARITY UNPINNED: the muscle law under it is the restated
// COOMM model (softrod_muscle.hpp); the env code itself is pinned against the executed reference files
// (tools/make_muscle_env_golden.py).
//
// Lane layout: the step kernel's (arm a of the env on slots a * seg .. a * seg + n_elem of the env's nw waves).
#pragma once

namespace softrod {

__device__ __forceinline__ bool is_mocto_env(int env) {
    return env == SOFTROD_ENV_CRAWL || env == SOFTROD_ENV_ARM_TWO || env == SOFTROD_ENV_REACH;
}

__device__ __forceinline__ int mocto_obs_dim(const RodParams& P) {
    const int n = P.n_elem, na = P.n_arm, nk = P.n_action;
    if (P.env_kind == SOFTROD_ENV_ARM_TWO) return na * ((n - 1) * 2 + nk + na + 3);
    return na * ((n - 1) + (n + 1) * 4 + nk + na + (P.env_kind == SOFTROD_ENV_CRAWL ? 17 : 18));
}

// `.astype(np.float32)` followed by np.nan_to_num (crawl_env.py:216-217)
__device__ __forceinline__ float mocto_f32(double v) {
    const float f = (float)v;
    return isnan(f) ? 0.0f : (isinf(f) ? copysignf(3.4028234663852886e38f, f) : f);
}

// get_state on one thread's slot: x, v = the node's position / velocity, kap0 = kappa[0] of its Voronoi vertex
__device__ __forceinline__ void mocto_write_obs(const RodParams& P, const StatePtrs& S, int env, int tid,
                                                const double x[3], const double v[3], double kap0,
                                                const HeadState& H, float* __restrict__ o) {
    const size_t N = (size_t)P.n_envs;
    const int n = P.n_elem, na = P.n_arm, nk = P.n_action;
    const int r = tid & (P.seg - 1), arm = tid >> P.seg_shift;
    if (arm >= na) return;
    const float* pa = S.prev_action + (size_t)env * (size_t)(na * nk) + (size_t)arm * nk;
    if (P.env_kind == SOFTROD_ENV_ARM_TWO) {
        // [kappa0 (n-1) | _prev_kappa (n-1) | prev_action (nk) | eye(n_arm)[arm] | head v (3)]; _prev_kappa <- kappa0
        const int width = (n - 1) * 2 + nk + na + 3;
        float* row = o + (size_t)arm * width;
        if (r < n - 1) {
            float* pk = S.prev_kappa + (size_t)env * (size_t)(na * (n - 1)) + (size_t)arm * (n - 1) + r;
            row[r] = mocto_f32(kap0);
            row[(n - 1) + r] = mocto_f32((double)*pk);
            *pk = (float)kap0;                                   // self._prev_kappa[...] = kappa_state (float32 array)
        }
        for (int q = r; q < nk; q += P.seg) row[2 * (n - 1) + q] = mocto_f32((double)pa[q]);
        for (int q = r; q < na; q += P.seg) row[2 * (n - 1) + nk + q] = (q == arm) ? 1.0f : 0.0f;
        if (r < 3) row[2 * (n - 1) + nk + na + r] = mocto_f32(H.v[r]);
        return;
    }
    // [kappa0 (n-1) | x (n+1) | y | vx | vy | prev_action (nk) | eye(n_arm)[arm] | shared]
    // shared = [target (2: Crawl, 3: Reach) | head x (3) | head v (3) | head directors (9)]
    const int nt = P.env_kind == SOFTROD_ENV_CRAWL ? 2 : 3;
    const int width = (n - 1) + (n + 1) * 4 + nk + na + nt + 15;
    float* row = o + (size_t)arm * width;
    if (r < n - 1) row[r] = mocto_f32(kap0);
    if (r <= n) {
        row[(n - 1) + r] = mocto_f32(x[0]);
        row[(n - 1) + (n + 1) + r] = mocto_f32(x[1]);
        row[(n - 1) + 2 * (n + 1) + r] = mocto_f32(v[0]);
        row[(n - 1) + 3 * (n + 1) + r] = mocto_f32(v[1]);
    }
    const int off = (n - 1) + 4 * (n + 1);
    for (int q = r; q < nk; q += P.seg) row[off + q] = mocto_f32((double)pa[q]);
    for (int q = r; q < na; q += P.seg) row[off + nk + q] = (q == arm) ? 1.0f : 0.0f;
    float* sh = row + off + nk + na;
    if (r < nt) sh[r] = mocto_f32(S.aux[(size_t)r * N + env]);         // Crawl: np.float32(5), 0; Reach: float64 -> float32
    if (r < 3) { sh[nt + r] = mocto_f32(H.x[r]); sh[nt + 3 + r] = mocto_f32(H.v[r]); }
    if (r < 9) sh[nt + 6 + r] = mocto_f32(H.Q[r]);
}

// set_action.  grid = n_envs, block = 64 * nw (the step kernel's).
__global__ void __launch_bounds__(1024)
softrod_mocto_action_kernel(const RodParams P, const StatePtrs S, const float* __restrict__ actions, const int n_sub) {
    const int env = blockIdx.x, tid = threadIdx.x;
    if (S.skip && S.skip[env]) return;          // restarted by the auto-reset pass of this env.step: the action is not applied
    const int lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const size_t N = (size_t)P.n_envs, NR = N * (size_t)nw, NA = N * (size_t)P.n_arm;
    const size_t row = (size_t)env * nw + wave;
    const int n = P.n_elem, na = P.n_arm, nk = P.n_action;
    const int r = tid & (P.seg - 1), arm = tid >> P.seg_shift;
    const int adim = na * nk;
    const float* act = actions + (size_t)env * adim;
    if (arm < na) {
        const float* a = act + (size_t)arm * nk;
        const size_t ai = (size_t)env * na + arm;
        double* m0 = S.mact + ((size_t)0 * NR + row) * kLanes + lane;
        double* m1 = S.mact + ((size_t)1 * NR + row) * kLanes + lane;
        double* m2 = S.mact + ((size_t)2 * NR + row) * kLanes + lane;
        if (P.env_kind == SOFTROD_ENV_CRAWL) {
            // (location, activation, r_ratio): index = int(np.clip(location * n_elems, 0, n_elems - 1)) — the product of a
            // float32 and a Python int is a float32 (NumPy 2); the transverse layer takes the scalar; the ratio as given
            if (r < n) *m2 = (double)a[1];
            if (r == 0) {
                const float loc = fminf(fmaxf(a[0] * (float)n, 0.0f), (float)(n - 1));
                S.sucker_idx[ai] = (int)loc;
                S.sucker[ai] = (double)a[2];
            }
        } else if (P.env_kind == SOFTROD_ENV_ARM_TWO) {
            // sucker ratios a[0:3]; LM = a[3:6] - 0.5 (float32), LM1 = max(LM, 0), LM2 = |min(LM, 0)|, TM = a[6:9]; each
            // interpolated over the elements: interp1d(control_location, [0, *act, 0], "cubic")(range(n)) = basis @ act
            if (r < n) {
                double s0 = 0.0, s1 = 0.0, s2 = 0.0;
#pragma unroll
                for (int j = 0; j < 3; ++j) {
                    const double b = S.basis[r * 3 + j];
                    const double lm = (double)(a[3 + j] - 0.5f);
                    s0 = fma(b, fmax(lm, 0.0), s0);
                    s1 = fma(b, fabs(fmin(lm, 0.0)), s1);
                    s2 = fma(b, (double)a[6 + j], s2);
                }
                *m0 = s0; *m1 = s1; *m2 = s2;
            }
            if (r < 3) S.sucker[(size_t)r * NA + ai] = (double)a[r];
        } else {
            // ReachEnv: layer j takes action[arm, n * j : n * (j + 1)]
            if (r < n) { *m0 = (double)a[r]; *m1 = (double)a[n + r]; *m2 = (double)a[2 * n + r]; }
        }
    }
    for (int q = tid; q < adim; q += blockDim.x) S.prev_action[(size_t)env * adim + q] = act[q];
    // xposbefore (crawl_env.py:248).  A launch of zero substeps leaves the rows alone: the replay of the executed
    // reference's step() installs the post-loop state and the pre-loop head position by hand (tests/test_gpu_muscle_octopus.py)
    if (tid == 0 && n_sub > 0) {
        S.aux[(size_t)3 * N + env] = S.head[(size_t)0 * N + env];
        S.aux[(size_t)4 * N + env] = S.head[(size_t)1 * N + env];
    }
}

// get_state, and with `scalars` the rest of step() after the substep loop.  grid = n_envs, block = 64 * nw.
__global__ void __launch_bounds__(1024)
softrod_mocto_epilogue_kernel(const RodParams P, const StatePtrs S, float* __restrict__ obs,
                              double* __restrict__ reward, uint8_t* __restrict__ terminated,
                              uint8_t* __restrict__ truncated, const int scalars, const int pack) {
    __shared__ double tipd[16];
    const int env = blockIdx.x, tid = threadIdx.x;
    if (scalars && S.skip && S.skip[env]) {     // the auto-reset pass wrote this env's outputs; the flag is ours to clear
        __syncthreads();
        if (tid == 0) S.skip[env] = 0;
        return;
    }
    const int lane = tid & 63, wave = tid >> 6, nw = blockDim.x >> 6;
    const size_t N = (size_t)P.n_envs, NR = N * (size_t)nw;
    const size_t m = ((size_t)env * nw + wave) * kLanes + lane;
    const int n = P.n_elem, na = P.n_arm;
    const int r = tid & (P.seg - 1), arm = tid >> P.seg_shift;
    double x[3], v[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { x[c] = S.pos[c * NR * kLanes + m]; v[c] = S.vel[c * NR * kLanes + m]; }
    const double kap0 = S.kap[m];
    HeadState H;
    double unused[2];
    load_head(S, N, env, H, unused);
    const int od = mocto_obs_dim(P);
    float* o = out_row(obs, env, od, pack);
    mocto_write_obs(P, S, env, tid, x, v, kap0, H, o);
    if (!scalars) return;
    // _isnan_check over every arm's positions and velocities
    const bool node = arm < na && r <= n;
    const bool bad = node && (isnan(x[0]) || isnan(x[1]) || isnan(x[2]) || isnan(v[0]) || isnan(v[1]) || isnan(v[2]));
    const double tg[3] = {S.aux[env], S.aux[N + env], S.aux[2 * N + env]};
    if (P.env_kind == SOFTROD_ENV_REACH && node && r == n) {     // the arm's tip: |target - tip|
        const double d0 = tg[0] - x[0], d1 = tg[1] - x[1], d2 = tg[2] - x[2];
        tipd[arm] = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
    }
    const bool invalid = __syncthreads_or(bad ? 1 : 0) != 0;
    if (tid != 0) return;
    const double time = S.time[env];
    // CrawlEnv(config_random_final_time=True) draws its final_time per episode (crawl_env.py:135-136): aux row 5, 0 = the config's
    const double ft_env = S.aux[5 * N + env];
    const double final_time = ft_env > 0.0 ? ft_env : P.final_time;
    bool term = false, trunc = false;
    double survive = 0.0, forward = 0.0, rew;
    if (P.env_kind == SOFTROD_ENV_REACH) {                       // reach_env.py:220-262
        if (invalid) { term = true; survive = -5.0; }
        else {
            double dmin = tipd[0];
            for (int a = 1; a < na; ++a) dmin = (tipd[a] < dmin) ? tipd[a] : dmin;     // Python min(): first of equals, NaN-blind
            dmin = dmin / 0.25;
            forward = -(dmin * dmin);
            if (dmin < 0.1) { survive = 5.0; term = true; }
            if (time > final_time) trunc = true;
        }
        rew = forward + survive;
        if (isnan(rew)) { rew = -5.0; term = true; }
    } else {                                                      // crawl_env.py:264-299, arm_two_env.py:290-340
        double after = 0.0;
        if (invalid) { term = true; survive = -5.0; }
        else {
            const double bx = tg[0] - S.aux[3 * N + env], by = tg[1] - S.aux[4 * N + env];
            const double ax = tg[0] - H.x[0], ay = tg[1] - H.x[1];
            after = sqrt(ax * ax + ay * ay);
            forward = (sqrt(bx * bx + by * by) - after) * 1e2;
            if (after < 0.2) { survive = 5.0; term = true; }
        }
        if (!term && time > final_time) {
            if (P.env_kind == SOFTROD_ENV_ARM_TWO) forward -= after;
            trunc = true;
        }
        rew = forward - 0.0 + survive - 0.0;
        if (isnan(rew)) {
            // ArmTwoEnv: reward = -5; CrawlEnv: `reward -= 5` leaves the NaN and min(100.0, nan) returns its first argument
            rew = (P.env_kind == SOFTROD_ENV_ARM_TWO) ? -5.0 : 100.0;
            term = true;
        }
    }
    rew = (rew > 100.0) ? 100.0 : rew;                           // min(self.reward_range, reward)
    emit_scalars(o, od, pack, env, rew, term, trunc, reward, terminated, truncated, S.needs_reset);
}

}  // namespace softrod
