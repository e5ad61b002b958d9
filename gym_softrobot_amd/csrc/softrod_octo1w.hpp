// softrod_octo1w.hpp — OctoFlat-v0, the reference shape (n_arm * seg = 128 slots), as ONE wavefront
// per env with TWO slots per lane.  A/B variant of softrod_octo.hpp's two-wave kernel (VERDICT r2
// "next" #5a: re-measure the structural option on today's loop); selected with
// SOFTROD_OCTO_ONE_WAVE=1 at softrod_create, never by default unless it wins (profiles/README.md).
//
// What changes against the two-wave form: lane k owns slots 2k, 2k + 1 of the env's 128 (arm a =
// slots 16a .. 16a + 10, so an arm is 8 lanes and its base node is always the FIRST slot of a
// lane); the next node of a first slot is the same lane's second slot (no DPP); the head's net
// joint load is an in-wave xor-butterfly over the base lanes followed by a read of lane 0 — no LDS
// exchange, no flag, no barrier, no partner wave; the instruction stream of a substep carries two
// independent slots (ILP 2) at one wave per SIMD and 512 vector registers.  The state rows, the
// reset / observe / auto-reset kernels and the arithmetic per slot are those of the two-wave
// kernel (the same dynamic_n / kinematic_n / plane_contact_n), so results differ from it only in the
// summation order of the eight joint loads (a butterfly over eight lanes instead of two wave sums).
#pragma once

namespace softrod {

template <unsigned F>
__global__ void __launch_bounds__(kLanes, 1)
softrod_octo1w_step_kernel(const RodParams P, const StatePtrs S, const float* __restrict__ actions,
                           float* __restrict__ obs, double* __restrict__ reward,
                           uint8_t* __restrict__ terminated, uint8_t* __restrict__ truncated,
                           const int n_sub, const int epilogue, const int pack) {
    constexpr int EPL = 2;
    constexpr int SLOTS = kLanes * EPL;
    __shared__ double sxy[SLOTS + 1][2];
    __shared__ int scount;
    const int env = blockIdx.x, lane = threadIdx.x;
    const size_t N = (size_t)P.n_envs;
    if (epilogue && S.skip && S.skip[env]) {   // reset by the auto-reset pass of this env.step
        if (lane == 0) S.skip[env] = 0;
        return;
    }
    if constexpr (SOFTROD_OCTO_CONTACT_LDS && (F & kFeatPlaneZup) != 0) stage_contact_params(P);
    const int n = P.n_elem, nk = P.n_action;
    int rr[EPL], arm[EPL];
    bool arm_ok[EPL];
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        const int t = lane * EPL + s;
        rr[s] = t & (P.seg - 1);
        arm[s] = t >> P.seg_shift;
        arm_ok[s] = arm[s] < P.n_arm;
    }

    LaneN<EPL> L;
    load_lane<EPL, F>(S, N, env, lane, L);      // the env's two 64-slot rows are one 128-slot row
    sanitize_unused_slots<EPL>(P, lane, L);
    HeadState H;
    double tgt[2];
    load_head(S, N, env, H, tgt);
    H.v[2] = 0.0; H.w[0] = 0.0; H.w[1] = 0.0;
    const double before[2] = {H.x[0], H.x[1]};

    if (actions) {     // set_action (flat_env.py:288-311)
#pragma unroll
        for (int s = 0; s < EPL; ++s) {
            double rk0 = 0.0;
            if (arm_ok[s] && rr[s] < n - 1) {
                const float* a = actions + (size_t)env * (P.n_arm * nk) + arm[s] * nk;
                for (int j = 0; j < nk; ++j) rk0 += S.basis[rr[s] * nk + j] * (double)a[j];
            }
            L.rk[s][0] = rk0;
            S.rkap[(size_t)env * SLOTS + lane * EPL + s] = rk0;
        }
    }

    // joint frame of the arm whose base node is this lane's first slot (seg is even: a base node
    // is never a second slot)
    const bool base = arm_ok[0] && rr[0] == 0;
    const double ang = (360.0 / (double)P.n_arm * (double)arm[0]) / 180.0 * M_PI;
    const double ct = cos(ang), st = sin(ang);

    EnvAction A;
#pragma unroll
    for (int i = 0; i < 7; ++i) A.a[i] = 0.0f;
    A.force = 0.0;
    ConstN<EPL> C;
    build_const<F, EPL>(P, lane, A, C);
    BcTargets B;
#pragma unroll
    for (int i = 0; i < 3; ++i) { B.pos[i] = 0.0; B.vel[i] = 0.0; }
#pragma unroll
    for (int i = 0; i < 9; ++i) B.Q[i] = 0.0;
    RodParams Pk = P;
    if (!has<F>(P, SOFTROD_FEAT_ANALYTICAL_DAMPER)) Pk.damp_t = 1.0;
    const double head_inv_mass = 1.0 / P.head_mass;

    double time = S.time[env];
    double pend[3] = {0.0, 0.0, 0.0};
    double hk = P.half_dt;

    auto joints = [&](double (&f)[EPL][3], double (&tq)[EPL][3], const LaneN<EPL>& Lc, const double (&xn)[EPL][3]) {
        // FixedJoint2Rigid (joint.py:48-219), as in softrod_octo.hpp; slot 0 of the lane only
        const double b0 = H.Q[3], b1 = H.Q[4];
        const double dir[2] = {-(ct * b0 - st * b1), -(st * b0 + ct * b1)};
        const double pos[3] = {fma(dir[0], P.head_radius, H.x[0]), fma(dir[1], P.head_radius, H.x[1]), 0.0};
        double dv[3], d2 = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) { dv[i] = Lc.x[0][i] - pos[i]; d2 = fma(dv[i], dv[i], d2); }
        const bool apart = d2 > (2.220446049250313e-12 * 2.220446049250313e-12);
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
        force[0] = -P.joint_kt * (xn[0][0] - fma(P.rest_len, dir[0], pos[0]));
        force[1] = -P.joint_kt * (xn[0][1] - fma(P.rest_len, dir[1], pos[1]));
        force[2] = -P.joint_kt * xn[0][2];
        const double tj[3] = {link[1] * force[2] - link[2] * force[1], link[2] * force[0] - link[0] * force[2],
                              link[0] * force[1] - link[1] * force[0]};
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            f[0][i] -= base ? fj[i] : 0.0;
            const double* Q = Lc.Q[0];
            tq[0][i] += base ? fma(Q[3 * i + 2], tj[2], fma(Q[3 * i + 1], tj[1], Q[3 * i] * tj[0])) : 0.0;
        }
        // the eight base lanes (every seg / 2 = 8th) are summed by a xor-butterfly; lane 0 then holds
        // the env's net joint load, and everybody reads it from there (a scalar register)
        double part[3] = {base ? fj[0] : 0.0, base ? fj[1] : 0.0, base ? tj[2] : 0.0};
        for (int off = P.seg >> 1; off < kLanes; off <<= 1) {
#pragma unroll
            for (int i = 0; i < 3; ++i) part[i] += __shfl_xor(part[i], off);
        }
#pragma unroll
        for (int i = 0; i < 3; ++i) pend[i] = __shfl(part[i], 0);
    };
    auto head_step = [&]() {
        H.v[0] = fma(P.dt, pend[0] * head_inv_mass, H.v[0]);
        H.v[1] = fma(P.dt, pend[1] * head_inv_mass, H.v[1]);
        H.w[2] = fma(P.dt, P.head_invJ[2] * (-pend[2]), H.w[2]);
        head_kinematic(hk, H);
    };

    __syncthreads();      // the staged contact constants
    bool dead = false;
    if (n_sub > 0 && epilogue) {
        bool bad = isnan(H.x[0]) || isnan(H.x[1]) || isnan(H.v[0]) || isnan(H.v[1]) || isnan(H.w[2]) ||
                   isnan(H.Q[0]) || isnan(H.Q[1]) || isnan(H.Q[3]) || isnan(H.Q[4]);
#pragma unroll
        for (int s = 0; s < EPL; ++s) {
#pragma unroll
            for (int c = 0; c < 3; ++c)
                bad = bad || (arm_ok[s] && rr[s] <= n && (isnan(L.x[s][c]) || isnan(L.v[s][c]))) ||
                      (arm_ok[s] && rr[s] < n && isnan(L.w[s][c]));
#pragma unroll
            for (int c = 0; c < 9; ++c) bad = bad || (arm_ok[s] && rr[s] < n && isnan(L.Q[s][c]));
        }
        dead = __any(bad);
        if (dead) {
            poison_rod<EPL>(L);
            time = clock_after(P, S, time, n_sub);
        }
    }
    if (n_sub > 0 && !dead) {
        kinematic_n<EPL>(P.half_dt, C, L);
        head_normalize(H);
        head_step();                               // hk = dt/2, zero loads: the head's first half step
        for (int s = 0; s < n_sub; ++s) {
            dynamic_n<F, EPL>(Pk, C, B, lane, L, joints);
            const double h = (s == n_sub - 1) ? P.half_dt : P.dt;
            kinematic_n<EPL>(h, C, L);
            hk = h;
            head_step();
        }
        time = clock_after(P, S, time, n_sub);
    }
    store_lane<EPL, F>(S, N, env, lane, L);
    if (lane == 0) {
        S.time[env] = time;
        store_head(S, N, env, H);
    }
    if (!epilogue) return;

    // ---- FlatEnv.step epilogue, flat_env.py:330-408 (as softrod_octo.hpp) ----
    const int adim = P.n_arm * nk;
    if (lane < adim) S.prev_action[(size_t)env * adim + lane] = actions[(size_t)env * adim + lane];
    bool bad = false;
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        bool b = false;
#pragma unroll
        for (int c = 0; c < 3; ++c) b = b || isnan(L.x[s][c]) || isnan(L.v[s][c]);
        bad = bad || (b && arm_ok[s] && rr[s] <= n);
        sxy[lane * EPL + s][0] = L.x[s][0];
        sxy[lane * EPL + s][1] = L.x[s][1];
    }
    if (lane == 0) { scount = 0; sxy[SLOTS][0] = 0.0; sxy[SLOTS][1] = 0.0; }
    const bool invalid = __any(bad);
    __syncthreads();
    int cnt = 0;
    const int per = n * n, total = (P.n_arm - 1) * per;
    for (int q = lane; q < total; q += kLanes) {
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
    if (lane == 0) {
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
    // FlatEnv.get_state, one slot at a time through the one-slot writer
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        LaneN<1> one;
#pragma unroll
        for (int c = 0; c < 3; ++c) { one.x[0][c] = L.x[s][c]; one.v[0][c] = L.v[s][c]; one.kap[0][c] = L.kap[s][c]; }
        octo_write_obs(P, lane * EPL + s, one, H, tgt, actions + (size_t)env * adim, o);
    }
}

}  // namespace softrod
