// softrod_muscle.hpp — SOFTROD_FEAT_COOMM_MUSCLES: the force / couple of COOMM's muscle layers and their
// equivalent external loads, per substep, on the one-rod-per-wave lane layout.
//
// What it stands in for: `ApplyMuscles(muscles=[LongitudinalMuscle, LongitudinalMuscle, TransverseMuscle])`
// registered as a forcing on the arm (gym_softrobot/envs/octopus/arm_push_env.py:197-212, :596-604,
// build_muscle_octopus.py:160-172) over the layers of create_es_muscle_layers (octopus/build.py:295-338), with the
// activations `apply_activation` wrote (arm_push_env.py:257-271).  COOMM itself (git pin, uv.lock:173-175) is NOT on
// disk: the arithmetic restates the published model (Chang et al., Proc. R. Soc. A 479:20220593, 2023, section 2(c)) in
// the operation order recalled from coomm/actuations/muscles/muscle.py and coomm/actuations/actuation.py — PARITY
// UNPINNED; every recalled detail is a field of softrod_config (include/softrod.h) mirrored by the oracle's
// apply_muscles (oracle/softrod_oracle.c), against which tests/test_gpu_muscles.py holds this file.
//
// Per element (lane k owns element k; Voronoi vertex k sits between elements k and k + 1):
//   kav   = 1/2 (kappa_{k-1} + kappa_k)          average2D: Voronoi -> elements, half weights at both ends (DPP shift)
//   x_m   = radius ratio_m                       radius = r0 sqrt(l_rest / l) (current) or r0
//   nu_m  = e Q t + kav x x_m ;  l_m = |nu_m| (longitudinal) or |nu_m|^-1/2 (transverse) ;  t_m = nu_m / |nu_m|
//   F_m   = activation strength max(fl(l_m), 0) ;  f += F_m t_m ;  c += x_m x F_m t_m
// then  F_ext += D^h(Q^T f),  tau_ext += D^h(c_v) + A^h(kappa x c_v D^) + (e Q t) x f l^,  c_v = 1/2 (c_k + c_{k+1})
// (muscle_form 1: PyElastica's internal-load form, Q^T f / e, c_v / eps^3, (Q t) x f l^).  Two DPP shifts of three
// values in, two out: no LDS.  A layer whose activation x strength is zero in the whole wave is skipped (one
// s_cbranch): OctoArmPush-v0 drives the transverse layer only, and only while its action is 0.
#pragma once

namespace softrod {

// FASTM: Newton-refined v_rsq_f64 (softrod_kernels.hpp fast_rsqrt) for the two inverse square roots; false: sqrt and
// IEEE division as the oracle writes them (the LIBM kernel).
// RESIDENT: the layers' constants sit in registers for the whole launch (ConstN.mr / .amp: the instantiations
// compiled FOR a muscle feature set); false: they are read from their L2-resident rows where they are used — the
// run-time-mask and LIBM kernels carry every feature's code and must not pay 32 VGPRs for one they rarely run.
template <int EPL, bool FASTM, bool RESIDENT>
__device__ __forceinline__ void muscle_loads_n(const RodParams& P, const ConstN<EPL>& C, int lane, const LaneN<EPL>& L,
                                               const double (&e)[EPL], const double (&il)[EPL],
                                               const double (&qt)[EPL][3], const double (&kv)[EPL][3],
                                               const double (&e3v)[EPL], const double (&r0s)[EPL],
                                               double (&f)[EPL][3], double (&tq)[EPL][3]) {
    const int n = P.n_elem;
    double kav[EPL][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double a[EPL], o[EPL];
#pragma unroll
        for (int s = 0; s < EPL; ++s) a[s] = kv[s][c];
        shift_prev<EPL>(a, o);
#pragma unroll
        for (int s = 0; s < EPL; ++s) kav[s][c] = 0.5 * (kv[s][c] + o[s]);
    }
    double fi[EPL][3], ce[EPL][3], rad[EPL], sh[EPL][3];
    bool elem_valid[EPL];
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        elem_valid[s] = slot_local(P, lane * EPL + s) < n;
        const double ilv = elem_valid[s] ? il[s] : 1.0;
        const double scale = P.muscle_cur_radius ? ilv : P.inv_rest_len;      // r0 sqrt(l_rest / l) = r0s sqrt(1 / l)
        rad[s] = r0s[s] * (FASTM ? scale * fast_rsqrt(scale) : sqrt(scale));
#pragma unroll
        for (int c = 0; c < 3; ++c) { fi[s][c] = 0.0; ce[s][c] = 0.0; sh[s][c] = e[s] * qt[s][c]; }
    }
#pragma unroll
    for (int m = 0; m < SOFTROD_MAX_MUSCLES; ++m) {
        if (m >= P.n_muscles) continue;
        double amp[EPL], mr[EPL][3];
        bool live = false;
#pragma unroll
        for (int s = 0; s < EPL; ++s) {
            if constexpr (RESIDENT) {
                amp[s] = C.amp[s][m]; mr[s][0] = C.mr[s][m][0]; mr[s][1] = C.mr[s][m][1]; mr[s][2] = C.mr[s][m][2];
            } else {
                constexpr size_t W = (size_t)kLanes * EPL;
                const double* tab = C.mtab_lane + (size_t)m * 4 * W + s;
                const bool on = elem_valid[s] && C.mtab_lane != nullptr;
                amp[s] = on ? C.mact_lane[(size_t)m * (size_t)P.n_envs * W + s] * tab[3 * W] : 0.0;
                mr[s][0] = on ? tab[0] : 0.0; mr[s][1] = on ? tab[W] : 0.0; mr[s][2] = on ? tab[2 * W] : 0.0;
            }
            live = live || (amp[s] != 0.0);
        }
        if (__builtin_amdgcn_ballot_w64(live) == 0ull) continue;      // the layer is off in this rod
        const bool radial = P.muscle_kind[m] == SOFTROD_MUSCLE_TRANSVERSE && P.muscle_tm_law == 0;
#pragma unroll
        for (int s = 0; s < EPL; ++s) {
            const double p0 = rad[s] * mr[s][0], p1 = rad[s] * mr[s][1], p2 = rad[s] * mr[s][2];
            const double n0 = sh[s][0] + (kav[s][1] * p2 - kav[s][2] * p1);
            const double n1 = sh[s][1] + (kav[s][2] * p0 - kav[s][0] * p2);
            const double n2 = sh[s][2] + (kav[s][0] * p1 - kav[s][1] * p0);
            double ss = fma(n2, n2, fma(n1, n1, n0 * n0));
            ss = elem_valid[s] ? ss : 1.0;
            double rn, nrm;
            if (FASTM) { rn = fast_rsqrt(ss); nrm = ss * rn; }
            else { nrm = sqrt(ss); rn = 1.0 / nrm; }
            double len = nrm;
            if (radial) len = FASTM ? fast_rsqrt(nrm) : 1.0 / sqrt(nrm);
            // fl(l) by Horner with COMPILE-TIME indices only: a run-time index into fl_coef[] forces the kernels that step on
            // a local, edited copy of RodParams (the rigid-body ones: `Pk`) to keep the WHOLE struct in scratch — 1592 B per
            // lane and a scratch load for every parameter the loop reads.  The cubic of the paper is the common case.
            double w;
            if (P.fl_degree == 3) {
                w = fma(fma(fma(P.fl_coef[3], len, P.fl_coef[2]), len, P.fl_coef[1]), len, P.fl_coef[0]);
            } else {
                w = 0.0;
#pragma unroll
                for (int p = SOFTROD_MAX_FL_COEF - 1; p >= 0; --p) w = (p <= P.fl_degree) ? fma(w, len, P.fl_coef[p]) : w;
            }
            w = (w < 0.0) ? 0.0 : w;
            const double Fm = amp[s] * w * rn;
            const double g0 = Fm * n0, g1 = Fm * n1, g2 = Fm * n2;       // F_m t_m
            fi[s][0] += g0; fi[s][1] += g1; fi[s][2] += g2;
            ce[s][0] += p1 * g2 - p2 * g1;
            ce[s][1] += p2 * g0 - p0 * g2;
            ce[s][2] += p0 * g1 - p1 * g0;
        }
    }
    const bool pyel = P.muscle_form == 1;
    // F_ext += D^h(Q^T f [/ e])
    double cs[EPL][3];
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        const double* Q = L.Q[s];
        const double sc = pyel ? 1.0 / e[s] : 1.0;
        const double a0 = fma(Q[6], fi[s][2], fma(Q[3], fi[s][1], Q[0] * fi[s][0])) * sc;
        const double a1 = fma(Q[7], fi[s][2], fma(Q[4], fi[s][1], Q[1] * fi[s][0])) * sc;
        const double a2 = fma(Q[8], fi[s][2], fma(Q[5], fi[s][1], Q[2] * fi[s][0])) * sc;
        cs[s][0] = elem_valid[s] ? a0 : 0.0;
        cs[s][1] = elem_valid[s] ? a1 : 0.0;
        cs[s][2] = elem_valid[s] ? a2 : 0.0;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double a[EPL], o[EPL];
#pragma unroll
        for (int s = 0; s < EPL; ++s) a[s] = cs[s][c];
        shift_prev<EPL>(a, o);
#pragma unroll
        for (int s = 0; s < EPL; ++s) f[s][c] += cs[s][c] - o[s];
    }
    // tau_ext += D^h(c_v) + A^h(kappa x c_v D^) + (e Q t) x f l^
    double cn[EPL][3], up[EPL][3], um[EPL][3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double a[EPL], o[EPL];
#pragma unroll
        for (int s = 0; s < EPL; ++s) a[s] = elem_valid[s] ? ce[s][c] : 0.0;
        shift_next<EPL>(a, o);
#pragma unroll
        for (int s = 0; s < EPL; ++s) cn[s][c] = o[s];
    }
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        const bool vor_valid = slot_local(P, lane * EPL + s) < n - 1;
        const double ef = pyel ? e3v[s] : 1.0;
        double cv[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) cv[c] = vor_valid ? 0.5 * (ce[s][c] + cn[s][c]) : 0.0;
        const double hd = 0.5 * P.rest_vor * ef;
        const double h3[3] = {(kv[s][1] * cv[2] - kv[s][2] * cv[1]) * hd, (kv[s][2] * cv[0] - kv[s][0] * cv[2]) * hd,
                              (kv[s][0] * cv[1] - kv[s][1] * cv[0]) * hd};
#pragma unroll
        for (int c = 0; c < 3; ++c) { up[s][c] = cv[c] * ef + h3[c]; um[s][c] = cv[c] * ef - h3[c]; }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double a[EPL], o[EPL];
#pragma unroll
        for (int s = 0; s < EPL; ++s) a[s] = um[s][c];
        shift_prev<EPL>(a, o);
#pragma unroll
        for (int s = 0; s < EPL; ++s) tq[s][c] += up[s][c] - o[s];
    }
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        const double g = (pyel ? 1.0 : e[s]) * P.rest_len;
        const double q0 = g * qt[s][0], q1 = g * qt[s][1], q2 = g * qt[s][2];
        const double t0 = q1 * fi[s][2] - q2 * fi[s][1], t1 = q2 * fi[s][0] - q0 * fi[s][2], t2 = q0 * fi[s][1] - q1 * fi[s][0];
        tq[s][0] += elem_valid[s] ? t0 : 0.0;
        tq[s][1] += elem_valid[s] ? t1 : 0.0;
        tq[s][2] += elem_valid[s] ? t2 : 0.0;
    }
}

// Per-lane muscle constants of this rod for one launch: the layers' position ratios, and activation x strength with
// the activation either what set_action just applied (ArmPush: uniform over the elements, EnvAction.mu) or the
// resident per-element rows (softrod_state_view.muscle_activation).  RESIDENT = false: only the two row pointers
// (set_action has already written the rows it changed; a lane reads back its own stores in program order).
template <unsigned F, int EPL, bool RESIDENT>
__device__ __forceinline__ void build_muscle_const(const RodParams& P, const StatePtrs& S, size_t N, int rod, int lane,
                                                   const EnvAction& A, ConstN<EPL>& C) {
    constexpr size_t W = (size_t)kLanes * EPL;
    C.mtab_lane = nullptr;
    C.mact_lane = nullptr;
    if (!has<F>(P, SOFTROD_FEAT_COOMM_MUSCLES)) return;
    if constexpr (!RESIDENT) {
        if (S.mtab != nullptr && S.mact != nullptr) {
            C.mtab_lane = S.mtab + (size_t)lane * EPL;
            C.mact_lane = S.mact + (size_t)rod * W + (size_t)lane * EPL;
        }
        return;
    }
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        const int raw = lane * EPL + s;
        const bool elem_valid = slot_local(P, raw) < P.n_elem;
#pragma unroll
        for (int m = 0; m < SOFTROD_MAX_MUSCLES; ++m) {
            const bool on = m < P.n_muscles && elem_valid && S.mtab != nullptr;
            const double* tab = S.mtab + (size_t)m * 4 * W + raw;
            const double act = !on ? 0.0 : (A.mu_set ? A.mu[m] : S.mact[((size_t)m * N + rod) * W + raw]);
            C.mr[s][m][0] = on ? tab[0 * W] : 0.0;
            C.mr[s][m][1] = on ? tab[1 * W] : 0.0;
            C.mr[s][m][2] = on ? tab[2 * W] : 0.0;
            C.amp[s][m] = on ? act * tab[3 * W] : 0.0;
        }
    }
}

}  // namespace softrod
