// softrod_long.hpp — the fast step kernel generalised to EPL nodes/elements per lane.
//
// One rod still lives in ONE wavefront, but lane k now owns nodes EPL*k .. EPL*k+EPL-1 (and
// the elements / Voronoi vertices of the same indices), so a wave holds rods of up to
// 64*EPL - 1 elements: EPL = 2 covers BASELINE config 3 ("OctoArmSingle-style, 100
// elements").  Neighbours inside a lane are plain registers; only the last slot's "next"
// and the first slot's "previous" cross lanes, so the number of DPP shifts per substep is
// the same as for EPL = 1 while the arithmetic per wave doubles — the stencil traffic per
// element halves.  Global rows are 64*EPL wide, index = node index (each lane loads EPL
// adjacent doubles: still fully coalesced).
//
// Mathematics, operator order and polynomial range reductions are those of
// softrod_fast.hpp (which stays the EPL = 1 production path); this file shares its helpers.
#pragma once

namespace softrod {

template <int EPL>
struct LaneN {
    double x[EPL][3], v[EPL][3];
    double Q[EPL][9], w[EPL][3];
    double t[EPL][3];
    double kap[EPL][3], rk[EPL][3];
};

template <int EPL>
struct ConstN {
    double hx[EPL], hq[EPL];
    double cf[EPL], ca[EPL][3];
    double cw01[EPL], cw2[EPL];
    double s01[EPL], s2[EPL];
    double b01[EPL], bd[EPL];
    double mass[EPL], mass_next[EPL];
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
            if (F == kRuntimeFeatures || (F & SOFTROD_FEAT_REST_KAPPA_ACTION)) {
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
            if (F == kRuntimeFeatures || (F & SOFTROD_FEAT_REST_KAPPA_ACTION))
                S.kap[c * N * W + base + s] = L.kap[s][c];
        }
#pragma unroll
        for (int c = 0; c < 9; ++c) S.dir[c * N * W + base + s] = L.Q[s][c];
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
        const int idx = lane * EPL + s;
        qn[s] = (idx >= 1 && idx <= n - 1) ? 0.25 : 0.0;
        qe[s] = (idx >= 1 && idx <= n - 2) ? 0.25 : 0.0;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double rv[EPL], rw[EPL];
#pragma unroll
        for (int s = 0; s < EPL; ++s) {
            const int idx = lane * EPL + s;
            rv[s] = (idx <= n) ? L.v[s][c] : 0.0;
            rw[s] = (idx < n) ? L.w[s][c] : 0.0;
        }
        laplace_filter_n<EPL>(rv, qn, P.filter_order);
        laplace_filter_n<EPL>(rw, qe, P.filter_order);
#pragma unroll
        for (int s = 0; s < EPL; ++s) {
            const int idx = lane * EPL + s;
            L.v[s][c] = (idx <= n) ? rv[s] : L.v[s][c];
            L.w[s][c] = (idx < n) ? rw[s] : L.w[s][c];
        }
    }
}

// ---- kinematic step, per slot ------------------------------------------------------------------
template <int EPL>
__device__ __forceinline__ void kinematic_n(double h, const ConstN<EPL>& C, LaneN<EPL>& L) {
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        const double hp = h * C.hx[s];
        L.x[s][0] = fma(hp, L.v[s][0], L.x[s][0]);
        L.x[s][1] = fma(hp, L.v[s][1], L.x[s][1]);
        L.x[s][2] = fma(hp, L.v[s][2], L.x[s][2]);
        const double hh = h * C.hq[s];
        const double a0 = hh * L.w[s][0], a1 = hh * L.w[s][1], a2 = hh * L.w[s][2];
        const double q0 = a0 * a0, q1 = a1 * a1, q2 = a2 * a2;
        double sc, cc;
        sinc_cosc(q0 + q1 + q2, sc, cc);
        const double s0 = sc * a0, s1 = sc * a1, s2 = sc * a2;
        const double ca0 = cc * a0, ca1 = cc * a1;
        const double c01 = ca0 * a1, c02 = ca0 * a2, c12 = ca1 * a2;
        const double R0 = fma(-cc, q1 + q2, 1.0), R4 = fma(-cc, q0 + q2, 1.0), R8 = fma(-cc, q0 + q1, 1.0);
        const double R1 = c01 + s2, R3 = c01 - s2;
        const double R2 = c02 - s1, R6 = c02 + s1;
        const double R5 = c12 + s0, R7 = c12 - s0;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double b0 = L.Q[s][j], b1 = L.Q[s][3 + j], b2 = L.Q[s][6 + j];
            L.Q[s][j] = fma(R2, b2, fma(R1, b1, R0 * b0));
            L.Q[s][3 + j] = fma(R5, b2, fma(R4, b1, R3 * b0));
            L.Q[s][6 + j] = fma(R8, b2, fma(R7, b1, R6 * b0));
        }
    }
}

// ---- plane contact with anisotropic friction (softrod_contact.hpp, slot-interleaved) ----------
template <int EPL>
__device__ __forceinline__ void plane_contact_n(const ContactParams& C, const RodParams& P, int lane,
                                                const ConstN<EPL>& K, const LaneN<EPL>& L,
                                                const double (&xn)[EPL][3], const double (&vn)[EPL][3],
                                                const double (&len)[EPL], const double (&F)[EPL][3],
                                                double (&tq)[EPL][3], double (&fc)[EPL][3]) {
    const int n = P.n_elem;
    const double* nr = C.normal;
    double E[EPL][3], ax[EPL][3], ro[EPL][3], arm[EPL][3], radius[EPL], nmag[EPL];
    double slip_ax[EPL], slip_ro[EPL];
    bool contact[EPL];
    double Fn[EPL][3];
    // node -> element average of the total force
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double a[EPL], o[EPL];
#pragma unroll
        for (int s = 0; s < EPL; ++s) a[s] = F[s][i];
        shift_next<EPL>(a, o);
#pragma unroll
        for (int s = 0; s < EPL; ++s) Fn[s][i] = o[s];
    }
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        const int idx = lane * EPL + s;
        const bool first = (idx == 0), last = (idx == n - 1);
        radius[s] = C.r0_sqrt_rest_len / sqrt(len[s]);
        double fel[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            fel[i] = 0.5 * (F[s][i] + Fn[s][i]);
            fel[i] += first ? 0.5 * F[s][i] : 0.0;
            fel[i] += last ? 0.5 * Fn[s][i] : 0.0;
        }
        const double fn = nr[0] * fel[0] + nr[1] * fel[1] + nr[2] * fel[2];
        double dist = 0.0, vel[3], vnrm = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double xe = 0.5 * (L.x[s][i] + xn[s][i]);
            dist += nr[i] * (xe - C.origin[i]);
            vel[i] = (K.mass_next[s] * vn[s][i] + K.mass[s] * L.v[s][i]) / (K.mass_next[s] + K.mass[s]);
            vnrm += nr[i] * vel[i];
        }
        const double pen = fmin(dist - radius[s], 0.0);
        contact[s] = (idx < n) && !((dist - radius[s]) > C.surface_tol);
        const double resp = (fn > 0.0) ? 0.0 : -fn;
        nmag[s] = contact[s] ? fabs(resp) : 0.0;
        const double ntot = resp + (-C.k * pen) + (-C.nu * vnrm);
#pragma unroll
        for (int i = 0; i < 3; ++i) E[s][i] = contact[s] ? nr[i] * ntot : 0.0;
        // kinetic friction
        const double tn = nr[0] * L.t[s][0] + nr[1] * L.t[s][1] + nr[2] * L.t[s][2];
#pragma unroll
        for (int i = 0; i < 3; ++i) ax[s][i] = L.t[s][i] - nr[i] * tn;
        const double tpm = sqrt(ax[s][0] * ax[s][0] + ax[s][1] * ax[s][1] + ax[s][2] * ax[s][2]);
        const double itp = 1.0 / (tpm + 1e-14);
#pragma unroll
        for (int i = 0; i < 3; ++i) ax[s][i] *= itp;
        ro[s][0] = ax[s][1] * nr[2] - ax[s][2] * nr[1];
        ro[s][1] = ax[s][2] * nr[0] - ax[s][0] * nr[2];
        ro[s][2] = ax[s][0] * nr[1] - ax[s][1] * nr[0];
#pragma unroll
        for (int i = 0; i < 3; ++i) arm[s][i] = -nr[i] * radius[s];
        const double vax = vel[0] * ax[s][0] + vel[1] * ax[s][1] + vel[2] * ax[s][2];
        const double axn = sqrt(ax[s][0] * ax[s][0] + ax[s][1] * ax[s][1] + ax[s][2] * ax[s][2]);
        const double sgn = sign_of(vax);
        const double kmu = 0.5 * (C.kin_mu[0] * (1 + sgn) + C.kin_mu[1] * (1 - sgn));
        slip_ax[s] = slip_function(fabs(vax) * axn, C.slip_tol);
        const double vroll = vel[0] * ro[s][0] + vel[1] * ro[s][1] + vel[2] * ro[s][2];
        const double* Q = L.Q[s];
        const double* w = L.w[s];
        double qa[3], rot[3];
#pragma unroll
        for (int i = 0; i < 3; ++i)
            qa[i] = Q[3 * i] * arm[s][0] + Q[3 * i + 1] * arm[s][1] + Q[3 * i + 2] * arm[s][2];
        const double wq[3] = {w[1] * qa[2] - w[2] * qa[1], w[2] * qa[0] - w[0] * qa[2],
                              w[0] * qa[1] - w[1] * qa[0]};
#pragma unroll
        for (int i = 0; i < 3; ++i) rot[i] = Q[i] * wq[0] + Q[3 + i] * wq[1] + Q[6 + i] * wq[2];
        const double vrot = rot[0] * ro[s][0] + rot[1] * ro[s][1] + rot[2] * ro[s][2];
        const double sroll = vroll + vrot;
        const double ron = sqrt(ro[s][0] * ro[s][0] + ro[s][1] * ro[s][1] + ro[s][2] * ro[s][2]);
        slip_ro[s] = slip_function(fabs(sroll) * ron, C.slip_tol);
        const double vm = sqrt(vel[0] * vel[0] + vel[1] * vel[1] + vel[2] * vel[2]) + 1e-14;
        const double uax = (vel[0] / vm) * ax[s][0] + (vel[1] / vm) * ax[s][1] + (vel[2] / vm) * ax[s][2];
        const double uro = (vel[0] / vm) * ro[s][0] + (vel[1] / vm) * ro[s][1] + (vel[2] / vm) * ro[s][2];
        const double ka = contact[s] ? -((1.0 - slip_ax[s]) * kmu * nmag[s] * uax) : 0.0;
        const double kr = contact[s] ? -((1.0 - slip_ro[s]) * C.kin_mu[2] * nmag[s] * uro) : 0.0;
        double fr[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            fr[i] = kr * ro[s][i];
            E[s][i] += ka * ax[s][i] + fr[i];
        }
        const double cr[3] = {arm[s][1] * fr[2] - arm[s][2] * fr[1], arm[s][2] * fr[0] - arm[s][0] * fr[2],
                              arm[s][0] * fr[1] - arm[s][1] * fr[0]};
#pragma unroll
        for (int i = 0; i < 3; ++i)
            tq[s][i] += Q[3 * i] * cr[0] + Q[3 * i + 1] * cr[1] + Q[3 * i + 2] * cr[2];
    }
    // scatter round 1 and the updated nodal totals
    double F2[EPL][3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double a[EPL], o[EPL];
#pragma unroll
        for (int s = 0; s < EPL; ++s) a[s] = E[s][i];
        shift_prev<EPL>(a, o);
#pragma unroll
        for (int s = 0; s < EPL; ++s) {
            fc[s][i] = 0.5 * E[s][i] + 0.5 * o[s];
            F2[s][i] = F[s][i] + fc[s][i];
        }
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double a[EPL], o[EPL];
#pragma unroll
        for (int s = 0; s < EPL; ++s) a[s] = F2[s][i];
        shift_next<EPL>(a, o);
#pragma unroll
        for (int s = 0; s < EPL; ++s) Fn[s][i] = o[s];
    }
    // static friction
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        const int idx = lane * EPL + s;
        const bool first = (idx == 0), last = (idx == n - 1);
        double fel[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            fel[i] = 0.5 * (F2[s][i] + Fn[s][i]);
            fel[i] += first ? 0.5 * F2[s][i] : 0.0;
            fel[i] += last ? 0.5 * Fn[s][i] : 0.0;
        }
        const double fax = fel[0] * ax[s][0] + fel[1] * ax[s][1] + fel[2] * ax[s][2];
        const double sg = sign_of(fax);
        const double smu = 0.5 * (C.stat_mu[0] * (1 + sg) + C.stat_mu[1] * (1 - sg));
        const double sa = contact[s] ? -(fmin(fabs(fax), slip_ax[s] * smu * nmag[s]) * sg) : 0.0;
        const double* Q = L.Q[s];
        double tt[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) tt[i] = Q[i] * tq[s][0] + Q[3 + i] * tq[s][1] + Q[6 + i] * tq[s][2];
        const double tax = tt[0] * ax[s][0] + tt[1] * ax[s][1] + tt[2] * ax[s][2];
        const double fro = fel[0] * ro[s][0] + fel[1] * ro[s][1] + fel[2] * ro[s][2];
        const double noslip = -((radius[s] * fro - 2.0 * tax) / 3.0 / radius[s]);
        const double sr = contact[s]
            ? fmin(fabs(noslip), slip_ro[s] * C.stat_mu[2] * nmag[s]) * sign_of(noslip) : 0.0;
        double fr[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            fr[i] = sr * ro[s][i];
            E[s][i] = sa * ax[s][i] + fr[i];
        }
        const double cr[3] = {arm[s][1] * fr[2] - arm[s][2] * fr[1], arm[s][2] * fr[0] - arm[s][0] * fr[2],
                              arm[s][0] * fr[1] - arm[s][1] * fr[0]};
#pragma unroll
        for (int i = 0; i < 3; ++i)
            tq[s][i] += Q[3 * i] * cr[0] + Q[3 * i + 1] * cr[1] + Q[3 * i + 2] * cr[2];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        double a[EPL], o[EPL];
#pragma unroll
        for (int s = 0; s < EPL; ++s) a[s] = E[s][i];
        shift_prev<EPL>(a, o);
#pragma unroll
        for (int s = 0; s < EPL; ++s) fc[s][i] += 0.5 * E[s][i] + 0.5 * o[s];
    }
}

// ---- forces, torques, rate update, dampers, constrain_rates -------------------------------------
template <unsigned F, int EPL>
__device__ __forceinline__ void dynamic_n(const RodParams& P, const ConstN<EPL>& C, const BcTargets& B,
                                          int lane, LaneN<EPL>& L) {
    const int n = P.n_elem;
    double xn[EPL][3], vn[EPL][3], d[EPL][3];
    double len[EPL], il[EPL], e[EPL], ie[EPL];
    double qt[EPL][3], np[EPL][3], cs[EPL][3], f[EPL][3], tq[EPL][3];

    // next-node position / velocity per slot
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double a[EPL], o[EPL], av[EPL], ov[EPL];
#pragma unroll
        for (int s = 0; s < EPL; ++s) { a[s] = L.x[s][c]; av[s] = L.v[s][c]; }
        shift_next<EPL>(a, o);
        shift_next<EPL>(av, ov);
#pragma unroll
        for (int s = 0; s < EPL; ++s) { xn[s][c] = o[s]; vn[s][c] = ov[s]; }
    }
    // geometry, shear/stretch stress in the lab frame
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        const bool elem_valid = (lane * EPL + s) < n;
#pragma unroll
        for (int c = 0; c < 3; ++c) d[s][c] = xn[s][c] - L.x[s][c];
        double dd = fma(d[s][2], d[s][2], fma(d[s][1], d[s][1], d[s][0] * d[s][0]));
        dd = elem_valid ? dd : 1.0;
        const double r = fast_rsqrt(dd);
        len[s] = fma(dd, r, P.eps_length);
        il[s] = fma(-P.eps_length * r, r, r);
#pragma unroll
        for (int c = 0; c < 3; ++c) L.t[s][c] = d[s][c] * il[s];
        e[s] = len[s] * P.inv_rest_len;
        ie[s] = P.rest_len * il[s];
        const double* Q = L.Q[s];
        const double* t = L.t[s];
        qt[s][0] = fma(Q[2], t[2], fma(Q[1], t[1], Q[0] * t[0]));
        qt[s][1] = fma(Q[5], t[2], fma(Q[4], t[1], Q[3] * t[0]));
        qt[s][2] = fma(Q[8], t[2], fma(Q[7], t[1], Q[6] * t[0]));
        np[s][0] = C.s01[s] * qt[s][0];
        np[s][1] = C.s01[s] * qt[s][1];
        np[s][2] = C.s2[s] * (qt[s][2] - ie[s]);
        cs[s][0] = fma(Q[6], np[s][2], fma(Q[3], np[s][1], Q[0] * np[s][0]));
        cs[s][1] = fma(Q[7], np[s][2], fma(Q[4], np[s][1], Q[1] * np[s][0]));
        cs[s][2] = fma(Q[8], np[s][2], fma(Q[5], np[s][1], Q[2] * np[s][0]));
    }
    // nodal internal force: difference of the lab-frame stress
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double a[EPL], o[EPL];
#pragma unroll
        for (int s = 0; s < EPL; ++s) a[s] = cs[s][c];
        shift_prev<EPL>(a, o);
#pragma unroll
        for (int s = 0; s < EPL; ++s) f[s][c] = cs[s][c] - o[s];
    }
    // bend/twist on the Voronoi vertices
    double Qn[EPL][9], len_n[EPL];
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        double a[EPL], o[EPL];
#pragma unroll
        for (int s = 0; s < EPL; ++s) a[s] = L.Q[s][i];
        shift_next<EPL>(a, o);
#pragma unroll
        for (int s = 0; s < EPL; ++s) Qn[s][i] = o[s];
    }
    shift_next<EPL>(len, len_n);
    double up[EPL][3], um[EPL][3];
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        const bool vor_valid = (lane * EPL + s) < n - 1;
        const double* Q = L.Q[s];
        const double* N_ = Qn[s];
#define SR_RD(i, j) fma(N_[3 * (i) + 2], Q[3 * (j) + 2], fma(N_[3 * (i) + 1], Q[3 * (j) + 1], \
                        N_[3 * (i)] * Q[3 * (j)]))
#define SR_RD_SUB(i, j, acc) fma(-N_[3 * (i) + 2], Q[3 * (j) + 2], fma(-N_[3 * (i) + 1], \
                        Q[3 * (j) + 1], fma(-N_[3 * (i)], Q[3 * (j)], acc)))
        const double vec0 = SR_RD_SUB(1, 2, SR_RD(2, 1));
        const double vec1 = SR_RD_SUB(2, 0, SR_RD(0, 2));
        const double vec2 = SR_RD_SUB(0, 1, SR_RD(1, 0));
        const double trace = SR_RD(0, 0) + SR_RD(1, 1) + SR_RD(2, 2);
#undef SR_RD
#undef SR_RD_SUB
        const double y = fma(-0.25, trace, 0.75 + 0.5 * P.acos_shift);
        const double gk = theta_over_sin(y, vor_valid) * (-0.5 * P.inv_rest_vor);
        const double k0 = vec0 * gk, k1 = vec1 * gk, k2 = vec2 * gk;
        const double vd = (len_n[s] + len[s]) * (0.5 * P.inv_rest_vor);
        const double rvd = fast_rcp(vd);
        const double e3 = rvd * rvd * rvd;
        if (F == kRuntimeFeatures || (F & SOFTROD_FEAT_REST_KAPPA_ACTION)) {
            L.kap[s][0] = k0; L.kap[s][1] = k1; L.kap[s][2] = k2;
        }
        if (has<F>(P, SOFTROD_FEAT_REST_KAPPA_ACTION)) {
            const double m0 = C.b01[s] * (k0 - L.rk[s][0]), m1 = C.b01[s] * (k1 - L.rk[s][1]),
                         m2 = (C.b01[s] + C.bd[s]) * (k2 - L.rk[s][2]);
            const double hd = 0.5 * P.rest_vor * e3;
            const double c2[3] = {m0 * e3, m1 * e3, m2 * e3};
            const double h3[3] = {(k1 * m2 - k2 * m1) * hd, (k2 * m0 - k0 * m2) * hd, (k0 * m1 - k1 * m0) * hd};
#pragma unroll
            for (int c = 0; c < 3; ++c) { up[s][c] = c2[c] + h3[c]; um[s][c] = c2[c] - h3[c]; }
        } else {
            const double c20 = C.b01[s] * k0 * e3, c21 = C.b01[s] * k1 * e3,
                         c22 = (C.b01[s] + C.bd[s]) * k2 * e3;
            const double hz = 0.5 * P.rest_vor * C.bd[s] * k2 * e3;
            const double h30 = k1 * hz, h31 = -k0 * hz;
            up[s][0] = c20 + h30; um[s][0] = c20 - h30;
            up[s][1] = c21 + h31; um[s][1] = c21 - h31;
            up[s][2] = c22;       um[s][2] = c22;
        }
    }
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double a[EPL], o[EPL];
#pragma unroll
        for (int s = 0; s < EPL; ++s) a[s] = um[s][c];
        shift_prev<EPL>(a, o);
#pragma unroll
        for (int s = 0; s < EPL; ++s) tq[s][c] = up[s][c] - o[s];
    }
    // shear couple, transport, unsteady dilatation
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        const double* w = L.w[s];
        tq[s][0] = fma(len[s], fma(qt[s][1], np[s][2], -qt[s][2] * np[s][1]), tq[s][0]);
        tq[s][1] = fma(len[s], fma(qt[s][2], np[s][0], -qt[s][0] * np[s][2]), tq[s][1]);
        tq[s][2] = fma(len[s], fma(qt[s][0], np[s][1], -qt[s][1] * np[s][0]), tq[s][2]);
        const double num = fma(d[s][2], vn[s][2] - L.v[s][2],
                               fma(d[s][1], vn[s][1] - L.v[s][1], d[s][0] * (vn[s][0] - L.v[s][0])));
        const double sdil = num * il[s] * il[s];
        const double j01 = P.J[0] * ie[s], j2 = P.J[2] * ie[s];
        const double z = w[2] * (j01 - j2);
        tq[s][0] = fma(w[1], z, tq[s][0]);
        tq[s][1] = fma(-w[0], z, tq[s][1]);
        const double js01 = j01 * sdil, js2 = j2 * sdil;
        tq[s][0] = fma(js01, w[0], tq[s][0]);
        tq[s][1] = fma(js01, w[1], tq[s][1]);
        tq[s][2] = fma(js2, w[2], tq[s][2]);
    }
    // plane contact
    if (has<F>(P, SOFTROD_FEAT_PLANE_CONTACT_ANISO)) {
        double Fg[EPL][3], fc[EPL][3];
#pragma unroll
        for (int s = 0; s < EPL; ++s) {
            const bool node_valid = (lane * EPL + s) <= n;
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                Fg[s][c] = f[s][c];
                if (has<F>(P, SOFTROD_FEAT_GRAVITY) && !P.contact_before_forcing)
                    Fg[s][c] += node_valid ? P.gravity[c] * C.mass[s] : 0.0;
            }
        }
        plane_contact_n<EPL>(contact_params(P), P, lane, C, L, xn, vn, len, Fg, tq, fc);
#pragma unroll
        for (int s = 0; s < EPL; ++s)
#pragma unroll
            for (int c = 0; c < 3; ++c) f[s][c] += fc[s][c];
    }
    // rate update fused with the analytical damper
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        const bool elem_valid = (lane * EPL + s) < n;
#pragma unroll
        for (int c = 0; c < 3; ++c) L.v[s][c] = fma(P.damp_t, L.v[s][c], fma(C.cf[s], f[s][c], C.ca[s][c]));
        const double ce01 = C.cw01[s] * e[s], ce2 = C.cw2[s] * e[s];
        double w0 = fma(ce01, tq[s][0], L.w[s][0]), w1 = fma(ce01, tq[s][1], L.w[s][1]),
               w2 = fma(ce2, tq[s][2], L.w[s][2]);
        if (has<F>(P, SOFTROD_FEAT_ANALYTICAL_DAMPER)) {
            double ex0, ex2;
            exp_pair(e[s] * P.damp_logr[0], e[s] * P.damp_logr[2], elem_valid, ex0, ex2);
            w0 *= ex0; w1 *= ex0; w2 *= ex2;
        }
        L.w[s][0] = w0; L.w[s][1] = w1; L.w[s][2] = w2;
    }
    if (P.damp_before_constrain) {
        if (has<F>(P, SOFTROD_FEAT_LAPLACE_FILTER)) laplace_filter_rates_n<EPL>(P, lane, L);
        constrain_rates_n<F, EPL>(P, B, lane, L);
    } else {
        BcTargets Bs = B;
        Bs.vel[0] *= P.damp_t; Bs.vel[1] *= P.damp_t; Bs.vel[2] *= P.damp_t;
        constrain_rates_n<F, EPL>(P, Bs, lane, L);
        if (has<F>(P, SOFTROD_FEAT_LAPLACE_FILTER)) laplace_filter_rates_n<EPL>(P, lane, L);
    }
}

// ---- env prologue / epilogue on the slot-interleaved state --------------------------------------
template <int EPL>
__device__ __forceinline__ void sum_tangents(const RodParams& P, int lane, const LaneN<EPL>& L, double tm[3]) {
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double p = 0.0;
#pragma unroll
        for (int s = 0; s < EPL; ++s) p += ((lane * EPL + s) < P.n_elem) ? L.t[s][c] : 0.0;
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
        for (int s = 0; s < EPL; ++s) p += ((lane * EPL + s) <= P.n_elem) ? C.mass[s] * L.x[s][c] : 0.0;
        com[c] = wave_sum(p) / P.mass_total;
    }
}

template <int EPL>
__device__ __forceinline__ void arm_get_state_n(const RodParams& P, const StatePtrs& S, size_t N, int rod,
                                                int lane, const ConstN<EPL>& C, const LaneN<EPL>& L,
                                                const float* pa, float* __restrict__ obs) {
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
        float* o = obs + 25 * (size_t)rod;
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

template <int E, int EPL>
__device__ __forceinline__ void env_observe_n(const RodParams& P, const StatePtrs& S, size_t N, int rod,
                                              int lane, const ConstN<EPL>& C, const LaneN<EPL>& L,
                                              const float* pa, float* __restrict__ obs) {
    const int env = env_of<E>(P);
    if (env == SOFTROD_ENV_SOFTPENDULUM3D) {
        const double tilt = tilt_n<EPL>(P, lane, L);
        if (lane == 0) {
            float* o = obs + 9 * (size_t)rod;
            o[0] = (float)L.x[0][0]; o[1] = (float)L.x[0][1]; o[2] = (float)L.x[0][2];
            o[3] = (float)L.v[0][0]; o[4] = (float)L.v[0][1]; o[5] = (float)L.v[0][2];
            o[6] = pa[0]; o[7] = pa[1];
            o[8] = (float)tilt;
        }
    } else if (env == SOFTROD_ENV_ARM_SINGLE) {
        arm_get_state_n<EPL>(P, S, N, rod, lane, C, L, pa, obs);
    } else {
        const double th = theta_n<EPL>(P, lane, L);
        if (lane == 0) {
            float* o = obs + 4 * (size_t)rod;
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
                                               double* __restrict__ aux) {
    bool bad = false;
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        bool b = false;
#pragma unroll
        for (int c = 0; c < 3; ++c) b = b || isnan(L.x[s][c]) || isnan(L.v[s][c]);
        bad = bad || (((lane * EPL + s) <= P.n_elem) && b);
    }
    const bool invalid = __any(bad);
    const int env = env_of<E>(P);
    if (env == SOFTROD_ENV_SOFTPENDULUM3D) {
        const double tilt = tilt_n<EPL>(P, lane, L);
        if (lane == 0) {
            const double bx = S.ctrl[(size_t)0 * N + rod], by = S.ctrl[(size_t)1 * N + rod];
            const double base_distance = sqrt(bx * bx + by * by);
            const float ctl = 1e-3f * (A.a[0] * A.a[0] + A.a[1] * A.a[1]);
            double r = -(tilt * tilt + 0.1 * (base_distance * base_distance) + (double)ctl);
            if (invalid) r = -50.0;
            reward[rod] = r;
            terminated[rod] = invalid ? 1 : 0;
            truncated[rod] = (time >= P.final_time) ? 1 : 0;
            if (aux) aux[rod] = tilt;
            float* o = obs + 9 * (size_t)rod;
            o[0] = (float)L.x[0][0]; o[1] = (float)L.x[0][1]; o[2] = (float)L.x[0][2];
            o[3] = (float)L.v[0][0]; o[4] = (float)L.v[0][1]; o[5] = (float)L.v[0][2];
            o[6] = A.a[0]; o[7] = A.a[1];
            o[8] = (float)tilt;
        }
    } else if (env == SOFTROD_ENV_ARM_SINGLE) {
        double pw = 0.0;
#pragma unroll
        for (int s = 0; s < EPL; ++s)
            pw += ((lane * EPL + s) < P.n_elem)
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
            reward[rod] = forward - (double)pen + survive;
            terminated[rod] = term ? 1 : 0;
            truncated[rod] = (time > P.final_time) ? 1 : 0;
        }
        arm_get_state_n<EPL>(P, S, N, rod, lane, C, L, A.a, obs);
    } else {
        const double th = theta_n<EPL>(P, lane, L);
        if (lane == 0) {
            double forward = 0.0, survive = 0.0;
            if (invalid) survive = -50.0;
            else forward = fabs(L.x[0][0]) * 10.0 + th * th;
            reward[rod] = forward - 0.0 + survive;
            terminated[rod] = invalid ? 1 : 0;
            truncated[rod] = (time > P.final_time) ? 1 : 0;
            float* o = obs + 4 * (size_t)rod;
            o[0] = (float)L.x[0][0];
            o[1] = (float)L.v[0][0];
            o[2] = A.a[0];
            o[3] = (float)th;
        }
    }
}

template <unsigned F, int EPL>
__device__ __forceinline__ void build_const(const RodParams& P, int lane, const EnvAction& A,
                                            ConstN<EPL>& C) {
    const int n = P.n_elem;
    const bool damp = has<F>(P, SOFTROD_FEAT_ANALYTICAL_DAMPER);
    const double ct = damp ? P.damp_t : 1.0;
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        const int idx = lane * EPL + s;
        const bool first = (idx == 0);
        const bool held_q = first && has<F>(P, SOFTROD_FEAT_PENDULUM_BC | SOFTROD_FEAT_FIXED_BC |
                                               SOFTROD_FEAT_MOVING_BASE_BC);
        const bool held_x = first && has<F>(P, SOFTROD_FEAT_FIXED_BC | SOFTROD_FEAT_MOVING_BASE_BC);
        const bool node_valid = idx <= n, elem_valid = idx < n, vor_valid = idx < n - 1;
        const double mass = (idx == 0 || idx == n) ? 0.5 * P.mass_node : P.mass_node;
        C.mass[s] = mass;
        C.mass_next[s] = (idx + 1 == n) ? 0.5 * P.mass_node : P.mass_node;
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
        C.cw01[s] = elem_valid ? P.dt * P.invJ[0] : 0.0;
        C.cw2[s] = elem_valid ? P.dt * P.invJ[2] : 0.0;
        C.s01[s] = elem_valid ? P.shear[0] : 0.0;
        C.s2[s] = elem_valid ? P.shear[2] : 0.0;
        C.b01[s] = vor_valid ? P.bend[0] : 0.0;
        C.bd[s] = vor_valid ? P.bend[2] - P.bend[0] : 0.0;
    }
}

template <unsigned F, int E, int EPL>
__device__ __forceinline__ void set_action_n(const RodParams& P, const StatePtrs& S, size_t N, int rod,
                                             int lane, const float* __restrict__ actions, EnvAction& A,
                                             BcTargets& B, LaneN<EPL>& L) {
    constexpr size_t W = (size_t)kLanes * EPL;
#pragma unroll
    for (int i = 0; i < 7; ++i) A.a[i] = 0.0f;
    A.force = 0.0;
    const int env = env_of<E>(P);
    if (env == SOFTROD_ENV_SOFTPENDULUM3D) {
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
    } else {
        if (actions) A.a[0] = actions[rod];
        A.force = (double)A.a[0];
    }
}

// ---- kernels --------------------------------------------------------------------------------------
template <unsigned F, int E, int EPL>
__global__ void __launch_bounds__(kLanes, (EPL > 1 ? 1 : 2))
softrod_step_long_kernel(const RodParams P, const StatePtrs S, const float* __restrict__ actions,
                         float* __restrict__ obs, double* __restrict__ reward,
                         uint8_t* __restrict__ terminated, uint8_t* __restrict__ truncated,
                         double* __restrict__ aux, const int n_sub, const int epilogue) {
    const int rod = blockIdx.x;
    const int lane = threadIdx.x;
    const size_t N = (size_t)P.n_envs;

    LaneN<EPL> L;
    load_lane<EPL, F>(S, N, rod, lane, L);
    BcTargets B;
    load_bc(S, N, rod, B);
    EnvAction A;
    set_action_n<F, E, EPL>(P, S, N, rod, lane, actions, A, B, L);
    {
        BcTargets B0 = B;
        if (has<F>(P, SOFTROD_FEAT_MOVING_BASE_BC)) {
            B0.vel[0] = __shfl(L.v[0][0], 0); B0.vel[1] = __shfl(L.v[0][1], 0); B0.vel[2] = __shfl(L.v[0][2], 0);
        }
        constrain_rates_n<F, EPL>(P, B0, lane, L);
        constrain_values_n<F, EPL>(P, B, lane, L);
    }
    double time = S.time[rod];
    ConstN<EPL> C;
    build_const<F, EPL>(P, lane, A, C);
    RodParams Pk = P;
    if (!has<F>(P, SOFTROD_FEAT_ANALYTICAL_DAMPER)) Pk.damp_t = 1.0;

    if (n_sub > 0) {
        kinematic_n<EPL>(P.half_dt, C, L);
        if (P.time_two_half_adds) time += P.half_dt;
        for (int s = 0; s < n_sub; ++s) {
            dynamic_n<F, EPL>(Pk, C, B, lane, L);
            const bool last = (s == n_sub - 1);
            kinematic_n<EPL>(last ? P.half_dt : P.dt, C, L);
            time += P.time_two_half_adds ? P.half_dt : P.dt;
            if (!last && P.time_two_half_adds) time += P.half_dt;
        }
    }
    store_lane<EPL, F>(S, N, rod, lane, L);
    if (lane == 0) S.time[rod] = time;
    if (epilogue)
        env_epilogue_n<E, EPL>(P, S, N, rod, lane, C, L, time, A, obs, reward, terminated, truncated, aux);
}

template <int EPL>
__global__ void __launch_bounds__(kLanes)
softrod_observe_long_kernel(const RodParams P, const StatePtrs S, const float* __restrict__ prev_action,
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
    ConstN<EPL> C;
    build_const<kRuntimeFeatures, EPL>(P, lane, A, C);
    const int adim = (P.env_kind == SOFTROD_ENV_SOFTPENDULUM3D) ? 2
                   : (P.env_kind == SOFTROD_ENV_ARM_SINGLE) ? 7 : 1;
    float pa[7] = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
    if (prev_action)
        for (int i = 0; i < adim; ++i) pa[i] = prev_action[adim * (size_t)rod + i];
    env_observe_n<kRuntimeEnv, EPL>(P, S, N, rod, lane, C, L, pa, obs);
}

template <int EPL>
__global__ void __launch_bounds__(kLanes)
softrod_reset_long_kernel(const RodParams P, const StatePtrs S, const ResetArgs A) {
    constexpr size_t W = (size_t)kLanes * EPL;
    const int rod = blockIdx.x;
    const int lane = threadIdx.x;
    if (A.mask && !A.mask[rod]) return;
    const size_t N = (size_t)P.n_envs;
    const double* in = A.init + (size_t)rod * 18;
    const int n = P.n_elem;
    LaneN<EPL> L;
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
        build_const<kRuntimeFeatures, EPL>(P, lane, A0, C);
        com_xy_n<EPL>(P, C, lane, L, com);
    }
    if (lane == 0) {
        S.time[rod] = 0.0;
#pragma unroll
        for (int i = 0; i < 3; ++i) S.bc[(size_t)i * N + rod] = in[i];
#pragma unroll
        for (int i = 0; i < 9; ++i) S.bc[(size_t)(3 + i) * N + rod] = in[9 + i];
#pragma unroll
        for (int i = 0; i < 4; ++i) S.ctrl[(size_t)i * N + rod] = (i < 2) ? com[i] : 0.0;
    }
}

}  // namespace softrod
