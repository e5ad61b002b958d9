// softrod_contact.hpp — rod-plane contact with anisotropic friction, one element per lane.
//
// Restates PyElastica's RodPlaneContactWithAnisotropicFriction.apply_contact
// (elastica/contact_forces.py -> _contact_functions.py: the plane normal-force kernel of
// Gazzola et al. 2018 eq. 4.8 and `anisotropic_friction`; recalled, not on disk) as
// registered by build_arm (gym_softrobot/envs/octopus/build.py:236-283).  Mirrors
// plane_contact() of oracle/softrod_oracle.c operation for operation.
//
// Lane k holds node k and element k.  node->element averages pull node k+1 with one
// wave_shl DPP shift, element->node scatters pull element k-1 with one wave_shr shift;
// the two rounds (normal + kinetic friction, then static friction on the updated
// totals) cost 12 fp64 values across lanes per substep.
//
// Two compile-time switches keep the fp64 instruction count down (the kernel is bound by
// it, profiles/README.md):
//   ZUP  the plane normal is exactly e_z — what both reference builds use
//        (octopus/build.py:233) — so every product with the normal's zero components,
//        and the z component of the axial / rolling directions, drops out;
//   FM   SOFTROD_MATH_FAST: each division is one Newton-refined v_rcp_f64 / v_rsq_f64
//        (<= 1 ulp) instead of the IEEE division / sqrt sequence.
#pragma once

namespace softrod {

struct ContactParams {
    double origin[3], normal[3];
    double k, nu, slip_tol, surface_tol;
    double kin_mu[3], stat_mu[3];   // forward, backward, sideways
    double kin_am[2], stat_am[2];   // (forward + backward) / 2, (forward - backward) / 2
    double r0_sqrt_rest_len;        // r0 * sqrt(l_rest): radius = this / sqrt(len)
    double inv_r0_sqrt_rest_len;
};

__device__ __forceinline__ double sign_of(double x) { return (double)((x > 0.0) - (x < 0.0)); }

template <bool FM>
__device__ __forceinline__ double inv_of(double x) {
    if constexpr (FM) return fast_rcp(x);
    else return 1.0 / x;
}
template <bool FM>
__device__ __forceinline__ double sqrt_of(double x) {   // x >= 0
    if constexpr (FM) return x * fast_rsqrt(fmax(x, 1.0e-300));      // 0 at 0 without a branch around the seed
    else return sqrt(x);
}

// find_slipping_elements on a vector of magnitude |a|; inv_thr = 1 / slip_velocity_tol
template <bool FM>
__device__ __forceinline__ double slip_function(double a, double thr, double inv_thr) {
    const double q = FM ? a * inv_thr : a / thr;
    const double m = fmin(q - 1.0, 1.0);
    return (fabs(a) > thr) ? fabs(1.0 - m) : 1.0;
}

// 1 - slip_function for a >= 0 as one clamp: min(max(a / thr - 1, 0), 1)  (a NaN gives 0 here and 1
// there, like the comparison in slip_function)
__device__ __forceinline__ double slip_excess(double a, double inv_thr) {
    return fmin(fmax(fma(a, inv_thr, -1.0), 0.0), 1.0);
}

// In:  F[s][3]   nodal internal + external force of node EPL*lane+s (so far)
//      tq[s][3]  element internal + external torque (local frame) (so far)
// Out: fc[s][3]  contact force added to that node;  tq += contact torques.
template <int EPL, bool ZUP, bool FM, bool TAPER = false>
__device__ __forceinline__ void plane_contact_n(const ContactParams& C, const RodParams& P, int lane,
                                                const ConstN<EPL>& K, const LaneN<EPL>& L,
                                                const double (&xn)[EPL][3], const double (&vn)[EPL][3],
                                                const double (&len)[EPL], const double (&F)[EPL][3],
                                                double (&tq)[EPL][3], double (&fc)[EPL][3]) {
    const int n = P.n_elem;
    const double* nr = C.normal;
    constexpr int D = ZUP ? 2 : 3;          // non-zero components of the in-plane directions
    double E[EPL][3], ax[EPL][3], ro[EPL][3], radius[EPL], nmag[EPL];
    double slip_ax[EPL], slip_ro[EPL];
    bool contact[EPL];
    double Fn[EPL][3];
    const double inv_slip = inv_of<FM>(C.slip_tol);
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
    double wa[EPL], wb[EPL], inv_radius[EPL];
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        const int idx = slot_local(P, lane * EPL + s);
        const bool first = (idx == 0), last = (idx == n - 1);
        // node -> element: half of each end node, the rod's two end nodes in full
        wa[s] = first ? 1.0 : 0.5;
        wb[s] = last ? 1.0 : 0.5;
        double fel[3];
        if constexpr (FM) {
            const double rsl = fast_rsqrt(len[s]);
            radius[s] = (TAPER ? K.r0s[s] : C.r0_sqrt_rest_len) * rsl;
            inv_radius[s] = (len[s] * rsl) * (TAPER ? K.ir0s[s] : C.inv_r0_sqrt_rest_len);   // sqrt(len) / (r0 sqrt(l_rest))
#pragma unroll
            for (int i = 0; i < 3; ++i) fel[i] = fma(wb[s], Fn[s][i], wa[s] * F[s][i]);
        } else {
            radius[s] = (TAPER ? K.r0s[s] : C.r0_sqrt_rest_len) / sqrt(len[s]);
            inv_radius[s] = 1.0 / radius[s];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                fel[i] = 0.5 * (F[s][i] + Fn[s][i]);
                fel[i] += first ? 0.5 * F[s][i] : 0.0;
                fel[i] += last ? 0.5 * Fn[s][i] : 0.0;
            }
        }
        const double inv_m = FM ? K.inv_mass_pair[s] : 1.0 / (K.mass_next[s] + K.mass[s]);
        double vel[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const double num = K.mass_next[s] * vn[s][i] + K.mass[s] * L.v[s][i];
            vel[i] = FM ? num * inv_m : num / (K.mass_next[s] + K.mass[s]);
        }
        double fn, dist, vnrm;
        if constexpr (ZUP) {
            fn = fel[2];
            dist = 0.5 * (L.x[s][2] + xn[s][2]) - C.origin[2];
            vnrm = vel[2];
        } else {
            fn = nr[0] * fel[0] + nr[1] * fel[1] + nr[2] * fel[2];
            dist = 0.0; vnrm = 0.0;
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                const double xe = 0.5 * (L.x[s][i] + xn[s][i]);
                dist += nr[i] * (xe - C.origin[i]);
                vnrm += nr[i] * vel[i];
            }
        }
        const double pen = fmin(dist - radius[s], 0.0);
        contact[s] = (idx < n) && !((dist - radius[s]) > C.surface_tol);
        const double resp = FM ? fmax(-fn, 0.0) : ((fn > 0.0) ? 0.0 : -fn);
        nmag[s] = contact[s] ? (FM ? resp : fabs(resp)) : 0.0;
        const double ntot = contact[s] ? resp + (-C.k * pen) + (-C.nu * vnrm) : 0.0;
        // kinetic friction: axial direction = tangent projected on the plane, rolling = ax x n
        if constexpr (ZUP) {
            ax[s][0] = L.t[s][0]; ax[s][1] = L.t[s][1]; ax[s][2] = 0.0;
        } else {
            const double tn = nr[0] * L.t[s][0] + nr[1] * L.t[s][1] + nr[2] * L.t[s][2];
#pragma unroll
            for (int i = 0; i < 3; ++i) ax[s][i] = L.t[s][i] - nr[i] * tn;
        }
        double a2 = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) a2 += ax[s][i] * ax[s][i];
        const double itp = inv_of<FM>(sqrt_of<FM>(a2) + 1e-14);
#pragma unroll
        for (int i = 0; i < D; ++i) ax[s][i] *= itp;
        if constexpr (ZUP) {
            ro[s][0] = ax[s][1]; ro[s][1] = -ax[s][0]; ro[s][2] = 0.0;
        } else {
            ro[s][0] = ax[s][1] * nr[2] - ax[s][2] * nr[1];
            ro[s][1] = ax[s][2] * nr[0] - ax[s][0] * nr[2];
            ro[s][2] = ax[s][0] * nr[1] - ax[s][1] * nr[0];
        }
        double vax = 0.0, vroll = 0.0, axn2 = 0.0, ron2 = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            vax += vel[i] * ax[s][i];
            vroll += vel[i] * ro[s][i];
            axn2 += ax[s][i] * ax[s][i];
            ron2 += ro[s][i] * ro[s][i];
        }
        // |ax| after the normalisation = |ax_raw| / (|ax_raw| + 1e-14)
        const double axn = FM ? sqrt_of<true>(a2) * itp : sqrt(axn2);
        const double ron = ZUP ? axn : sqrt_of<FM>(ron2);   // |ax x e_z| = |ax| term by term
        // mu_forward for vax > 0, mu_backward for vax < 0, their mean at exactly 0 — where the
        // slip function below is 1 and the kinetic force vanishes whatever kmu is (FM: skip the mean)
        const double sgn = FM ? 0.0 : sign_of(vax);
        // (FM: mean + sign(vax) * half difference — the half difference is negative for the octopus,
        // whose backward friction is the larger — one v_bfi and one fma where a select between two
        // wave-uniform doubles costs four moves and two v_cndmask; mu_f / mu_b to an ulp)
        const double kmu = FM ? fma(copysign(1.0, vax), C.kin_am[1], C.kin_am[0])
                              : 0.5 * (C.kin_mu[0] * (1 + sgn) + C.kin_mu[1] * (1 - sgn));
        double ex_ax = 0.0, ex_ro = 0.0;      // FM: 1 - slip, computed first
        if constexpr (FM) { ex_ax = slip_excess(fabs(vax) * axn, inv_slip); slip_ax[s] = 1.0 - ex_ax; }
        else slip_ax[s] = slip_function<FM>(fabs(vax) * axn, C.slip_tol, inv_slip);
        // velocity of the contact point from the element's spin: Q^T (w x Q arm), arm = -n radius
        const double* Q = L.Q[s];
        const double* w = L.w[s];
        double qa[3], rot[3];
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            if constexpr (ZUP) qa[i] = -radius[s] * Q[3 * i + 2];
            else qa[i] = -radius[s] * (Q[3 * i] * nr[0] + Q[3 * i + 1] * nr[1] + Q[3 * i + 2] * nr[2]);
        }
        const double wq[3] = {w[1] * qa[2] - w[2] * qa[1], w[2] * qa[0] - w[0] * qa[2],
                              w[0] * qa[1] - w[1] * qa[0]};
        double vrot = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            rot[i] = Q[i] * wq[0] + Q[3 + i] * wq[1] + Q[6 + i] * wq[2];
            vrot += rot[i] * ro[s][i];
        }
        const double sroll = vroll + vrot;
        if constexpr (FM) { ex_ro = slip_excess(fabs(sroll) * ron, inv_slip); slip_ro[s] = 1.0 - ex_ro; }
        else slip_ro[s] = slip_function<FM>(fabs(sroll) * ron, C.slip_tol, inv_slip);
        // unit vector of the total slip velocity in the plane, sroll ro + vax ax, with PyElastica's
        // 1e-14 added to every component before the norm (all three: a zero component adds 1e-28)
        double tot[3] = {0.0, 0.0, 0.0}, m2 = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) tot[i] = sroll * ro[s][i] + vax * ax[s][i];
#pragma unroll
        for (int i = 0; i < 3; ++i) { const double te = tot[i] + 1e-14; m2 += te * te; }
        const double ivm = FM ? fast_rsqrt(m2) : 1.0 / sqrt(m2);
        double uax = 0.0, uro = 0.0;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            const double u = FM ? tot[i] * ivm : tot[i] / sqrt(m2);
            uax += u * ax[s][i];
            uro += u * ro[s][i];
        }
        // nmag is 0 off the plane, so the products vanish there by themselves (FM: no second select)
        const double ka = FM ? -(ex_ax * kmu * nmag[s] * uax)
                             : (contact[s] ? -((1.0 - slip_ax[s]) * kmu * nmag[s] * uax) : 0.0);
        const double kr = FM ? -(ex_ro * C.kin_mu[2] * nmag[s] * uro)
                             : (contact[s] ? -((1.0 - slip_ro[s]) * C.kin_mu[2] * nmag[s] * uro) : 0.0);
        double fr[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < D; ++i) fr[i] = kr * ro[s][i];
        if constexpr (ZUP) {
            E[s][0] = ka * ax[s][0] + fr[0];
            E[s][1] = ka * ax[s][1] + fr[1];
            E[s][2] = ntot;
            // arm x fr with arm = (0, 0, -radius)
            const double cr0 = radius[s] * fr[1], cr1 = -radius[s] * fr[0];
#pragma unroll
            for (int i = 0; i < 3; ++i) tq[s][i] += Q[3 * i] * cr0 + Q[3 * i + 1] * cr1;
        } else {
            double arm[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) {
                arm[i] = -nr[i] * radius[s];
                E[s][i] = nr[i] * ntot + ka * ax[s][i] + fr[i];
            }
            const double cr[3] = {arm[1] * fr[2] - arm[2] * fr[1], arm[2] * fr[0] - arm[0] * fr[2],
                                  arm[0] * fr[1] - arm[1] * fr[0]};
#pragma unroll
            for (int i = 0; i < 3; ++i)
                tq[s][i] += Q[3 * i] * cr[0] + Q[3 * i + 1] * cr[1] + Q[3 * i + 2] * cr[2];
        }
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
    // static friction acts in the plane: with ZUP only x, y of the totals are needed
#pragma unroll
    for (int i = 0; i < D; ++i) {
        double a[EPL], o[EPL];
#pragma unroll
        for (int s = 0; s < EPL; ++s) a[s] = F2[s][i];
        shift_next<EPL>(a, o);
#pragma unroll
        for (int s = 0; s < EPL; ++s) Fn[s][i] = o[s];
    }
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        double fel[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < D; ++i) {
            if constexpr (FM) {
                fel[i] = fma(wb[s], Fn[s][i], wa[s] * F2[s][i]);
            } else {
                fel[i] = 0.5 * (F2[s][i] + Fn[s][i]);
                fel[i] += (wa[s] == 1.0) ? 0.5 * F2[s][i] : 0.0;
                fel[i] += (wb[s] == 1.0) ? 0.5 * Fn[s][i] : 0.0;
            }
        }
        double fax = 0.0, fro = 0.0, tax = 0.0;
        const double* Q = L.Q[s];
#pragma unroll
        for (int i = 0; i < D; ++i) {
            fax += fel[i] * ax[s][i];
            fro += fel[i] * ro[s][i];
            const double tt = Q[i] * tq[s][0] + Q[3 + i] * tq[s][1] + Q[6 + i] * tq[s][2];
            tax += tt * ax[s][i];
        }
        // FM: -min(|fax|, cap) sign(fax) = copysign(min(|fax|, cap), -fax); fax = 0 gives 0 either way,
        // and cap = 0 off the plane
        const double sg = FM ? 0.0 : sign_of(fax);
        const double smu = FM ? fma(copysign(1.0, fax), C.stat_am[1], C.stat_am[0])
                              : 0.5 * (C.stat_mu[0] * (1 + sg) + C.stat_mu[1] * (1 - sg));
        const double sa = FM ? copysign(fmin(fabs(fax), slip_ax[s] * smu * nmag[s]), -fax)
                             : (contact[s] ? -(fmin(fabs(fax), slip_ax[s] * smu * nmag[s]) * sg) : 0.0);
        const double noslip = FM ? -((radius[s] * fro - 2.0 * tax) * (1.0 / 3.0) * inv_radius[s])
                                 : -((radius[s] * fro - 2.0 * tax) / 3.0 / radius[s]);
        const double sr = FM ? copysign(fmin(fabs(noslip), slip_ro[s] * C.stat_mu[2] * nmag[s]), noslip)
                             : (contact[s] ? fmin(fabs(noslip), slip_ro[s] * C.stat_mu[2] * nmag[s]) * sign_of(noslip) : 0.0);
        double fr[3] = {0.0, 0.0, 0.0};
#pragma unroll
        for (int i = 0; i < D; ++i) {
            fr[i] = sr * ro[s][i];
            E[s][i] = sa * ax[s][i] + fr[i];
        }
        if constexpr (ZUP) {
            E[s][2] = 0.0;
            const double cr0 = radius[s] * fr[1], cr1 = -radius[s] * fr[0];
#pragma unroll
            for (int i = 0; i < 3; ++i) tq[s][i] += Q[3 * i] * cr0 + Q[3 * i + 1] * cr1;
        } else {
            double arm[3];
#pragma unroll
            for (int i = 0; i < 3; ++i) arm[i] = -nr[i] * radius[s];
            const double cr[3] = {arm[1] * fr[2] - arm[2] * fr[1], arm[2] * fr[0] - arm[0] * fr[2],
                                  arm[0] * fr[1] - arm[1] * fr[0]};
#pragma unroll
            for (int i = 0; i < 3; ++i)
                tq[s][i] += Q[3 * i] * cr[0] + Q[3 * i + 1] * cr[1] + Q[3 * i + 2] * cr[2];
        }
    }
#pragma unroll
    for (int i = 0; i < D; ++i) {
        double a[EPL], o[EPL];
#pragma unroll
        for (int s = 0; s < EPL; ++s) a[s] = E[s][i];
        shift_prev<EPL>(a, o);
#pragma unroll
        for (int s = 0; s < EPL; ++s) fc[s][i] += 0.5 * E[s][i] + 0.5 * o[s];
    }
}


}  // namespace softrod
