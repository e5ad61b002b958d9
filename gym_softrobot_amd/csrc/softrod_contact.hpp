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
#pragma once

namespace softrod {

struct ContactParams {
    double origin[3], normal[3];
    double k, nu, slip_tol, surface_tol;
    double kin_mu[3], stat_mu[3];   // forward, backward, sideways
    double r0_sqrt_rest_len;        // r0 * sqrt(l_rest): radius = this / sqrt(len)
};

__device__ __forceinline__ double sign_of(double x) { return (double)((x > 0.0) - (x < 0.0)); }

// find_slipping_elements on a vector of magnitude |a|
__device__ __forceinline__ double slip_function(double a, double thr) {
    const double m = fmin(a / thr - 1.0, 1.0);
    return (fabs(a) > thr) ? fabs(1.0 - m) : 1.0;
}

// In:  F[s][3]   nodal internal + external force of node EPL*lane+s (so far)
//      tq[s][3]  element internal + external torque (local frame) (so far)
// Out: fc[s][3]  contact force added to that node;  tq += contact torques.
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
        const int idx = slot_local(P, lane * EPL + s);
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
        const int idx = slot_local(P, lane * EPL + s);
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


}  // namespace softrod
