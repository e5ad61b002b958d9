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

// In:  F[3]   nodal internal + external force of node k (so far)
//      tq[3]  element internal + external torque (local frame) of element k (so far)
// Out: fc[3]  contact force added to node k;  tq[] += contact torques.
__device__ __forceinline__ void plane_contact(const ContactParams& C, int lane, int n,
                                              double mass, double mass_next,
                                              const double x[3], const double xn[3],
                                              const double v[3], const double vn[3],
                                              const double t[3], const double Q[9], const double w[3],
                                              double len, const double F[3], double tq[3],
                                              double fc[3]) {
    const bool elem_valid = lane < n;
    const bool first = (lane == 0), last = (lane == n - 1);
    const double* nr = C.normal;
    const double radius = C.r0_sqrt_rest_len / sqrt(len);

    // ---- element total force: node_to_element_mass_or_force ----
    double Fn[3], fel[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        Fn[i] = from_next(F[i]);
        fel[i] = 0.5 * (F[i] + Fn[i]);
        fel[i] += first ? 0.5 * F[i] : 0.0;
        fel[i] += last ? 0.5 * Fn[i] : 0.0;
    }
    // ---- normal response, penalty spring and damper ----
    const double fn = nr[0] * fel[0] + nr[1] * fel[1] + nr[2] * fel[2];
    double dist = 0.0, vel[3], vnrm = 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double xe = 0.5 * (x[i] + xn[i]);
        dist += nr[i] * (xe - C.origin[i]);
        vel[i] = (mass_next * vn[i] + mass * v[i]) / (mass_next + mass);
        vnrm += nr[i] * vel[i];
    }
    const double pen = fmin(dist - radius, 0.0);
    const bool contact = elem_valid && !((dist - radius) > C.surface_tol);
    const double resp = (fn > 0.0) ? 0.0 : -fn;   // response magnitude along +normal
    const double nmag = contact ? fabs(resp) : 0.0;   // |plane_response_force|
    double E[3];
    const double ntot = resp + (-C.k * pen) + (-C.nu * vnrm);
#pragma unroll
    for (int i = 0; i < 3; ++i) E[i] = contact ? nr[i] * ntot : 0.0;

    // ---- kinetic friction ----
    const double tn = nr[0] * t[0] + nr[1] * t[1] + nr[2] * t[2];
    double ax[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) ax[i] = t[i] - nr[i] * tn;
    const double tpm = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
    const double itp = 1.0 / (tpm + 1e-14);
#pragma unroll
    for (int i = 0; i < 3; ++i) ax[i] *= itp;
    const double ro[3] = {ax[1] * nr[2] - ax[2] * nr[1], ax[2] * nr[0] - ax[0] * nr[2],
                          ax[0] * nr[1] - ax[1] * nr[0]};
    const double arm[3] = {-nr[0] * radius, -nr[1] * radius, -nr[2] * radius};
    const double vax = vel[0] * ax[0] + vel[1] * ax[1] + vel[2] * ax[2];
    const double axn = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
    const double sgn = sign_of(vax);
    const double kmu = 0.5 * (C.kin_mu[0] * (1 + sgn) + C.kin_mu[1] * (1 - sgn));
    const double slip_ax = slip_function(fabs(vax) * axn, C.slip_tol);
    const double vroll = vel[0] * ro[0] + vel[1] * ro[1] + vel[2] * ro[2];
    // rotation velocity Q^T (omega x (Q arm))
    double qa[3], rot[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) qa[i] = Q[3 * i] * arm[0] + Q[3 * i + 1] * arm[1] + Q[3 * i + 2] * arm[2];
    const double wq[3] = {w[1] * qa[2] - w[2] * qa[1], w[2] * qa[0] - w[0] * qa[2],
                          w[0] * qa[1] - w[1] * qa[0]};
#pragma unroll
    for (int i = 0; i < 3; ++i) rot[i] = Q[i] * wq[0] + Q[3 + i] * wq[1] + Q[6 + i] * wq[2];
    const double vrot = rot[0] * ro[0] + rot[1] * ro[1] + rot[2] * ro[2];
    const double sroll = vroll + vrot;
    const double ron = sqrt(ro[0] * ro[0] + ro[1] * ro[1] + ro[2] * ro[2]);
    const double slip_ro = slip_function(fabs(sroll) * ron, C.slip_tol);
    const double vm = sqrt(vel[0] * vel[0] + vel[1] * vel[1] + vel[2] * vel[2]) + 1e-14;
    const double uax = (vel[0] / vm) * ax[0] + (vel[1] / vm) * ax[1] + (vel[2] / vm) * ax[2];
    const double uro = (vel[0] / vm) * ro[0] + (vel[1] / vm) * ro[1] + (vel[2] / vm) * ro[2];
    const double ka = contact ? -((1.0 - slip_ax) * kmu * nmag * uax) : 0.0;
    const double kr = contact ? -((1.0 - slip_ro) * C.kin_mu[2] * nmag * uro) : 0.0;
    double fr[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        fr[i] = kr * ro[i];
        E[i] += ka * ax[i] + fr[i];
    }
    {   // torque = Q (arm x F_roll)
        const double cr[3] = {arm[1] * fr[2] - arm[2] * fr[1], arm[2] * fr[0] - arm[0] * fr[2],
                              arm[0] * fr[1] - arm[1] * fr[0]};
#pragma unroll
        for (int i = 0; i < 3; ++i) tq[i] += Q[3 * i] * cr[0] + Q[3 * i + 1] * cr[1] + Q[3 * i + 2] * cr[2];
    }
    // scatter round 1: node k += 1/2 (E_k + E_{k-1})
    double F2[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        fc[i] = 0.5 * E[i] + 0.5 * from_prev(E[i]);
        F2[i] = F[i] + fc[i];
    }

    // ---- static friction on the updated totals ----
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        Fn[i] = from_next(F2[i]);
        fel[i] = 0.5 * (F2[i] + Fn[i]);
        fel[i] += first ? 0.5 * F2[i] : 0.0;
        fel[i] += last ? 0.5 * Fn[i] : 0.0;
    }
    const double fax = fel[0] * ax[0] + fel[1] * ax[1] + fel[2] * ax[2];
    const double sg = sign_of(fax);
    const double smu = 0.5 * (C.stat_mu[0] * (1 + sg) + C.stat_mu[1] * (1 - sg));
    const double sa = contact ? -(fmin(fabs(fax), slip_ax * smu * nmag) * sg) : 0.0;
    // rolling: total torque in the lab frame Q^T tq, no-slip force
    double tt[3];
#pragma unroll
    for (int i = 0; i < 3; ++i) tt[i] = Q[i] * tq[0] + Q[3 + i] * tq[1] + Q[6 + i] * tq[2];
    const double tax = tt[0] * ax[0] + tt[1] * ax[1] + tt[2] * ax[2];
    const double fro = fel[0] * ro[0] + fel[1] * ro[1] + fel[2] * ro[2];
    const double noslip = -((radius * fro - 2.0 * tax) / 3.0 / radius);
    const double sr = contact ? fmin(fabs(noslip), slip_ro * C.stat_mu[2] * nmag) * sign_of(noslip) : 0.0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        fr[i] = sr * ro[i];
        E[i] = sa * ax[i] + fr[i];
    }
    {
        const double cr[3] = {arm[1] * fr[2] - arm[2] * fr[1], arm[2] * fr[0] - arm[0] * fr[2],
                              arm[0] * fr[1] - arm[1] * fr[0]};
#pragma unroll
        for (int i = 0; i < 3; ++i) tq[i] += Q[3 * i] * cr[0] + Q[3 * i + 1] * cr[1] + Q[3 * i + 2] * cr[2];
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) fc[i] += 0.5 * E[i] + 0.5 * from_prev(E[i]);
}

}  // namespace softrod
