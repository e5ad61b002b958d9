// softrod_fast.hpp — SOFTROD_MATH_FAST step kernel: the PositionVerlet substep of
// softrod_kernels.hpp reorganised for the CDNA4 fp64 VALU (a wave64 fp64 FMA issues in
// 4 cycles, so the kernel is bound by its fp64 instruction count, not by memory).
//
// Same mathematics as the LIBM kernel / the oracle; what changes is how it is evaluated:
//   * the second kinematic half-step of substep s and the first of substep s+1 use the
//     same (v, omega) and are merged into one full step: R(h w)R(h w) = R(2h w) and
//     x + h v + h v = x + 2h v (only the first and last of a launch stay half steps);
//   * constrain_values is an invariant instead of a per-half-step reset: the held
//     components of node 0 have v = 0 and the held element's omega is pinned by
//     constrain_rates, so x += h*0 leaves them in place, and the director of the
//     held element is simply not rotated (PendulumBoundaryConditions resets rows 0,2
//     every time and row 1 is untouched by a rotation about d2: build.py:71-79).  The
//     invariant is (re)established once at kernel entry, so states written by reset or
//     by a previous step are handled exactly like the reference;
//   * sin/cos of the Rodrigues angle, theta/sin(theta) of _inv_rotate and the damper's
//     pow() are short polynomials in their (tiny) arguments with wave-uniform range
//     checks; outside the range the argument is halved until it fits and the result
//     rebuilt with double-angle / squaring identities (no libm calls, which keeps the
//     register budget small enough for several waves per SIMD);
//   * divisions become one Newton-refined v_rcp_f64 / v_rsq_f64 each; circular cross
//     sections (I1 = I2, always true for CosseratRod.straight_rod) are exploited;
//   * lane-validity masks are folded into per-lane stiffness / time-step constants.
// All of these are ulp-level reorderings; tests/test_gpu_parity.py holds the kernel to
// the same rtol 1e-5 against the oracle as the LIBM kernel.
#pragma once

namespace softrod {

// 1/x: v_rcp_f64 seed (~2^-25) + two Newton steps -> <= 1 ulp for normal x.
__device__ __forceinline__ double fast_rcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    double e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    e = fma(-x, r, 1.0);
    r = fma(r, e, r);
    return r;
}
// 1/sqrt(x): v_rsq_f64 seed + two Newton steps.
__device__ __forceinline__ double fast_rsqrt(double x) {
    double r = __builtin_amdgcn_rsq(x);
    const double hx = 0.5 * x;
    double e = fma(-hx * r, r, 0.5);
    r = fma(r, e, r);
    e = fma(-hx * r, r, 0.5);
    r = fma(r, e, r);
    return r;
}

// Per-lane constants (VGPRs), built once per launch.  Lane-validity masks and the
// loop-invariant parts of the rate update live here.
struct FastConst {
    double hx;        // 1, or 0 for a node whose position the BC holds against its velocity
    double hq;        // 1, or 0 for an element whose director the BC holds
    double cf;        // damp_t*dt/m_k            (0 beyond the last node)
    double ca[3];     // damp_t*dt*F_ext,i/m_k    (gravity, action, tip force)
    double cw01, cw2; // dt/J_i                   (0 beyond the last element)
    double s01, s2;   // shear/stretch stiffness  (0 beyond the last element)
    double b01, bd;   // bend stiffness EI and (GI3 - EI)  (0 beyond the last Voronoi vertex)
};

// sin(th)/th and (1-cos(th))/th^2 from t = th^2.  t < 1e-3: degree-3 Taylor in t
// (remainders t^4/9! < 3e-18, t^4/10! < 3e-19).  Otherwise (|omega| dt > 0.03 rad: only
// when a simulation is blowing up) the angle is halved k times and rebuilt with
//   sinc(2p) = sinc(p) cos(p),  cosc(2p) = sinc(p)^2 / 2,  cos(p) = 1 - cosc(p) p^2.
__device__ __forceinline__ void sinc_cosc(double t, double& sc, double& cc) {
    int k = 0;
    while (__any(t >= 1.0e-3) && k < 48) { t *= 0.25; ++k; }   // wave-uniform trip count
    sc = fma(t, fma(t, fma(t, -1.0 / 5040.0, 1.0 / 120.0), -1.0 / 6.0), 1.0);
    cc = fma(t, fma(t, fma(t, -1.0 / 40320.0, 1.0 / 720.0), -1.0 / 24.0), 0.5);
    for (; k > 0; --k) {
        const double c = fma(-cc, t, 1.0);
        cc = 0.5 * sc * sc;
        sc = sc * c;
        t *= 4.0;
    }
}

// theta/sin(theta) as a function of y = sin^2(theta/2) = (1 - cos theta)/2:
//   asin(sqrt y)/(sqrt y sqrt(1-y)) = sum_k (2k)!!/(2k+1)!! y^k.
// y < 2.5e-3 (neighbouring elements < 0.1 rad apart): 6 terms, remainder 0.34 y^6 < 1e-16.
// Otherwise the half-angle recursion y' = y / (2 (1 + sqrt(1-y))) is applied until it
// fits and theta/sin(theta) = 2^k (phi/sin phi) sin(phi)/sin(theta).
__device__ __forceinline__ double theta_over_sin(double y, bool valid) {
    const double y0 = y;
    int k = 0;
    while (__any(valid && !(y < 2.5e-3)) && k < 12) {
        const double om = fmax(1.0 - y, 1.0e-300);
        y = 0.5 * y / (1.0 + om * fast_rsqrt(om));
        ++k;
    }
    double g = 256.0 / 693.0;            // k = 5
    g = fma(g, y, 128.0 / 315.0);        // k = 4
    g = fma(g, y, 16.0 / 35.0);          // k = 3
    g = fma(g, y, 8.0 / 15.0);           // k = 2
    g = fma(g, y, 2.0 / 3.0);            // k = 1
    g = fma(g, y, 1.0);
    if (k > 0) {
        // sin(phi) = 2 sqrt(y(1-y)) at both levels
        const double a = y * (1.0 - y), b = fmax(y0 * (1.0 - y0), 1.0e-300);
        g *= (double)(1 << k) * (a * fast_rsqrt(a)) * fast_rsqrt(b);
    }
    return g;
}

// exp(x) for the damper: |x| < 1e-3 -> degree-4 Taylor (remainder x^5/120 < 1e-17);
// larger |x| (strong damping constants) are halved k times and squared back.
__device__ __forceinline__ void exp_pair(double x0, double x2, bool valid, double& e0, double& e2) {
    int k = 0;
    while (__any(valid && !(fmax(fabs(x0), fabs(x2)) < 1.0e-3)) && k < 60) {
        x0 *= 0.5; x2 *= 0.5; ++k;
    }
    e0 = fma(x0, fma(x0, fma(x0, fma(x0, 1.0 / 24.0, 1.0 / 6.0), 0.5), 1.0), 1.0);
    e2 = fma(x2, fma(x2, fma(x2, fma(x2, 1.0 / 24.0, 1.0 / 6.0), 0.5), 1.0), 1.0);
    for (; k > 0; --k) { e0 *= e0; e2 *= e2; }
}

// x += h v ;  Q <- R(h w) Q   (h = dt/2 at the ends of a launch, dt in between)
__device__ __forceinline__ void fast_kinematic_step(double h, const FastConst& C, LaneState& L) {
    const double hp = h * C.hx;
    L.x[0] = fma(hp, L.v[0], L.x[0]);
    L.x[1] = fma(hp, L.v[1], L.x[1]);
    L.x[2] = fma(hp, L.v[2], L.x[2]);
    const double hh = h * C.hq;
    const double a0 = hh * L.w[0], a1 = hh * L.w[1], a2 = hh * L.w[2];
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
        const double b0 = L.Q[j], b1 = L.Q[3 + j], b2 = L.Q[6 + j];
        L.Q[j] = fma(R2, b2, fma(R1, b1, R0 * b0));
        L.Q[3 + j] = fma(R5, b2, fma(R4, b1, R3 * b0));
        L.Q[6 + j] = fma(R8, b2, fma(R7, b1, R6 * b0));
    }
}

// forces, torques, rate update, damper, constrain_rates — steps (3)-(6) of the substep.
template <unsigned F>
__device__ __forceinline__ void fast_dynamic_step(const RodParams& P, const FastConst& C,
                                                  const BcTargets& B, int lane, LaneState& L) {
    const int n = P.n_elem;
    const bool elem_valid = lane < n;
    const bool vor_valid = lane < n - 1;

    // ---- geometry ----
    const double xn0 = from_next(L.x[0]), xn1 = from_next(L.x[1]), xn2 = from_next(L.x[2]);
    const double d0 = xn0 - L.x[0], d1 = xn1 - L.x[1], d2 = xn2 - L.x[2];
    double dd = fma(d2, d2, fma(d1, d1, d0 * d0));
    dd = elem_valid ? dd : 1.0;  // lanes beyond the rod stay finite; their stiffness is 0
    const double r = fast_rsqrt(dd);
    const double len = fma(dd, r, P.eps_length);
    const double il = fma(-P.eps_length * r, r, r);   // 1/(|d| + eps) to first order in eps
    L.t[0] = d0 * il; L.t[1] = d1 * il; L.t[2] = d2 * il;
    const double e = len * P.inv_rest_len;
    const double ie = P.rest_len * il;

    // ---- shear/stretch: n/e = S (Q t - z/e), lab-frame stress Q^T n / e ----
    const double qt0 = fma(L.Q[2], L.t[2], fma(L.Q[1], L.t[1], L.Q[0] * L.t[0]));
    const double qt1 = fma(L.Q[5], L.t[2], fma(L.Q[4], L.t[1], L.Q[3] * L.t[0]));
    const double qt2 = fma(L.Q[8], L.t[2], fma(L.Q[7], L.t[1], L.Q[6] * L.t[0]));
    const double np0 = C.s01 * qt0, np1 = C.s01 * qt1, np2 = C.s2 * (qt2 - ie);
    const double cs0 = fma(L.Q[6], np2, fma(L.Q[3], np1, L.Q[0] * np0));
    const double cs1 = fma(L.Q[7], np2, fma(L.Q[4], np1, L.Q[1] * np0));
    const double cs2 = fma(L.Q[8], np2, fma(L.Q[5], np1, L.Q[2] * np0));
    const double f0 = cs0 - from_prev(cs0);
    const double f1 = cs1 - from_prev(cs1);
    const double f2 = cs2 - from_prev(cs2);

    // ---- bend/twist on the Voronoi vertex between elements k and k+1 ----
    double Qn[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) Qn[i] = from_next(L.Q[i]);
    const double len_n = from_next(len);
#define SR_RD(i, j) fma(Qn[3 * (i) + 2], L.Q[3 * (j) + 2], fma(Qn[3 * (i) + 1], L.Q[3 * (j) + 1], \
                        Qn[3 * (i)] * L.Q[3 * (j)]))
#define SR_RD_SUB(i, j, acc) fma(-Qn[3 * (i) + 2], L.Q[3 * (j) + 2], fma(-Qn[3 * (i) + 1], \
                        L.Q[3 * (j) + 1], fma(-Qn[3 * (i)], L.Q[3 * (j)], acc)))
    const double vec0 = SR_RD_SUB(1, 2, SR_RD(2, 1));
    const double vec1 = SR_RD_SUB(2, 0, SR_RD(0, 2));
    const double vec2 = SR_RD_SUB(0, 1, SR_RD(1, 0));
    const double trace = SR_RD(0, 0) + SR_RD(1, 1) + SR_RD(2, 2);
#undef SR_RD
#undef SR_RD_SUB
    // y = (1 - cos theta)/2 with cos theta = trace/2 - 1/2 - acos_shift
    const double y = fma(-0.25, trace, 0.75 + 0.5 * P.acos_shift);
    const double gk = theta_over_sin(y, vor_valid) * (-0.5 * P.inv_rest_vor);
    const double k0 = vec0 * gk, k1 = vec1 * gk, k2 = vec2 * gk;
    const double vd = (len_n + len) * (0.5 * P.inv_rest_vor);
    const double rvd = fast_rcp(vd);
    const double e3 = rvd * rvd * rvd;
    double tq0, tq1, tq2;
    if (F == kRuntimeFeatures) { L.kap[0] = k0; L.kap[1] = k1; L.kap[2] = k2; }
    if (has<F>(P, SOFTROD_FEAT_REST_KAPPA_ACTION)) {
        // intrinsic curvature: m = B (kappa - kappa_rest), general cross product
        L.kap[0] = k0; L.kap[1] = k1; L.kap[2] = k2;
        const double m0 = C.b01 * (k0 - L.rk[0]), m1 = C.b01 * (k1 - L.rk[1]),
                     m2 = (C.b01 + C.bd) * (k2 - L.rk[2]);
        const double c20 = m0 * e3, c21 = m1 * e3, c22 = m2 * e3;
        const double hd = 0.5 * P.rest_vor * e3;
        const double h30 = (k1 * m2 - k2 * m1) * hd, h31 = (k2 * m0 - k0 * m2) * hd,
                     h32 = (k0 * m1 - k1 * m0) * hd;
        tq0 = (c20 + h30) - from_prev(c20 - h30);
        tq1 = (c21 + h31) - from_prev(c21 - h31);
        tq2 = (c22 + h32) - from_prev(c22 - h32);
    } else {
        // couples / eps^3: c2 = B kappa, c3 = (kappa x B kappa) D ; with B = diag(b, b, b + bd)
        // kappa x B kappa = bd k2 (k1, -k0, 0)
        const double c20 = C.b01 * k0 * e3, c21 = C.b01 * k1 * e3, c22 = (C.b01 + C.bd) * k2 * e3;
        const double hz = 0.5 * P.rest_vor * C.bd * k2 * e3;   // (1/2) |c3| factor
        const double h30 = k1 * hz, h31 = -k0 * hz;             // (1/2) c3
        // element k: (c2_k - c2_{k-1}) + 1/2 (c3_k + c3_{k-1}) = (c2 + h3)_k - (c2 - h3)_{k-1}
        tq0 = (c20 + h30) - from_prev(c20 - h30);
        tq1 = (c21 + h31) - from_prev(c21 - h31);
        tq2 = c22 - from_prev(c22);
    }

    // shear/stretch couple (Q t) x n l_rest = len (Q t) x (n/e)
    tq0 = fma(len, fma(qt1, np2, -qt2 * np1), tq0);
    tq1 = fma(len, fma(qt2, np0, -qt0 * np2), tq1);
    tq2 = fma(len, fma(qt0, np1, -qt1 * np0), tq2);

    // transport (J w/e) x w = (J1 - J3)/e w2 (w1, -w0, 0) ; unsteady dilatation (J w/e)(de/dt)/e
    const double vn0 = from_next(L.v[0]), vn1 = from_next(L.v[1]), vn2 = from_next(L.v[2]);
    const double num = fma(d2, vn2 - L.v[2], fma(d1, vn1 - L.v[1], d0 * (vn0 - L.v[0])));
    const double sdil = num * il * il;                 // (de/dt)/e = (dx.dv)/|dx|^2
    const double j01 = P.J[0] * ie, j2 = P.J[2] * ie;
    const double z = L.w[2] * (j01 - j2);
    tq0 = fma(L.w[1], z, tq0);
    tq1 = fma(-L.w[0], z, tq1);
    const double js01 = j01 * sdil, js2 = j2 * sdil;
    tq0 = fma(js01, L.w[0], tq0);
    tq1 = fma(js01, L.w[1], tq1);
    tq2 = fma(js2, L.w[2], tq2);

    // ---- plane contact + anisotropic friction (after the forcing operators) ----
    double fc0 = f0, fc1 = f1, fc2 = f2;
    if (has<F>(P, SOFTROD_FEAT_PLANE_CONTACT_ANISO)) {
        const bool node_valid = lane <= n;
        const double mass = (lane == 0 || lane == n) ? 0.5 * P.mass_node : P.mass_node;
        const double mass_next = (lane + 1 == n) ? 0.5 * P.mass_node : P.mass_node;
        // nodal internal + external force so far (gravity; no point/tip force with contact)
        double Fg[3] = {f0, f1, f2};
        if (has<F>(P, SOFTROD_FEAT_GRAVITY) && !P.contact_before_forcing) {
#pragma unroll
            for (int i = 0; i < 3; ++i) Fg[i] += node_valid ? P.gravity[i] * mass : 0.0;
        }
        const double xn[3] = {xn0, xn1, xn2}, vn[3] = {vn0, vn1, vn2};
        double tq[3] = {tq0, tq1, tq2}, fc[3];
        plane_contact(contact_params(P), lane, n, mass, mass_next, L.x, xn, L.v, vn, L.t, L.Q, L.w,
                      len, Fg, tq, fc);
        fc0 += fc[0]; fc1 += fc[1]; fc2 += fc[2];
        tq0 = tq[0]; tq1 = tq[1]; tq2 = tq[2];
    }

    // ---- rate update fused with the analytical damper ----
    //   v <- c_t (v + dt (f + f_ext)/m)          w <- (w + dt e tau/J) c_r^e
    L.v[0] = fma(P.damp_t, L.v[0], fma(C.cf, fc0, C.ca[0]));
    L.v[1] = fma(P.damp_t, L.v[1], fma(C.cf, fc1, C.ca[1]));
    L.v[2] = fma(P.damp_t, L.v[2], fma(C.cf, fc2, C.ca[2]));
    const double ce01 = C.cw01 * e, ce2 = C.cw2 * e;
    double w0 = fma(ce01, tq0, L.w[0]), w1 = fma(ce01, tq1, L.w[1]), w2 = fma(ce2, tq2, L.w[2]);
    if (has<F>(P, SOFTROD_FEAT_ANALYTICAL_DAMPER)) {
        double ex0, ex2;
        exp_pair(e * P.damp_logr[0], e * P.damp_logr[2], elem_valid, ex0, ex2);
        w0 *= ex0; w1 *= ex0; w2 *= ex2;
    }
    L.w[0] = w0; L.w[1] = w1; L.w[2] = w2;
    // The analytical damper is fused into the update above.  Where constrain_rates is
    // registered BEFORE the dampers (operator order of the env's build function), the
    // imposed rates are damped like every other entry: targets scaled by c_t (held omega
    // targets are 0 either way), then the Laplace filter runs on the constrained field.
    if (P.damp_before_constrain) {
        if (has<F>(P, SOFTROD_FEAT_LAPLACE_FILTER)) laplace_filter_rates(P, lane, L);
        constrain_rates<F>(P, B, lane, L);
    } else {
        BcTargets Bs = B;
        Bs.vel[0] *= P.damp_t; Bs.vel[1] *= P.damp_t; Bs.vel[2] *= P.damp_t;
        constrain_rates<F>(P, Bs, lane, L);
        if (has<F>(P, SOFTROD_FEAT_LAPLACE_FILTER)) laplace_filter_rates(P, lane, L);
    }
}

// SOFTROD_FAST_WAVES: minimum waves per SIMD the register allocator must leave room for
// (2nd __launch_bounds__ argument = waves per EU on gfx950); tuned in profiles/README.md.
#ifndef SOFTROD_FAST_WAVES
#define SOFTROD_FAST_WAVES 3
#endif
// The contact instantiation needs more live values per lane; it trades a wave of occupancy
// for not spilling.
template <unsigned F, int E>
__global__ void __launch_bounds__(kLanes, ((F != kRuntimeFeatures && (F & SOFTROD_FEAT_PLANE_CONTACT_ANISO)) ? 2 : SOFTROD_FAST_WAVES))
softrod_step_fast_kernel(const RodParams P, const StatePtrs S, const float* __restrict__ actions,
                         float* __restrict__ obs, double* __restrict__ reward,
                         uint8_t* __restrict__ terminated, uint8_t* __restrict__ truncated,
                         double* __restrict__ aux, const int n_sub, const int epilogue) {
    const int rod = blockIdx.x;
    const int lane = threadIdx.x;
    const size_t N = (size_t)P.n_envs;
    const size_t row = (size_t)rod * kLanes + lane;
    const int n = P.n_elem;

    LaneState L;
    load_state<F>(S, N, row, L);
    BcTargets B;
    load_bc(S, N, rod, B);
    EnvAction A;
    env_set_action<F, E>(P, S, N, rod, lane, actions, A, B, L);
    {   // (re)establish the boundary-condition invariant once.  The imposed base velocity
        // of MOVING_BASE_BC is NOT applied here: the reference keeps the previous step's
        // base velocity until the first constrain_rates of the new step.
        BcTargets B0 = B;
        if (has<F>(P, SOFTROD_FEAT_MOVING_BASE_BC)) {
            const double v0x = __shfl(L.v[0], 0), v0y = __shfl(L.v[1], 0), v0z = __shfl(L.v[2], 0);
            B0.vel[0] = v0x; B0.vel[1] = v0y; B0.vel[2] = v0z;
        }
        constrain_rates<F>(P, B0, lane, L);
        constrain_values<F>(P, B, lane, L);
    }
    double time = S.time[rod];

    FastConst C;
    {
        const bool l0 = (lane == 0);
        const bool held_q = l0 && has<F>(P, SOFTROD_FEAT_PENDULUM_BC | SOFTROD_FEAT_FIXED_BC |
                                            SOFTROD_FEAT_MOVING_BASE_BC);
        const bool held_x = l0 && has<F>(P, SOFTROD_FEAT_FIXED_BC | SOFTROD_FEAT_MOVING_BASE_BC);
        const bool node_valid = lane <= n, elem_valid = lane < n, vor_valid = lane < n - 1;
        const double mass = (lane == 0 || lane == n) ? 0.5 * P.mass_node : P.mass_node;
        const bool damp = has<F>(P, SOFTROD_FEAT_ANALYTICAL_DAMPER);
        const double ct = damp ? P.damp_t : 1.0;
        C.hx = held_x ? 0.0 : 1.0;
        C.hq = held_q ? 0.0 : 1.0;
        const double cdm = node_valid ? ct * P.dt / mass : 0.0;
        C.cf = cdm;
        double fe0 = 0.0, fe1 = 0.0, fe2 = 0.0;
        if (has<F>(P, SOFTROD_FEAT_GRAVITY)) {
            fe0 = P.gravity[0] * mass; fe1 = P.gravity[1] * mass; fe2 = P.gravity[2] * mass;
        }
        // PendulumPointForces ASSIGNS external_forces[0,0] (soft_pendulum/build.py:101)
        if (has<F>(P, SOFTROD_FEAT_POINT_FORCE_NODE0_X)) fe0 = l0 ? A.force : fe0;
        if (has<F>(P, SOFTROD_FEAT_TIP_FORCE) && lane == n) {
            fe0 += P.tip_force[0]; fe1 += P.tip_force[1]; fe2 += P.tip_force[2];
        }
        C.ca[0] = cdm * fe0; C.ca[1] = cdm * fe1; C.ca[2] = cdm * fe2;
        C.cw01 = elem_valid ? P.dt * P.invJ[0] : 0.0;
        C.cw2 = elem_valid ? P.dt * P.invJ[2] : 0.0;
        C.s01 = elem_valid ? P.shear[0] : 0.0;
        C.s2 = elem_valid ? P.shear[2] : 0.0;
        C.b01 = vor_valid ? P.bend[0] : 0.0;
        C.bd = vor_valid ? P.bend[2] - P.bend[0] : 0.0;
    }
    RodParams Pk = P;
    if (!has<F>(P, SOFTROD_FEAT_ANALYTICAL_DAMPER)) Pk.damp_t = 1.0;

    if (n_sub > 0) {
        fast_kinematic_step(P.half_dt, C, L);
        if (P.time_two_half_adds) time += P.half_dt;
        for (int s = 0; s < n_sub; ++s) {
            fast_dynamic_step<F>(Pk, C, B, lane, L);
            const bool last = (s == n_sub - 1);
            fast_kinematic_step(last ? P.half_dt : P.dt, C, L);
            time += P.time_two_half_adds ? P.half_dt : P.dt;          // end of substep s
            if (!last && P.time_two_half_adds) time += P.half_dt;      // start of substep s+1
        }
    }

    store_state<F>(S, N, row, L);
    if (lane == 0) S.time[rod] = time;
    if (epilogue)
        env_epilogue<E>(P, S, N, rod, lane, L, time, A, obs, reward, terminated, truncated, aux);
}

}  // namespace softrod
