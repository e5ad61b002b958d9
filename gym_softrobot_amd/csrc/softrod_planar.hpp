// softrod_planar.hpp — the SoftPendulum-v0 substep when the rod lies in the x-y plane.
//
// build_soft_pendulum (gym_softrobot/envs/soft_pendulum/build.py:46-113) starts the rod in
// the plane z = 0 (direction (cos t, sin t, 0), normal (sin t, -cos t, 0)), and everything
// that acts on it is in-plane: gravity along -y (:88-91), the point force along x (:94-105),
// the pendulum constraint (:65-85).  The 3-D Cosserat update then keeps, IN IEEE
// ARITHMETIC AND EXACTLY, every out-of-plane quantity at zero: x_z = v_z = 0, d1_z = d3_z =
// 0, d2 = (0, 0, +-1), omega = (0, w, 0) in the local frame, kappa = (0, k, 0); products
// with exact zeros are exact zeros and sums of them stay zero.  So for such a state the
// general kernel spends two thirds of its fp64 instructions computing zeros.
//
// This file is that same substep with the zeros removed: positions and velocities have two
// components, the frame is the unit vector d3 = (c, s) (d1 = d2 x d3 = (-s2 s, s2 c) with
// s2 = d2_z), the angular velocity and the curvature are scalars.  Every formula below is
// the corresponding line of dynamic_n / kinematic_n (softrod_fast.hpp) with the identically
// zero terms dropped; the comment on each block says which.  The only non-identical step is
// that d1 is rebuilt from d3 instead of being rotated separately (they differ by the
// rounding of one rotation, ~1e-16).
//
// One more consequence of planarity is used: element k turns about z at the rate s2 w_k, so
// the bending angle D_k = angle(d3_{k+1}) - angle(d3_k) obeys dD_k/dt = s2 (w_{k+1} - w_k)
// exactly, and kappa = -log(Q+ Q^T)/D^ reduces to s2 D_k / D^ (the theta/sin(theta) factor of
// _inv_rotate cancels the sine it multiplies).  D_k is therefore carried as one more state
// variable per Voronoi vertex — initialised from the directors with atan2 at kernel entry,
// advanced with one subtraction and one FMA per kinematic step — instead of being recovered
// from the directors every substep through sin D, cos D and the theta/sin(theta) series.  The
// reference's `- 1e-10` inside arccos turns theta^2 into D^2 + 2e-10, i.e. multiplies kappa by
// (sin D / D)(theta / sin theta) = 1 + 1e-10/3 + O(1e-10 D^2); that constant factor is kept.
// Its `+ 1e-14` inside the sine (theta / sin(theta + 1e-14), softrod_config.eps_sin) multiplies
// kappa by 1 - 1e-14 cot(theta) = 1 - 1e-14 / sqrt(D^2 + 2e-10) to 1e-16: up to 7e-10 for a nearly
// straight joint, a SYSTEMATIC change of the bending stiffness.  It is applied (one raw v_rsq_f64
// seed and two FMAs per substep: the term is < 1e-9, so 2^-23 of it is nothing) — without it the
// stabilised inverted pendulum leaves the 1e-5 tolerance after 89 env.steps instead of 107
// (profiles/r3_fastmath_cost.json; every other reformulation in this file costs no horizon).
//
// The step kernel takes this path only if the loaded state IS planar (planar_from_lane:
// exact zeros where the argument above needs them, d1 consistent with d3 to 1e-12); any
// other state — e.g. one written through softrod_state_view — runs the general 3-D loop.
//
// The loop is VALU-issue bound, so what is left is written to the instruction:
//   * the state carries the rotation rate about z, wz = s2 w, instead of w: every s2 in the
//     formulas multiplies another s2 or cancels against it (multiplying by +-1 is exact, so
//     this is bit-identical), and d1 never has to be formed;
//   * products of loop-invariant factors are folded into per-lane constants (PlanarC);
//   * masks are folded into those constants instead of being applied with selects: the pinned
//     v_y of node 0 has a zero force coefficient, an invalid slot has |d|^2 clamped from below
//     (its stiffnesses are zero) and may carry a finite, unused bending angle;
//   * the Taylor coefficients are handed to the compiler as opaque registers: as immediates it
//     turns every Horner step into v_mov + v_fmac instead of one v_fma;
//   * 1/x and 1/sqrt(x) take one third-order correction of the 2^-23 hardware seed.
//
// Diagnostic builds (tools/fastmath_cost.sh -> variants/, never the shipped library) undo ONE of
// these reformulations each, so that its share of the parity horizon can be measured on the
// stabilised inverted pendulum (DESIGN.md §3 "fast-math cost"):
//   SOFTROD_DIAG_IEEE_DIV          IEEE 1/x and 1/sqrt(x) instead of the refined seeds
//   SOFTROD_DIAG_LIBM_TRIG         libm sin / cos / exp instead of the range-checked polynomials
//   SOFTROD_DIAG_RECOMPUTE_EDGES   edge vectors from the node positions every substep (as the
//                                  reference does) instead of integrating them with v_{k+1} - v_k
//   SOFTROD_DIAG_RECOMPUTE_ANGLE   bending angle from the directors (atan2) every substep instead
//                                  of integrating it with the rotation-rate difference
//   SOFTROD_DIAG_TWO_HALF_STEPS    two kinematic half steps between force evaluations, as
//                                  PositionVerlet takes them, instead of one merged step
//   SOFTROD_DIAG_NO_PLANAR         never take this path: the general 3-D fast loop steps the rod
//   SOFTROD_DIAG_NO_EPS_SIN        drop the sin(theta + eps_sin) term (what rounds 1 and 2 shipped)
#pragma once

namespace softrod {

template <int EPL>
struct PlanarN {
    double x[EPL][2], v[EPL][2];
    double c[EPL], s[EPL];     // d3 = (c, s, 0)
    double s2;                 // d2 = (0, 0, s2), s2 = +-1, the same for every element
    double wz[EPL];            // rotation rate about z: omega = (0, s2 wz, 0) in the local frame
    double t[EPL][2];          // tangents as of the last force evaluation
    double dl[EPL];            // bending angle between elements k and k+1
    double d[EPL][2];          // edge vector x_{k+1} - x_k, carried: dd/dt = v_{k+1} - v_k
    double dv[EPL][2];         // v_{k+1} - v_k of the current velocities
};

// loop-invariant per-lane products (see the header) and the opaque polynomial coefficients
template <int EPL>
struct PlanarC {
    double cfy[EPL], cay[EPL];     // C.cf, C.ca[1] with node 0 (v_y pinned) zeroed
    double cwl[EPL];               // C.cw01 / rest_len:   cw01 * e = cwl * len
    double bk[EPL];                // C.b01 * kappa scale * (2 rest_vor)^3:  B kappa / vd^3 = bk D / (l + l+)^3
    double bke[EPL];               // -eps_sin * bk: the sin(theta + eps_sin) term, bk (1 - eps_sin / theta) = bk + bke / theta
    double xl[EPL];                // damp_logr / rest_len:  e * logr = xl * len  (0 on invalid slots)
    double hq_dt[EPL], hq_hdt[EPL];  // C.hq * dt, C.hq * dt/2
    double jr;                     // J * rest_len:          J / e = jr / len
    double s3, s2, s1, c3, c2, e4, e3;
    double eps_length, rest_len, damp_t;   // RodParams values the loop uses (see uniform_k)
    double two_shift;                      // 2 acos_shift (+ a denormal guard): theta^2 = D^2 + two_shift
};

__device__ __forceinline__ double opaque_s(double k) { asm("" : "+s"(k)); return k; }
__device__ __forceinline__ double opaque_v(double k) { asm("" : "+v"(k)); return k; }
// A wave-uniform constant for the loop: a scalar register at one slot per lane; a vector
// register at two — that instantiation has 512 vector registers and runs out of the 102 scalar
// ones (64 v_readlane per substep of spilled scalars otherwise).
template <int EPL>
__device__ __forceinline__ double uniform_k(double k) { return EPL > 1 ? opaque_v(k) : opaque_s(k); }

// x = m * x + a written over x (the compiler's v_fmac would put it over a and copy it back)
__device__ __forceinline__ void fma_inplace(double& x, double m, double a) {
    asm("v_fma_f64 %0, %1, %0, %2" : "+v"(x) : "v"(m), "v"(a));
}

__device__ __forceinline__ double rcp3(double x) { return fast_rcp(x); }       // (third order, softrod_kernels.hpp)
__device__ __forceinline__ double rsqrt3(double x) { return fast_rsqrt(x); }

template <int EPL>
__device__ __forceinline__ void planar_build_const(const RodParams& P, const ConstN<EPL>& C, int lane,
                                                   PlanarC<EPL>& K) {
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        const bool pinned = (lane * EPL + s) == 0;
        K.cfy[s] = pinned ? 0.0 : C.cf[s];
        K.cay[s] = pinned ? 0.0 : C.ca[s][1];
        K.cwl[s] = C.cw01[s] * P.inv_rest_len;
        const double two_vor = 2.0 * P.rest_vor;
        K.bk[s] = C.b01[s] * (P.inv_rest_vor * (1.0 + P.acos_shift * (1.0 / 3.0))) * (two_vor * two_vor * two_vor);
        K.bke[s] = P.neg_eps_sin * K.bk[s];
        K.xl[s] = (lane * EPL + s) < P.n_elem ? P.damp_logr[0] * P.inv_rest_len : 0.0;
        K.hq_dt[s] = C.hq[s] * P.dt;
        if (EPL > 1) K.hq_dt[s] = opaque_v(K.hq_dt[s]);     // (or it is recomputed in the loop from a spilled dt)
        K.hq_hdt[s] = C.hq[s] * P.half_dt;
    }
    K.jr = P.J[0] * P.rest_len;
    K.s3 = uniform_k<EPL>(-1.0 / 5040.0); K.s2 = opaque_v(1.0 / 120.0); K.s1 = uniform_k<EPL>(-1.0 / 6.0);
    K.c3 = uniform_k<EPL>(-1.0 / 720.0); K.c2 = opaque_v(1.0 / 24.0);
    K.e4 = uniform_k<EPL>(1.0 / 24.0); K.e3 = opaque_v(1.0 / 6.0);
    K.eps_length = EPL > 1 ? opaque_v(P.eps_length) : P.eps_length;
    K.rest_len = EPL > 1 ? opaque_v(P.rest_len) : P.rest_len;
    K.damp_t = EPL > 1 ? opaque_v(P.damp_t) : P.damp_t;
    K.two_shift = EPL > 1 ? opaque_v(P.two_acos_shift) : P.two_acos_shift;
    if (EPL > 1) K.jr = opaque_v(K.jr);
}

// exp(x) for the damper (see exp_pair)
template <int EPL>
__device__ __forceinline__ double exp_one(const PlanarC<EPL>& K, double x) {
#ifdef SOFTROD_DIAG_LIBM_TRIG
    return exp(x);
#endif
    if (__builtin_expect(!wave_any(!(fabs(x) < 1.0e-3)), 1))      // (the hint keeps the in-range path the fall-through)
        return fma(x, fma(x, fma(x, fma(x, K.e4, K.e3), 0.5), 1.0), 1.0);
    int k = 0;
    while (wave_any(!(fabs(x) < 1.0e-3)) && k < 24) { x *= 0.5; ++k; }
    double e = fma(x, fma(x, fma(x, fma(x, 1.0 / 24.0, 1.0 / 6.0), 0.5), 1.0), 1.0);
    for (; k > 0; --k) e *= e;
    return e;
}

// Wave-uniform: true if the whole rod (and its forcing) is planar in the sense above.
template <int EPL>
__device__ __forceinline__ bool planar_from_lane(const RodParams& P, const BcTargets& B, int lane,
                                                 const LaneN<EPL>& L, PlanarN<EPL>& Z) {
    const int n = P.n_elem;
    bool ok = (P.gravity[2] == 0.0) && (B.pos[2] == 0.0) && (B.Q[2] == 0.0) && (B.Q[8] == 0.0);
    // d2_z = -(cos^2 + sin^2) as straight_rod's cross product rounds it: +-1 to an ulp; the
    // planar update leaves row 1 of Q untouched, exactly like the 3-D one (R4 = 1)
    Z.s2 = __shfl((L.Q[0][5] < 0.0) ? -1.0 : 1.0, 0);
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        const int idx = lane * EPL + s;
        const double* Q = L.Q[s];
        const double s2 = (Q[5] < 0.0) ? -1.0 : 1.0, c = Q[6], sn = Q[7];
        const bool node_ok = (L.x[s][2] == 0.0) && (L.v[s][2] == 0.0);
        const bool elem_ok = (Q[2] == 0.0) && (Q[3] == 0.0) && (Q[4] == 0.0) && (Q[8] == 0.0) && (s2 == Z.s2) &&
                             (fabs(fabs(Q[5]) - 1.0) <= 1.0e-12) && (L.w[s][0] == 0.0) && (L.w[s][2] == 0.0) &&
                             (fabs(fma(s2, sn, Q[0])) <= 1.0e-12) && (fabs(fma(-s2, c, Q[1])) <= 1.0e-12) &&
                             (fabs(fma(c, c, fma(sn, sn, -1.0))) <= 1.0e-12);
        ok = ok && (idx > n || node_ok) && (idx >= n || elem_ok);
        Z.x[s][0] = L.x[s][0]; Z.x[s][1] = L.x[s][1];
        Z.v[s][0] = L.v[s][0]; Z.v[s][1] = L.v[s][1];
        Z.c[s] = c; Z.s[s] = sn;
        Z.wz[s] = (idx < n) ? Z.s2 * L.w[s][1] : 0.0;
        Z.t[s][0] = L.t[s][0]; Z.t[s][1] = L.t[s][1];
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        double a[EPL], o[EPL], av[EPL], ov[EPL];
#pragma unroll
        for (int s = 0; s < EPL; ++s) { a[s] = Z.x[s][c]; av[s] = Z.v[s][c]; }
        shift_next<EPL>(a, o);
        shift_next<EPL>(av, ov);
#pragma unroll
        for (int s = 0; s < EPL; ++s) { Z.d[s][c] = o[s] - a[s]; Z.dv[s][c] = ov[s] - av[s]; }
    }
    {
        double cnx[EPL], snx[EPL];
        shift_next<EPL>(Z.c, cnx);
        shift_next<EPL>(Z.s, snx);
#pragma unroll
        for (int s = 0; s < EPL; ++s) {
            const bool vor_valid = (lane * EPL + s) < n - 1;
            const double sinD = fma(snx[s], Z.c[s], -cnx[s] * Z.s[s]);
            const double cosD = fma(cnx[s], Z.c[s], snx[s] * Z.s[s]);
            Z.dl[s] = vor_valid ? atan2(sinD, cosD) : 0.0;
        }
    }
#ifdef SOFTROD_DIAG_NO_PLANAR
    return false;
#endif
    return !__any(!ok);
}

template <int EPL>
__device__ __forceinline__ void planar_to_lane(const PlanarN<EPL>& Z, LaneN<EPL>& L) {
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        L.x[s][0] = Z.x[s][0]; L.x[s][1] = Z.x[s][1];
        L.v[s][0] = Z.v[s][0]; L.v[s][1] = Z.v[s][1];
        L.Q[s][0] = -Z.s2 * Z.s[s]; L.Q[s][1] = Z.s2 * Z.c[s];
        L.Q[s][6] = Z.c[s]; L.Q[s][7] = Z.s[s];
        L.w[s][1] = Z.s2 * Z.wz[s];
        L.t[s][0] = Z.t[s][0]; L.t[s][1] = Z.t[s][1]; L.t[s][2] = 0.0;
    }
}

// What a planar step changes, written to the resident rows: x, y, v_x, v_y, d1 = (-s2 s, s2 c), d3 = (c, s),
// omega_2 and the tangents — 12 of the rod's 21 rows.  The other nine (z components, d2, the
// out-of-plane directors and rates) hold the exact values planar_from_lane has just checked.
template <int EPL>
__device__ __forceinline__ void planar_store(const StatePtrs& S, size_t N, int rod, int lane, const PlanarN<EPL>& Z) {
    constexpr size_t W = (size_t)kLanes * EPL;
    const size_t base = (size_t)rod * W + (size_t)lane * EPL;
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            S.pos[c * N * W + base + s] = Z.x[s][c];
            S.vel[c * N * W + base + s] = Z.v[s][c];
            S.tan[c * N * W + base + s] = Z.t[s][c];
        }
        S.tan[2 * N * W + base + s] = 0.0;
        S.dir[0 * N * W + base + s] = -Z.s2 * Z.s[s];
        S.dir[1 * N * W + base + s] = Z.s2 * Z.c[s];
        S.dir[6 * N * W + base + s] = Z.c[s];
        S.dir[7 * N * W + base + s] = Z.s[s];
        S.omg[1 * N * W + base + s] = Z.s2 * Z.wz[s];
    }
}

// kinematic_n with a = (0, h w, 0): R0 = R8 = cos, R6 = -R2 = sin, R4 = 1, the rest 0, so
// the new d3 = sin * d1 + cos * d3 — the rotation of (c, s) about z by h wz.
template <int EPL>
__device__ __forceinline__ void planar_kinematic_n(double h, const double (&hq_h)[EPL], const ConstN<EPL>& C,
                                                   const PlanarC<EPL>& K, PlanarN<EPL>& Z) {
    double ra[EPL], ran[EPL];      // rotation angle of each element about z
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        const double hp = h * C.hx[s];
        Z.x[s][0] = fma(hp, Z.v[s][0], Z.x[s][0]);
        Z.x[s][1] = fma(hp, Z.v[s][1], Z.x[s][1]);
        // the edge vectors move with the velocity differences (no node is held in x or y by a
        // position constraint in this feature set: hx = 1), so the next force evaluation needs
        // no neighbour positions
        Z.d[s][0] = fma(h, Z.dv[s][0], Z.d[s][0]);
        Z.d[s][1] = fma(h, Z.dv[s][1], Z.d[s][1]);
        const double a = hq_h[s] * Z.wz[s];
        ra[s] = a;
        const double t = a * a;
        double sc, cs;
#ifdef SOFTROD_DIAG_LIBM_TRIG
        cs = cos(a);
        const double sn = sin(a);
        sc = 0.0;
        (void)sc;
#else
        if (__builtin_expect(!wave_any(t >= 1.0e-3), 1)) {      // sinc_cosc's range, on the opaque coefficients; the
            sc = fma(t, fma(t, fma(t, K.s3, K.s2), K.s1), 1.0);       // cosine directly (t^4/8! < 3e-17)
            cs = fma(t, fma(t, fma(t, K.c3, K.c2), -0.5), 1.0);
        } else {
            double cc;
            sinc_cosc<false>(t, sc, cc);
            cs = fma(-cc, t, 1.0);
        }
        const double sn = sc * a;
#endif
        // both products of the old c first, so that c and s are then updated in place
        const double p = cs * Z.c[s], q = sn * Z.c[s];
        Z.c[s] = fma(-sn, Z.s[s], p);
        fma_inplace(Z.s[s], cs, q);
    }
    // invalid slots turn by exactly 0 (hq wz = 0 there), so the angle of the vertex after the
    // last element only collects a finite value that its zero stiffness never lets out
    shift_next<EPL>(ra, ran);
#pragma unroll
    for (int s = 0; s < EPL; ++s) Z.dl[s] += ran[s] - ra[s];
#ifdef SOFTROD_DIAG_RECOMPUTE_EDGES
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        double a[EPL], o[EPL];
#pragma unroll
        for (int s = 0; s < EPL; ++s) a[s] = Z.x[s][c];
        shift_next<EPL>(a, o);
#pragma unroll
        for (int s = 0; s < EPL; ++s) Z.d[s][c] = o[s] - a[s];
    }
#endif
#ifdef SOFTROD_DIAG_RECOMPUTE_ANGLE
    {
        double cnx[EPL], snx[EPL];
        shift_next<EPL>(Z.c, cnx);
        shift_next<EPL>(Z.s, snx);
#pragma unroll
        for (int s = 0; s < EPL; ++s) {
            const double sinD = fma(snx[s], Z.c[s], -cnx[s] * Z.s[s]);
            const double cosD = fma(cnx[s], Z.c[s], snx[s] * Z.s[s]);
            Z.dl[s] = atan2(sinD, cosD);      // (finite on invalid slots; their stiffness is zero)
        }
    }
#endif
}

// dynamic_n for SOFTROD_FEATURES_SOFTPENDULUM (gravity and the point force live in C.ca,
// the analytical damper is fused, the pendulum constraint pins v_y of node 0).
template <int EPL>
__device__ __forceinline__ void planar_dynamic_n(const RodParams& P, const ConstN<EPL>& C,
                                                 const PlanarC<EPL>& K, int lane, PlanarN<EPL>& Z) {
    double d[EPL][2];
    double len[EPL], il[EPL];
    double qt0[EPL], qt2[EPL], np0[EPL], np2[EPL], cs[EPL][2], f[EPL][2], tq[EPL];
    // geometry and shear/stretch in the frame (d1', d3) with d1' = s2 d1 = (-s, c): Q t has no
    // d2 component, so n = S (Q t - z/e) has none; qt0, np0 are s2 times the 3-D ones
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        d[s][0] = Z.d[s][0];
        d[s][1] = Z.d[s][1];
        // slots past the last element have d = 0: clamped, they stay finite and their zero
        // stiffnesses keep them out of every sum
        const double dd = fmax(fma(d[s][1], d[s][1], d[s][0] * d[s][0]), 1.0e-20);
        const double r = rsqrt3(dd);
        len[s] = fma(dd, r, K.eps_length);
        il[s] = fma(-K.eps_length * r, r, r);
        Z.t[s][0] = d[s][0] * il[s];
        Z.t[s][1] = d[s][1] * il[s];
        qt0[s] = fma(Z.c[s], Z.t[s][1], -Z.s[s] * Z.t[s][0]);
        qt2[s] = fma(Z.s[s], Z.t[s][1], Z.c[s] * Z.t[s][0]);
        np0[s] = C.s01[s] * qt0[s];
        np2[s] = C.s2[s] * fma(-K.rest_len, il[s], qt2[s]);
        cs[s][0] = fma(Z.c[s], np2[s], -Z.s[s] * np0[s]);
        cs[s][1] = fma(Z.s[s], np2[s], Z.c[s] * np0[s]);
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        double a[EPL], o[EPL];
#pragma unroll
        for (int s = 0; s < EPL; ++s) a[s] = cs[s][c];
        shift_prev<EPL>(a, o);
#pragma unroll
        for (int s = 0; s < EPL; ++s) f[s][c] = cs[s][c] - o[s];
    }
    // bend: kappa = s2 D / D^ with the carried bending angle D (see the header); the Voronoi
    // couple B kappa / vd^3 about z, and kappa x B kappa = 0 for a single component
    double len_n[EPL], up[EPL];
    shift_next<EPL>(len, len_n);
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        const double rl = rcp3(len_n[s] + len[s]);          // vd = (l + l+) / (2 rest_vor), folded into bk
        // 1 - eps_sin / theta with theta^2 = D^2 + 2 acos_shift (see the header)
#ifdef SOFTROD_DIAG_NO_EPS_SIN
        const double bg = K.bk[s];
#else
        const double bg = fma(K.bke[s], __builtin_amdgcn_rsq(fma(Z.dl[s], Z.dl[s], K.two_shift)), K.bk[s]);
#endif
        up[s] = (bg * Z.dl[s]) * (rl * rl * rl);
    }
    {
        double o[EPL];
        shift_prev<EPL>(up, o);
#pragma unroll
        for (int s = 0; s < EPL; ++s) tq[s] = up[s] - o[s];
    }
    // shear couple (component 1 of (Q t) x n), unsteady dilatation; the transport term
    // (J w) x w vanishes for w = (0, w, 0)
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        tq[s] = fma(len[s], fma(qt2[s], np0[s], -qt0[s] * np2[s]), tq[s]);
        // (J / e) (de/dt / e) = J rest_len (t . dv) / l^2
        const double tdv = fma(Z.t[s][1], Z.dv[s][1], Z.t[s][0] * Z.dv[s][0]);
        tq[s] = fma((K.jr * il[s] * il[s]) * tdv, Z.wz[s], tq[s]);
    }
    // rate update fused with the analytical damper; constrain_rates (v_y of node 0) is the zero
    // in cfy / cay
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        Z.v[s][0] = fma(C.cf[s], f[s][0], fma(K.damp_t, Z.v[s][0], C.ca[s][0]));
        Z.v[s][1] = fma(K.cfy[s], f[s][1], fma(K.damp_t, Z.v[s][1], K.cay[s]));
        const double w = fma(K.cwl[s] * len[s], tq[s], Z.wz[s]);
        Z.wz[s] = w * exp_one<EPL>(K, K.xl[s] * len[s]);
    }
    // velocity differences of the new velocities: for the edge vectors' step and the next
    // dilatation rate
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        double av[EPL], ov[EPL];
#pragma unroll
        for (int s = 0; s < EPL; ++s) av[s] = Z.v[s][c];
        shift_next<EPL>(av, ov);
#pragma unroll
        for (int s = 0; s < EPL; ++s) Z.dv[s][c] = ov[s] - av[s];
    }
}

}  // namespace softrod
