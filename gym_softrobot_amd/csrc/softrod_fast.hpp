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
//     checks (one s_cbranch on the fast path); outside the range the trigonometric arguments are
//     halved until they fit and the result rebuilt with double-angle identities, the exponential is
//     reduced by powers of two (exp_wide_pair) — no libm calls, which keeps the register budget small
//     enough for several waves per SIMD;
//     the damper's c_r^e is evaluated as c_r * c_r^(e-1) so that only the strain e-1, not
//     log c_r, has to be small;
//   * divisions become one Newton-refined v_rcp_f64 / v_rsq_f64 each; circular cross
//     sections (I1 = I2, always true for CosseratRod.straight_rod) are exploited;
//   * lane-validity masks are folded into per-lane stiffness / time-step constants;
//   * everything is written over EPL slots per lane (softrod_kernels.hpp), so the same
//     code serves rods of up to 63 (EPL = 1) and 126 (EPL = 2) elements;
//   * SoftPendulum's rod is planar: its instantiation runs softrod_planar.hpp and keeps the
//     general loop as an out-of-line cold call for states that are not.
// All of these are ulp-level reorderings; tests/test_gpu_parity.py holds the kernel to
// the same rtol 1e-5 against the oracle as the LIBM kernel.
#pragma once

namespace softrod {

// The loops are bound by VALU issue, and the compiler spends instructions on register copies that
// a three-address v_fma_f64 does not need: it prefers the two-address v_fmac_f64 (result over the
// ADDEND) and then copies — a Horner step with its coefficient in a register becomes v_mov_b64 +
// v_fmac, an update x <- m x + a lands over a and is copied back to x at the loop's end.  These
// wrappers pin the form (same operation, same rounding).
//   horner(g, y, c)        g y + c with the coefficient in a vector register (as a literal in scalar
//                          registers it cost the SoftPendulum kernel its scalar budget: 32
//                          v_readlane per substep of spilled loop constants)
//   fma_over(d, a, b, c)   d <- a b + c, where d's old value is dead (its register is reused)
//   fma_inplace_u(x, m, a) x <- m x + a with a wave-uniform m
// "does any lane …": the ballot itself.  (__any goes through an integer — v_cndmask 0/1, v_cmp_ne —
// before it branches: two more VALU instructions per range check.)
__device__ __forceinline__ bool wave_any(bool p) { return __builtin_amdgcn_ballot_w64(p) != 0ull; }
// "... any VALID lane": the validity as a lane mask ANDed on the scalar side.  (`valid && cond` as one
// predicate makes the compiler rebuild it through v_cndmask 0/1 + v_cmp_ne before the ballot.)
__device__ __forceinline__ bool wave_any_of(unsigned long long lanes, bool p) {
    return (__builtin_amdgcn_ballot_w64(p) & lanes) != 0ull;
}

__device__ __forceinline__ double horner(double g, double y, double c) {
    double d;
    asm("v_fma_f64 %0, %1, %2, %3" : "=v"(d) : "v"(g), "v"(y), "v"(c));
    return d;
}
__device__ __forceinline__ void fma_over(double& d, double a, double b, double c) {
    asm("v_fma_f64 %0, %1, %2, %3" : "+v"(d) : "v"(a), "v"(b), "v"(c));
}
__device__ __forceinline__ void fma_inplace_u(double& x, double m, double a) {
    asm("v_fma_f64 %0, %1, %0, %2" : "+v"(x) : "s"(m), "v"(a));
}

// sin(th)/th and (1-cos(th))/th^2 from t = th^2.  t < 1e-3: degree-3 Taylor in t
// (remainders t^4/9! < 3e-18, t^4/10! < 3e-19).  Otherwise (|omega| dt > 0.03 rad: only
// when a simulation is blowing up) the angle is halved k times and rebuilt with
//   sinc(2p) = sinc(p) cos(p),  cosc(2p) = sinc(p)^2 / 2,  cos(p) = 1 - cosc(p) p^2.
// (LEAN = false: plain fma, for the one place — the planar loop's out-of-range tier — where the
// scalar registers of the literals are not to spare)
template <bool LEAN = true>
__device__ __forceinline__ void sinc_cosc(double t, double& sc, double& cc) {
    if (!wave_any(t >= 1.0e-3)) {     // the whole wave is in range: straight-line, one s_cbranch
        if constexpr (LEAN) {
            sc = fma(t, horner(horner(-1.0 / 5040.0, t, 1.0 / 120.0), t, -1.0 / 6.0), 1.0);
            cc = fma(t, horner(horner(-1.0 / 40320.0, t, 1.0 / 720.0), t, -1.0 / 24.0), 0.5);
        } else {
            sc = fma(t, fma(t, fma(t, -1.0 / 5040.0, 1.0 / 120.0), -1.0 / 6.0), 1.0);
            cc = fma(t, fma(t, fma(t, -1.0 / 40320.0, 1.0 / 720.0), -1.0 / 24.0), 0.5);
        }
        return;
    }
    // up to 4^16 * 1e-3: more than 250 rad per (sub)step — beyond that the state is garbage on
    // its way to NaN and only the cost of getting there matters
    int k = 0;
    while (wave_any(t >= 1.0e-3) && k < 16) { t *= 0.25; ++k; }   // wave-uniform trip count
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
//   asin(sqrt y)/(sqrt y sqrt(1-y)) = sum_k c_k y^k,  c_k = (2k)!!/(2k+1)!! = c_{k-1} 2k/(2k+1).
// The series is truncated where its remainder c_K y^K/(1-y) drops below 1e-17, in three
// wave-uniform tiers (the trip from one tier to the next costs one s_cbranch):
//   y < 2.5e-3 (neighbouring elements < 0.1 rad apart, a 50-element rod at rest):  6 terms
//   y < 0.04   (< 0.40 rad: the falling / whipping pendulum):                      12 terms
//   y < 0.15   (< 0.79 rad: 10-element octopus arms at full curl):                 20 terms
// Beyond that two half-angle steps y' = (1 - sqrt(1-y)) / 2 bring any y <= 1 into the last tier and
// theta/sin(theta) = (phi/sin phi) / (cos(theta/2) cos(theta/4)).
template <int K>
__device__ __forceinline__ double theta_over_sin_series(double y) {
    // c_k by the recurrence, folded at compile time
    double c[K];
    c[0] = 1.0;
#pragma unroll
    for (int k = 1; k < K; ++k) c[k] = c[k - 1] * (double)(2 * k) / (double)(2 * k + 1);
    double g = c[K - 1];
#pragma unroll
    for (int k = K - 2; k >= 1; --k) g = horner(g, y, c[k]);
    return fma(g, y, 1.0);       // c[0] = 1 is an inline constant
}

// BATCHED: the three range tests are issued together and branched on afterwards — one VALU -> SALU round trip instead of three on
// the way to the last tier, two compares more on the way to the first.  The rigid-body kernel's arms (ten elements curled to
// 0.8-1.8 rad per joint under the benchmark's actions) take the last tier in 90 % of their substeps: -1.0 % of the OctoFlat
// env.step (profiles/r6_octo_ab.txt, "cmp3"); the one-rod kernels live in the first tier and keep the sequential tests.
template <bool BATCHED = false>
__device__ __forceinline__ double theta_over_sin(double y, bool valid_lane) {
    const unsigned long long valid = __builtin_amdgcn_ballot_w64(valid_lane);     // (loop-invariant: two scalar registers)
    if constexpr (BATCHED) {
        const unsigned long long m1 = __builtin_amdgcn_ballot_w64(!(y < 2.5e-3)) & valid;
        const unsigned long long m2 = __builtin_amdgcn_ballot_w64(!(y < 0.04)) & valid;
        const unsigned long long m3 = __builtin_amdgcn_ballot_w64(!(y < 0.15)) & valid;
        if (m1 == 0) return theta_over_sin_series<6>(y);
        if (m2 == 0) return theta_over_sin_series<12>(y);
        if (m3 == 0) return theta_over_sin_series<20>(y);
    } else {
        if (!wave_any_of(valid, !(y < 2.5e-3))) return theta_over_sin_series<6>(y);
        if (!wave_any_of(valid, !(y < 0.04))) return theta_over_sin_series<12>(y);
#ifdef SOFTROD_DIAG_THETA_LOOP      // round 5's tier, for the A/B of tools/octo_ab.sh (variant "thetaloop")
        {
            const double y0 = y;
            int k = 0;
            while (wave_any_of(valid, !(y < 0.15)) && k < 12) {
                const double om = fmax(1.0 - y, 1.0e-300);
                y = 0.5 * y * fast_rcp(1.0 + om * fast_rsqrt(om));
                ++k;
            }
            double g = theta_over_sin_series<20>(y);
            if (k > 0) {
                const double a = fmax(y * (1.0 - y), 1.0e-300), b = fmax(y0 * (1.0 - y0), 1.0e-300);
                g *= (double)(1 << k) * (a * fast_rsqrt(a)) * fast_rsqrt(b);
                g = (y0 < 1.0e-30) ? 1.0 : g;
            }
            return g;
        }
#endif
        if (!wave_any_of(valid, !(y < 0.15))) return theta_over_sin_series<20>(y);
    }
    // Beyond: two half-angle steps, unconditionally — y <= 1 means sin^2(theta/8) <= sin^2(pi/8) = 0.1465 < 0.15, so two always
    // suffice and no wave-uniform loop (a VALU -> SALU round trip per test) is needed.
    //   cos(theta/2) = sqrt(1 - y),  sin^2(theta/4) = (1 - cos(theta/2)) / 2,
    //   theta/sin(theta) = (phi/sin(phi)) / (cos(theta/2) cos(theta/4)),  phi = theta/4.
    // The subtraction cancels for a straight joint in a wave that has a curled one, harmlessly: the series is 1 + (2/3) y' + ...,
    // so an ABSOLUTE error of 1e-16 in y' is a relative 1e-16 in the result; the two cosines enter as the refined 1/sqrt the
    // half-angle steps compute anyway.  (Until round 6: a ballot loop of up to 12 trips with a division per trip and two more
    // square roots to rebuild sin(phi)/sin(theta) — twice the instructions, in one dependent chain; OctoFlat 9.39 -> 8.84 ms.)
    const double om0 = fmax(1.0 - y, 1.0e-300);
    const double r0 = fast_rsqrt(om0);
    const double y1 = fma(-0.5 * om0, r0, 0.5);
    const double om1 = 1.0 - y1;                       // >= 1/2
    const double r1 = fast_rsqrt(om1);
    const double y2 = fma(-0.5 * om1, r1, 0.5);
    return theta_over_sin_series<20>(y2) * (r0 * r1);
}

// The reference divides by sin(theta + 1e-14) (PyElastica _inv_rotate; softrod_config.eps_sin):
//   theta / sin(theta + d) = (theta / sin theta) (1 - d cot theta + O(d^2)),
//   cot theta = (1 - 2y) / (2 sqrt(y (1 - y))).
// For a nearly straight joint theta is the 1.4e-5 that the `- 1e-10` inside arccos leaves, so the
// term is a relative 7e-10 on kappa — a systematic change of the bending stiffness, not a
// rounding: without it the stabilised inverted pendulum leaves 1e-5 after 89 env.steps instead of
// the 107 that two roundings of the same algorithm reach (profiles/r3_fastmath_cost.json).  The
// raw 2^-23 hardware seed of 1/sqrt is enough (the term itself is < 1e-9).
__device__ __forceinline__ double eps_sin_factor(double y, double eps_sin) {
#ifdef SOFTROD_DIAG_NO_EPS_SIN
    return 1.0;
#endif
    const double rho = __builtin_amdgcn_rsq(fmax(fma(-y, y, y), 1.0e-300));      // 2 / sin(theta)
    return fma(fma(y, eps_sin, -0.5 * eps_sin), rho, 1.0);
}

// A double literal that is materialised WHERE IT IS USED: two s_mov_b32 in a volatile asm.  The out-of-range tiers sit inside the
// substep loops; left to the compiler, their sixteen 64-bit literals are hoisted in front of the loop as scalar register
// pairs, stay live across it, and the allocator spills the HOT path's scalars instead (round 6: 32 v_readlane_b32 per
// planar substep, 99 -> 131 VALU, caught by tests/test_codegen.py).  A volatile asm cannot be hoisted.
template <unsigned long long BITS>
__device__ __forceinline__ double cold_literal() {
    int lo, hi;
    asm volatile("s_mov_b32 %0, %2\n\ts_mov_b32 %1, %3"
                 : "=s"(lo), "=s"(hi)
                 : "i"((int)(unsigned)(BITS & 0xffffffffull)), "i"((int)(unsigned)(BITS >> 32)));
    return __hiloint2double(hi, lo);
}
#ifdef SOFTROD_DIAG_HOIST_LITERALS      // A/B (tools/octo_ab.sh "hoistlit"): plain literals, the compiler's placement
#define SOFTROD_COLD_LIT(x) ((double)(x))
#else
#define SOFTROD_COLD_LIT(x) cold_literal<__builtin_bit_cast(unsigned long long, (double)(x))>()
#endif

// exp(x0), exp(x2) for any x: exp(x) = 2^n exp(r), n = rint(x log2 e), r = x - n ln 2 with ln 2 in two parts (fdlibm's split:
// n ln2_hi is exact for |n| < 2^20), |r| <= ln 2 / 2 = 0.347, degree-13 Taylor (remainder r^14 / 14! < 5e-18), v_ldexp_f64.
// ~21 VALU instructions each, no loop, relative error ~1e-16 plus the argument's own rounding; the two chains share every
// literal (one scalar pair live at a time) and interleave.
__device__ __forceinline__ void exp_wide_pair(double x0, double x2, double& e0, double& e2) {
    double c = SOFTROD_COLD_LIT(1.4426950408889634);
    const double n0 = rint(x0 * c), n2 = rint(x2 * c);
    c = SOFTROD_COLD_LIT(6.93147180369123816490e-01);
    double r0 = fma(-n0, c, x0), r2 = fma(-n2, c, x2);
    c = SOFTROD_COLD_LIT(1.90821492927058770002e-10);
    r0 = fma(-n0, c, r0);
    r2 = fma(-n2, c, r2);
    double p0 = SOFTROD_COLD_LIT(1.0 / 6227020800.0), p2 = p0;
#define SOFTROD_EXP_TERM(k)                                   \
    c = SOFTROD_COLD_LIT(1.0 / (k));                          \
    p0 = fma(p0, r0, c);                                      \
    p2 = fma(p2, r2, c);
    SOFTROD_EXP_TERM(479001600.0)
    SOFTROD_EXP_TERM(39916800.0)
    SOFTROD_EXP_TERM(3628800.0)
    SOFTROD_EXP_TERM(362880.0)
    SOFTROD_EXP_TERM(40320.0)
    SOFTROD_EXP_TERM(5040.0)
    SOFTROD_EXP_TERM(720.0)
    SOFTROD_EXP_TERM(120.0)
    SOFTROD_EXP_TERM(24.0)
    SOFTROD_EXP_TERM(6.0)
#undef SOFTROD_EXP_TERM
    p0 = fma(p0, r0, 0.5);
    p2 = fma(p2, r2, 0.5);
    p0 = fma(p0, r0, 1.0);
    p2 = fma(p2, r2, 1.0);
    p0 = fma(p0, r0, 1.0);
    p2 = fma(p2, r2, 1.0);
    c = SOFTROD_COLD_LIT(1100.0);
    e0 = ldexp(p0, (int)fmin(fmax(n0, -c), c));
    e2 = ldexp(p2, (int)fmin(fmax(n2, -c), c));
}

// exp(x) for the damper: |x| < 1e-3 -> degree-4 Taylor (remainder x^5/120 < 1e-17), one s_cbranch on the fast path;
// larger |x| — strong damping constants, and above all the thin end of a TAPERED arm, whose log c_r = -nu dt m / J
// reaches -1500: (e - 1) log c_r is 10 .. 700 in every substep of a stretched arm — take exp_wide_pair.  (Until round 6 this
// tier halved the argument until it fitted and squared the result back: up to 21 wave-uniform loop trips and a
// dependent chain of 42 multiplications per substep, 2.7 of the 4.5 ms of an OctoArmPush-v0 env.step.)
__device__ __forceinline__ void exp_pair(double x0, double x2, bool valid_lane, double& e0, double& e2) {
    const unsigned long long valid = __builtin_amdgcn_ballot_w64(valid_lane);
    if (!wave_any_of(valid, !(fmax(fabs(x0), fabs(x2)) < 1.0e-3))) {
        e0 = fma(x0, fma(x0, fma(x0, horner(1.0 / 24.0, x0, 1.0 / 6.0), 0.5), 1.0), 1.0);
        e2 = fma(x2, fma(x2, fma(x2, horner(1.0 / 24.0, x2, 1.0 / 6.0), 0.5), 1.0), 1.0);
        return;
    }
    // A middle tier: |x| < 0.04 — the strains of ordinary arms under ordinary damping (the octopus arm's log c_r = -0.057
    // times a strain of a few per cent; SoftPendulum3D's uniform damper) — by the degree-8 Taylor polynomial (remainder
    // x^9 / 9! < 8e-19), its literals materialised where they are used like exp_wide_pair's.  exp_wide_pair costs ~50
    // instructions whatever the argument; the halving loop it replaced took 6 for an argument of 2e-3.  Measured (one box,
    // alternating runs): OctoFlat 8.99 -> 8.48 ms, SoftPendulum3D 1.675 -> 1.62, the 100-element arm 5.82 -> 5.78.  (A wider
    // FIRST tier for the rigid-body kernels — degree 6 up to 6e-3 — was measured too: -0.9 % curled, +0.7 % at rest; not adopted.
    // The same idea for sinc_cosc — a degree-5 middle tier up to t = 0.05 — is SLOWER everywhere, OctoFlat 8.47 -> 8.71 ms: its tier
    // is rarely taken and the extra literals cost the loop scalar registers.)
    if (!wave_any_of(valid, !(fmax(fabs(x0), fabs(x2)) < 0.04))) {
        double c = SOFTROD_COLD_LIT(1.0 / 40320.0);
        double p0 = c, p2 = c;
#define SOFTROD_EXP_TERM(k)                                   \
        c = SOFTROD_COLD_LIT(1.0 / (k));                      \
        p0 = fma(p0, x0, c);                                  \
        p2 = fma(p2, x2, c);
        SOFTROD_EXP_TERM(5040.0)
        SOFTROD_EXP_TERM(720.0)
        SOFTROD_EXP_TERM(120.0)
        SOFTROD_EXP_TERM(24.0)
        SOFTROD_EXP_TERM(6.0)
#undef SOFTROD_EXP_TERM
        p0 = fma(p0, x0, 0.5); p2 = fma(p2, x2, 0.5);
        p0 = fma(p0, x0, 1.0); p2 = fma(p2, x2, 1.0);
        e0 = fma(p0, x0, 1.0); e2 = fma(p2, x2, 1.0);
        return;
    }
    exp_wide_pair(x0, x2, e0, e2);
}

}  // namespace softrod
#include "softrod_planar.hpp"
namespace softrod {

#ifndef SOFTROD_IDLE_LANES_EXEC_MASK     // the 3-D loop's idle lanes sit it out (0: they execute on zeros)
#define SOFTROD_IDLE_LANES_EXEC_MASK 1
#endif

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
            // each new entry's last product lands in the register of the entry it replaces (row 2
            // last: its own old value is a factor), so Q needs no copy at the loop's end
            const double b0 = L.Q[s][j], b1 = L.Q[s][3 + j];
            const double p0 = fma(R1, b1, R0 * b0), p1 = fma(R4, b1, R3 * b0), p2 = fma(R7, b1, R6 * b0);
            fma_over(L.Q[s][j], R2, L.Q[s][6 + j], p0);
            fma_over(L.Q[s][3 + j], R5, L.Q[s][6 + j], p1);
            fma_inplace(L.Q[s][6 + j], R8, p2);
        }
    }
}

// Connections (joints) plug in between the internal loads and the contact operator, in
// the registration order of build_octopus (octopus/build.py:117-200).
struct NoConnections {
    template <int EPL>
    __device__ __forceinline__ void operator()(double (&)[EPL][3], double (&)[EPL][3], const LaneN<EPL>&,
                                               const double (&)[EPL][3]) const {}
};

// ---- forces, torques, rate update, dampers, constrain_rates -------------------------------------
template <unsigned F, int EPL, bool TAPER = false, class Connections = NoConnections>
__device__ __forceinline__ void dynamic_n(const RodParams& P, const ConstN<EPL>& C, const BcTargets& B,
                                          int lane, LaneN<EPL>& L, Connections&& connect = Connections()) {
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
        const bool elem_valid = slot_local<F>(P, lane * EPL + s) < n;
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
    double kv[EPL][3], e3v[EPL];      // kappa (zero off the rod) and 1 / eps^3 for the muscle layers (dead otherwise)
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        const bool vor_valid = slot_local<F>(P, lane * EPL + s) < n - 1;
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
        const double gk = theta_over_sin<(F != kRuntimeFeatures && (F & SOFTROD_FEAT_OCTO_HEAD) != 0)>(y, vor_valid) * eps_sin_factor(y, P.eps_sin) * (-0.5 * P.inv_rest_vor);
        const double k0 = vec0 * gk, k1 = vec1 * gk, k2 = vec2 * gk;
        const double vd = (len_n[s] + len[s]) * (0.5 * P.inv_rest_vor);
        const double rvd = fast_rcp(vd);
        const double e3 = rvd * rvd * rvd;
        if constexpr (kMusclesCompiled<F>) {
            kv[s][0] = vor_valid ? k0 : 0.0; kv[s][1] = vor_valid ? k1 : 0.0; kv[s][2] = vor_valid ? k2 : 0.0;
            e3v[s] = e3;
        }
        if (F == kRuntimeFeatures || (F & (SOFTROD_FEAT_REST_KAPPA_ACTION | SOFTROD_FEAT_SPLINE_MUSCLE_TORQUES))) {
            L.kap[s][0] = k0; L.kap[s][1] = k1; L.kap[s][2] = k2;
        } else if constexpr (kMusclesCompiled<F> && (F & SOFTROD_FEAT_OCTO_HEAD) != 0) {
            L.kap[s][0] = k0;        // rod.kappa[0]: what the muscle octopus envs' get_state reads (softrod_mocto.hpp)
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
        const double j01 = (TAPER ? C.j01[s] : P.J[0]) * ie[s], j2 = (TAPER ? C.j2[s] : P.J[2]) * ie[s];
        const double z = w[2] * (j01 - j2);
        tq[s][0] = fma(w[1], z, tq[s][0]);
        tq[s][1] = fma(-w[0], z, tq[s][1]);
        const double js01 = j01 * sdil, js2 = j2 * sdil;
        tq[s][0] = fma(js01, w[0], tq[s][0]);
        tq[s][1] = fma(js01, w[1], tq[s][1]);
        tq[s][2] = fma(js2, w[2], tq[s][2]);
    }
    // forcing on the torques: the two spline muscles add their profiles to external_torques
    // (compute_muscle_torques, muscle_torques_with_bspline.py:199-201), normal then binormal
    if (has<F>(P, SOFTROD_FEAT_SPLINE_MUSCLE_TORQUES)) {
        if (L.mflag & 3) spline_muscle_rebuild<EPL>(P, lane, len, L);
#pragma unroll
        for (int s = 0; s < EPL; ++s) { tq[s][0] += L.rk[s][0]; tq[s][1] += L.rk[s][1]; }
    }
    // forcing: the COOMM muscle layers add their equivalent loads to external_forces / external_torques
    // (ApplyMuscles, arm_push_env.py:208-212; softrod_muscle.hpp)
    // Compiled into the instantiations FOR a muscle feature set only (kMusclesCompiled): inside the run-time-mask
    // instantiation the block's live ranges tripled the scratch of every other feature mix (824 -> 2496 B per lane);
    // softrod_create sends other mixes with muscles to the LIBM kernel.
    if constexpr (kMusclesCompiled<F>) {
        double r0s[EPL];
#pragma unroll
        for (int s = 0; s < EPL; ++s) r0s[s] = TAPER ? C.r0s[s] : P.r0_sqrt_rest_len;
        muscle_loads_n<EPL, true, true>(P, C, lane, L, e, il, qt, kv, e3v, r0s, f, tq);
    }
    connect(f, tq, L, xn);
    // plane contact
    if (has<F>(P, SOFTROD_FEAT_PLANE_CONTACT_ANISO)) {
        double Fg[EPL][3], fc[EPL][3];
#pragma unroll
        for (int s = 0; s < EPL; ++s) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                Fg[s][c] = f[s][c] + C.gm[s][c];     // (a per-lane constant: no select, no scalar operand)
            }
        }
        if constexpr (SOFTROD_OCTO_CONTACT_LDS && F != kRuntimeFeatures && (F & SOFTROD_FEAT_OCTO_HEAD) != 0 && (F & kFeatPlaneZup) != 0)
            plane_contact_n<EPL, true, true, TAPER>(contact_params_lds(P), P, lane, C, L, xn, vn, len, Fg, tq, fc);
        else if (has<F>(P, kFeatPlaneZup))
            plane_contact_n<EPL, true, true, TAPER>(contact_params(P), P, lane, C, L, xn, vn, len, Fg, tq, fc);
        else
            plane_contact_n<EPL, false, true, TAPER>(contact_params(P), P, lane, C, L, xn, vn, len, Fg, tq, fc);
#pragma unroll
        for (int s = 0; s < EPL; ++s)
#pragma unroll
            for (int c = 0; c < 3; ++c) f[s][c] += fc[s][c];
    }
    // rate update fused with the analytical damper
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        const bool elem_valid = slot_local<F>(P, lane * EPL + s) < n;
#pragma unroll
        for (int c = 0; c < 3; ++c) fma_inplace_u(L.v[s][c], P.damp_t, fma(C.cf[s], f[s][c], C.ca[s][c]));
        const double ce01 = C.cw01[s] * e[s], ce2 = C.cw2[s] * e[s];
        double w0 = fma(ce01, tq[s][0], L.w[s][0]), w1 = fma(ce01, tq[s][1], L.w[s][1]),
               w2 = fma(ce2, tq[s][2], L.w[s][2]);
        if (has<F>(P, SOFTROD_FEAT_ANALYTICAL_DAMPER)) {
            double ex0, ex2;
            // c_r^e = c_r * c_r^(e-1): the strain e-1 is small, so the exponent stays in the
            // polynomial's range also where log c_r is not (octopus arms: log c_r = -0.057)
            const double em1 = e[s] - 1.0;
            double x0 = em1 * (TAPER ? C.dlog0[s] : P.damp_logr[0]), x2 = em1 * (TAPER ? C.dlog2[s] : P.damp_logr[2]);
            if constexpr (TAPER) {
                // The thin end of a tapered arm has log c_r = -nu dt m / J of -2000 and less (c_r underflows to 0: the
                // damper annihilates omega there, pow(0, e) = 0 in the reference's arithmetic); a COMPRESSED element
                // (e < 1) then asks for exp(+1000) = inf, and inf * 0 is NaN.  Capped where exp is still finite:
                // the product with c_r = 0 stays the 0 it is.
                x0 = fmin(x0, 700.0); x2 = fmin(x2, 700.0);
            }
            exp_pair(x0, x2, elem_valid, ex0, ex2);
            ex0 *= TAPER ? C.dr0[s] : P.damp_r[0]; ex2 *= TAPER ? C.dr2[s] : P.damp_r[2];
            w0 *= ex0; w1 *= ex0; w2 *= ex2;
        }
        L.w[s][0] = w0; L.w[s][1] = w1; L.w[s][2] = w2;
    }
    if (P.damp_before_constrain) {
        if (has<F>(P, SOFTROD_FEAT_LAPLACE_FILTER)) laplace_filter_rates_fast<EPL>(P, lane, L);
        constrain_rates_n<F, EPL>(P, B, lane, L);
        sucker_rates_n<F, EPL>(P, B, lane, L);
    } else {
        BcTargets Bs = B;
        Bs.vel[0] *= P.damp_t; Bs.vel[1] *= P.damp_t; Bs.vel[2] *= P.damp_t;
        constrain_rates_n<F, EPL>(P, Bs, lane, L);
        sucker_rates_n<F, EPL>(P, B, lane, L);
        if (has<F>(P, SOFTROD_FEAT_LAPLACE_FILTER)) laplace_filter_rates_fast<EPL>(P, lane, L);
    }
}

// Wave-uniform: does the rod's dynamic state hold a NaN in a valid node / element?
template <int EPL>
__device__ __forceinline__ bool rod_has_nan(const RodParams& P, int lane, const LaneN<EPL>& L) {
    bool bad = false;
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
        const int idx = slot_local(P, lane * EPL + s);
        bool b = false, q = false;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            b = b || isnan(L.x[s][c]) || isnan(L.v[s][c]);
            q = q || isnan(L.w[s][c]);
        }
#pragma unroll
        for (int c = 0; c < 9; ++c) q = q || isnan(L.Q[s][c]);
        bad = bad || (idx <= P.n_elem && b) || (idx < P.n_elem && q);
    }
    return __any(bad);
}
template <int EPL>
__device__ __forceinline__ void poison_rod(LaneN<EPL>& L) {
    const double nan = __builtin_nan("");
#pragma unroll
    for (int s = 0; s < EPL; ++s) {
#pragma unroll
        for (int c = 0; c < 3; ++c) { L.x[s][c] = nan; L.v[s][c] = nan; L.w[s][c] = nan; L.t[s][c] = nan; }
#pragma unroll
        for (int c = 0; c < 9; ++c) L.Q[s][c] = nan;
    }
}

// Wave priority by progress.  The SIMD's instruction arbiter serves the OLDEST of its waves first
// (at equal s_setprio level): of the rods sharing a SIMD the first runs at the pace of a lone wave
// (71 % of the issue slots for the planar loop: dependent fp64 operations), the others fill the gaps,
// and the last one ends up running alone — measured with tools/phase_clocks.py: 4096 rods x 400
// substeps finish at 105 / 180 / 250 / 317 us on every SIMD, the last 70 us at lone-wave pace.
// Lowering a wave's priority as it advances (four levels: the quarters below, in 64ths of the
// loop) makes the arbiter favour whoever is behind, so that the waves of a SIMD leapfrog and end
// together: the same launch takes 303 us (-5 %).  Two scalar instructions per substep.
struct ProgressPriority {
    int next, q, n;
    __device__ __forceinline__ static int bound(int n, int q) {
        return (int)(((long long)n * (q == 0 ? 32 : q == 1 ? 48 : q == 2 ? 60 : 64)) / 64);
    }
    __device__ __forceinline__ explicit ProgressPriority(int n_) : next(bound(n_, 0)), q(0), n(n_) {}
    __device__ __forceinline__ void tick(int s) {
        if (s + 1 == next) {
            ++q;
            next = q < 4 ? bound(n, q) : -1;
            if (q == 1) __builtin_amdgcn_s_setprio(2);
            else if (q == 2) __builtin_amdgcn_s_setprio(1);
            else __builtin_amdgcn_s_setprio(0);
        }
    }
};

// The clock after n_sub substeps, as the reference accumulates it (2 n_sub additions of dt/2, or
// n_sub of dt: `self.time = self.do_step(self.simulator, self.time, self.time_step)`).  The result
// does not depend on the rod, so the host has accumulated the same float64 additions from a reset
// into a table (StatePtrs.time_tab, one entry per env.step); a clock that IS an entry — always,
// unless somebody wrote the time row — advances to the next entry, any other takes the additions
// here.  Bit-identical either way; the substep loops carry no clock (4 VALU instructions per
// substep with their selects in the general loop).
__device__ __forceinline__ double clock_after(const RodParams& P, const StatePtrs& S, double time, int n_sub) {
    if (n_sub <= 0) return time;
    if (S.time_tab && n_sub == P.tab_n_sub) {
        const double kf = rint(time * P.inv_step_time);
        const int k = (kf >= 0.0 && kf < (double)(P.tab_len - 1)) ? (int)kf : -1;
        if (__builtin_amdgcn_readfirstlane(k) >= 0 && S.time_tab[k] == time) return S.time_tab[k + 1];
    }
    const double ta = P.time_two_half_adds ? P.half_dt : P.dt;
    const double tb = P.time_two_half_adds ? P.half_dt : 0.0;
    for (int s = 0; s < n_sub; ++s) time = (time + ta) + tb;
    return time;
}

// The substeps of one launch on the general 3-D state.  `always_inline` for every kernel whose
// only path it is; the SoftPendulum kernel keeps it OUT of line (cold fallback for non-planar
// states) so that its register demand cannot leak spills into the planar hot loop.
template <unsigned F, int EPL, bool TAPER = false>
__device__ __forceinline__ void general_substeps(const RodParams& P, const RodParams& Pk, const ConstN<EPL>& C,
                                                 const BcTargets& B, int lane, LaneN<EPL>& L, int n_sub) {
#if SOFTROD_IDLE_LANES_EXEC_MASK
    // The lanes past the rod's end sit the loop out like in the planar loop (EXEC cleared: a DPP
    // shift that would read one returns 0, the element behind the last node has no stiffness —
    // bit-identical).  Worth what the box's power budget makes of it: OctoArmSingle 2.574 -> 2.514 ms
    // and SoftPendulum3D 1.700 -> 1.687 ms on a box that holds these loops under its clock ceiling,
    // nothing (2.500 / 2.502, 1.685 / 1.684) on one that does not (profiles/README.md r3h / r3i).
    // Not with the spline muscles, whose rebuild runs prefix sums over the whole wave.
    // INVARIANTS this branch rests on (it is lane-divergent control flow around code that holds
    // barriers — laplace_filter_rates_lds7's __syncthreads — and cross-lane reads):
    //   (1) every wave keeps at least one active lane: lane 0 carries node 0 of a rod (n_elem >= 2);
    //   (2) every barrier inside the loop is reached the same number of times by every wave of the
    //       workgroup: the trip counts (n_sub, filter_order) are kernel arguments, wave-uniform;
    //   (3) a DPP / ds_bpermute read of an EXEC-disabled lane returns 0 (bound_ctrl:1; bpermute of an
    //       inactive source) — the value the idle lanes used to hold.
    // `make -C gym_softrobot_amd/csrc nomask` builds the library with every such mask off, and
    // tests/test_gpu_mask_ab.py holds the two builds bit-identical on all four workloads, so a
    // compiler or ISA change that breaks (1)-(3) fails a test instead of corrupting a rollout.
    if (has<F>(P, SOFTROD_FEAT_SPLINE_MUSCLE_TORQUES) || slot_local<F>(P, lane * EPL) <= P.n_elem)
#endif
    {
        kinematic_n<EPL>(P.half_dt, C, L);
        ProgressPriority prio(n_sub);
        for (int s = 0; s < n_sub; ++s) {
            dynamic_n<F, EPL, TAPER>(Pk, C, B, lane, L);
            const bool last = (s == n_sub - 1);
            kinematic_n<EPL>(last ? P.half_dt : P.dt, C, L);
            prio.tick(s);
        }
    }
}
// The SoftPendulum kernel's fallback for a state that is NOT planar (written through the state
// view, or reset with an out-of-plane frame): the general 3-D substeps as an out-of-line call that
// shares NOTHING with its caller but scalars — it reads the parameters and the array pointers
// from their device-memory copies (StatePtrs.params / .self), loads the rod from its HBM rows,
// establishes the constraint invariants, steps, and stores the rod and its clock back; the
// caller reloads.  Nothing of the caller's lane state is address-taken that way, so the planar
// path keeps every value in registers and the kernel writes no scratch (r1k: a by-reference
// call parked 692 B per lane in scratch on EVERY launch, 164 MB of the 240 MB HBM traffic).
template <unsigned F, int E, int EPL>
__device__ __attribute__((noinline)) void general_step_cold(const RodParams* __restrict__ params,
                                                            const StatePtrs* __restrict__ sp, int rod, int lane,
                                                            const float* __restrict__ actions, int n_sub) {
    const RodParams P = *params;
    const StatePtrs S = *sp;
    const size_t N = (size_t)P.n_envs;
    LaneN<EPL> L;
    load_lane<EPL, F>(S, N, rod, lane, L);
    BcTargets B;
    load_bc(S, N, rod, B);
    EnvAction A;
    set_action_n<F, E, EPL>(P, S, N, rod, lane, actions, A, B, L);
    constrain_rates_n<F, EPL>(P, B, lane, L);
    constrain_values_n<F, EPL>(P, B, lane, L);
    double time = S.time[rod];
    ConstN<EPL> C;
    build_const<F, EPL>(P, lane, A, C);
    RodParams Pk = P;
    if (!has<F>(P, SOFTROD_FEAT_ANALYTICAL_DAMPER)) Pk.damp_t = 1.0;
    general_substeps<F, EPL>(P, Pk, C, B, lane, L, n_sub);
    time = clock_after(P, S, time, n_sub);
    store_lane<EPL, F>(S, N, rod, lane, L);
    if (lane == 0) S.time[rod] = time;
}

// SOFTROD_FAST_WAVES: minimum waves per SIMD the register allocator must leave room for
// (2nd __launch_bounds__ argument = waves per EU on gfx950); tuned in profiles/README.md.
#ifndef SOFTROD_FAST_WAVES
#define SOFTROD_FAST_WAVES 3
#endif
#ifndef SOFTROD_PLANAR_WAVES      // the SoftPendulum instantiation: all four rods of a SIMD resident (4096 envs)
#define SOFTROD_PLANAR_WAVES 4
#endif
#ifndef SOFTROD_PLANAR_EXEC_MASK
#define SOFTROD_PLANAR_EXEC_MASK 1
#endif
#ifndef SOFTROD_CONTACT_WAVES
#define SOFTROD_CONTACT_WAVES 2
#endif
#ifndef SOFTROD_RUNTIME_WAVES      // the run-time-mask instantiations (custom feature mixes, tapered rods)
#define SOFTROD_RUNTIME_WAVES 2
#endif
// Two slots per lane need the whole 512-entry register file (1 wave per SIMD); the contact
// and Laplace-filter instantiations trade a wave of occupancy for not spilling in the loop.
// TAPER: the rod's radius varies along its length (softrod_set_radius_profile): material constants
// come from the per-lane table S.mat instead of the kernel arguments.
// Diagnostic build only (-DSOFTROD_PHASE_CLOCKS, tools/phase_clocks.py): where inside a launch the
// time goes.  Lane 0 of every wave stamps the 100 MHz wall clock at the phase boundaries.
#ifdef SOFTROD_PHASE_CLOCKS
__device__ unsigned long long g_phase_clock[16384][8];
#define SR_PHASE(i) do { __builtin_amdgcn_s_waitcnt(0); if (threadIdx.x == 0 && blockIdx.x < 16384) \
        g_phase_clock[blockIdx.x][i] = wall_clock64(); } while (0)
#else
#define SR_PHASE(i) do {} while (0)
#endif

// Waves per SIMD the register allocator must leave room for, by instantiation.  The run-time-mask
// instantiation carries EVERY feature's code (contact, Laplace filter, spline muscles, suckers) and,
// with TAPER, per-lane material constants in registers instead of kernel arguments: at 3 waves
// (168 VGPRs) it parked 237 VGPRs in scratch (1168 B per lane), at 2 waves (256 VGPRs) it fits like
// the contact instantiations do.
template <unsigned F, int EPL, bool TAPER>
constexpr int fast_kernel_waves() {
    if (EPL > 1) return 1;
    if (F == kRuntimeFeatures) return SOFTROD_RUNTIME_WAVES;
    if (F & (SOFTROD_FEAT_PLANE_CONTACT_ANISO | SOFTROD_FEAT_LAPLACE_FILTER | SOFTROD_FEAT_SPLINE_MUSCLE_TORQUES |
             SOFTROD_FEAT_COOMM_MUSCLES))
        return SOFTROD_CONTACT_WAVES;
    if (TAPER) return SOFTROD_CONTACT_WAVES;
    return F == SOFTROD_FEATURES_SOFTPENDULUM ? SOFTROD_PLANAR_WAVES : SOFTROD_FAST_WAVES;
}

template <unsigned F, int E, int EPL, bool TAPER = false>
__global__ void __launch_bounds__(kLanes, (fast_kernel_waves<F, EPL, TAPER>()))
softrod_step_fast_kernel(const RodParams P, const StatePtrs S, const float* __restrict__ actions,
                         float* __restrict__ obs, double* __restrict__ reward,
                         uint8_t* __restrict__ terminated, uint8_t* __restrict__ truncated,
                         double* __restrict__ aux, const int n_sub, const int epilogue, const int pack) {
    const int rod = blockIdx.x;
    const int lane = threadIdx.x;
    const size_t N = (size_t)P.n_envs;
    if (epilogue && S.skip && S.skip[rod]) {   // reset by the auto-reset pass of this env.step
        if (lane == 0) S.skip[rod] = 0;
        return;
    }

    __builtin_amdgcn_s_setprio(3);     // ProgressPriority lowers it as the substeps advance
    SR_PHASE(0);
    LaneN<EPL> L;
    load_lane<EPL, F>(S, N, rod, lane, L);
    SR_PHASE(1);
    if (has<F>(P, SOFTROD_FEAT_LAPLACE_FILTER)) sanitize_unused_rates<EPL>(P, lane, L);
    if (has<F>(P, SOFTROD_FEAT_PLANE_CONTACT_ANISO)) sanitize_unused_slots<EPL>(P, lane, L);
    BcTargets B;
    load_bc(S, N, rod, B);
    load_suckers<F>(P, S, N, rod, B);
    EnvAction A;
    set_action_n<F, E, EPL>(P, S, N, rod, lane, actions, A, B, L);
    if (epilogue) push_store_prev_com<F, E, EPL>(P, S, N, rod, lane, L, n_sub);
    if (has<F>(P, SOFTROD_FEAT_SPLINE_MUSCLE_TORQUES)) muscle_load<EPL>(P, S, rod, actions != nullptr, A, L);
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
    build_const<F, EPL, TAPER>(P, lane, A, C, S.mat);
    if constexpr (kMusclesCompiled<F>) build_muscle_const<F, EPL, true>(P, S, N, rod, lane, A, C);
    RodParams Pk = P;
    if (!has<F>(P, SOFTROD_FEAT_ANALYTICAL_DAMPER)) Pk.damp_t = 1.0;

    bool stepped = false, stored = false;
    // A rod whose state already holds a NaN (an env that blew up and was not reset — the
    // reference reports terminated and -50 for it on every further step, soft_pendulum.py:
    // 196-208) would only smear that NaN over all of its nodes while dragging every
    // range-reduction loop to its iteration cap.  It is not integrated: its state becomes NaN
    // throughout (where the reference's would end up after a few substeps), its clock advances.
    if (n_sub > 0 && epilogue && rod_has_nan<EPL>(P, lane, L)) {
        poison_rod<EPL>(L);
        time = clock_after(P, S, time, n_sub);
        stepped = true;
    }
    if constexpr (F == SOFTROD_FEATURES_SOFTPENDULUM) {
        // SoftPendulum-v0 lives in the x-y plane: same substep without the identically
        // zero out-of-plane terms (softrod_planar.hpp); any other state takes the 3-D loop
        PlanarN<EPL> Z;
        if (n_sub > 0 && !stepped && planar_from_lane<EPL>(P, B, lane, L, Z)) {
            PlanarC<EPL> K;
            planar_build_const<EPL>(Pk, C, lane, K);
            // The clock takes the reference's additions in the reference's order: 2 n_sub times
            // +dt/2, or n_sub times +dt.  They are 2 of the loop's ~100 VALU instructions, and the
            // result does not depend on the rod: the host has accumulated the same sequence of
            // float64 additions from a reset into a table (StatePtrs.time_tab).  If this rod's clock
            // IS the table's k-th entry — it is, unless somebody wrote the time row — the new clock
            // is entry k + 1 and the loop carries no clock; otherwise the additions run after the
            // loop (bit-identical either way).
            const double ta = P.time_two_half_adds ? P.half_dt : P.dt;
            const double tb = P.time_two_half_adds ? P.half_dt : 0.0;
            int tk = -1;
            if (S.time_tab && n_sub == P.tab_n_sub) {
                const double kf = rint(time * P.inv_step_time);
                const int k = (kf >= 0.0 && kf < (double)(P.tab_len - 1)) ? (int)kf : -1;
                tk = (k >= 0 && S.time_tab[k] == time) ? k : -1;
            }
            // two half kinematic steps between force evaluations are one whole step; the last
            // substep (half a step) is peeled so that the loop's step length is a constant
            // (the step length as a vector register where scalar ones are short: uniform_k)
            const double step_dt = EPL > 1 ? opaque_v(P.dt) : P.dt;
            SR_PHASE(2);
#if SOFTROD_PLANAR_EXEC_MASK
            // The lanes past the rod's end node (13 of 64 at 50 elements) take no part in the loop:
            // with EXEC cleared they issue nothing to the fp64 datapath, which under a kernel that
            // keeps the VALU busy in every cycle is power — and the clock this kernel sustains is
            // what the power controller leaves it (profiles/README.md r3h: 0.3008 -> 0.2846 ms, bit-
            // identical).  A DPP shift that would read a disabled lane returns 0 (bound_ctrl), exactly
            // what it returns past lane 63, and the element after the last node has no stiffness either
            // way: results are unchanged.  (The 3-D loops carry the same mask, general_substeps, and so
            // do OctoFlat's ghost slots, softrod_octo.hpp.)
            if (lane * EPL <= P.n_elem + (SOFTROD_PLANAR_EXEC_MASK == 2 ? 64 : 0)) {     // (2: the same code with no lane masked, an A/B control)
#endif
                planar_kinematic_n<EPL>(P.half_dt, K.hq_hdt, C, K, Z);
                {   // four copies of the loop, one per priority level (ProgressPriority explains; here
                    // without its two scalar instructions and the branch in every trip)
                    int s = 0;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (q == 1) __builtin_amdgcn_s_setprio(2);
                        if (q == 2) __builtin_amdgcn_s_setprio(1);
                        if (q == 3) __builtin_amdgcn_s_setprio(0);
                        const int end = ProgressPriority::bound(n_sub - 1, q);
                        for (; s < end; ++s) {
                            planar_dynamic_n<EPL>(Pk, C, K, lane, Z);
#ifdef SOFTROD_DIAG_TWO_HALF_STEPS
                            planar_kinematic_n<EPL>(P.half_dt, K.hq_hdt, C, K, Z);
                            planar_kinematic_n<EPL>(P.half_dt, K.hq_hdt, C, K, Z);
#else
                            planar_kinematic_n<EPL>(step_dt, K.hq_dt, C, K, Z);
#endif
                        }
                    }
                }
                planar_dynamic_n<EPL>(Pk, C, K, lane, Z);
                planar_kinematic_n<EPL>(P.half_dt, K.hq_hdt, C, K, Z);
#if SOFTROD_PLANAR_EXEC_MASK
            }
#endif
            if (__builtin_amdgcn_readfirstlane(tk) >= 0) time = S.time_tab[tk + 1];
            else
                for (int s = 0; s < n_sub; ++s) time = (time + ta) + tb;
            SR_PHASE(3);
            // Only the rows the step changed go back (12 of 21).  The epilogue reads positions,
            // velocities and tangents; everything else of the 3-D lane state is dead from here on
            // the planar path — said explicitly, so that none of it stays in registers (at 128:
            // in scratch) across the loop for the sake of the other path's merge.
            planar_store<EPL>(S, N, rod, lane, Z);
#pragma unroll
            for (int s = 0; s < EPL; ++s) {
#pragma unroll
                for (int c = 0; c < 3; ++c) { L.x[s][c] = 0.0; L.v[s][c] = 0.0; L.w[s][c] = 0.0; L.t[s][c] = 0.0; }
#pragma unroll
                for (int c = 0; c < 9; ++c) L.Q[s][c] = 0.0;
            }
            planar_to_lane<EPL>(Z, L);
            stepped = true;
            stored = true;
        }
    }
    if (n_sub > 0 && !stepped) {
        if constexpr (F == SOFTROD_FEATURES_SOFTPENDULUM) {
            general_step_cold<F, E, EPL>(S.params, S.self, rod, lane, actions, n_sub);   // HBM -> HBM
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            load_lane<EPL, F>(S, N, rod, lane, L);
            time = S.time[rod];
            stored = true;
        } else
        {
            general_substeps<F, EPL, TAPER>(P, Pk, C, B, lane, L, n_sub);
            time = clock_after(P, S, time, n_sub);
        }
    }
    if (!stored) store_lane<EPL, F>(S, N, rod, lane, L);
    if (has<F>(P, SOFTROD_FEAT_SPLINE_MUSCLE_TORQUES)) muscle_store<EPL>(P, S, rod, lane, L);
    if (env_of<E>(P) == SOFTROD_ENV_SOFT_ARM && lane == 0)   // self.tick += 1 per substep, soft_arm_tracking.py:222
        S.ctrl[(size_t)0 * N + rod] += (double)n_sub;
    if (lane == 0) S.time[rod] = time;
    SR_PHASE(4);
    if (epilogue)
        env_epilogue_n<E, EPL>(P, S, N, rod, lane, C, L, time, A, obs, reward, terminated, truncated, aux, pack);
    SR_PHASE(5);
}


}  // namespace softrod
