/*
 * softrod_oracle.c — CPU restatement (plain C, fp64) of the SoftPendulum-v0 hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load this; the product path
 * (gym_softrobot_amd/, libsoftrod_hip.so) never links, imports or calls it.
 *
 * PARITY UNPINNED.  The arithmetic of this path does not live in
 * /root/reference: gym-softrobot delegates it to the third-party package
 * pyelastica==1.0.0 (uv.lock:845-846, pyproject.toml:22), which is neither
 * vendored, installed nor installable here, and the reference's own tests hold
 * no golden numbers for it (tests/envs/test_determinism.py:46-54 compares two
 * runs with each other only).  What follows restates PyElastica's published
 * algorithm (Gazzola, Dudte, McCormick, Mahadevan, R. Soc. Open Sci. 5:171628,
 * 2018) in the operation order of the PyElastica modules named per function
 * (module names recalled, not on disk), anchored on the reference's call sites:
 *   - assembly:     gym_softrobot/envs/soft_pendulum/build.py:29-115
 *   - hot loop:     gym_softrobot/envs/soft_pendulum/soft_pendulum.py:183-184
 *   - epilogue:     soft_pendulum.py:149-161,196-251
 * Every constant SURVEY.md App. A marks "(?)" is a field of softrod_config so it
 * can be flipped once pyelastica is importable.  The only reference-derived
 * golden vectors that exist (reset observations, App. B) are checked in
 * tests/test_oracle_golden.py; the physics is pinned by known-answer tests
 * (tests/test_oracle_physics.py).
 *
 * Round-2 corrections of the recollection (each moves results far below 1e-5; regression pins
 * regenerated): (1) _get_rotation_matrix normalises the axis with the UNSCALED |omega| + 1e-14
 * and scales the angle afterwards; (2) anisotropic_friction distributes the kinetic friction with
 * the unit vector of the total SLIP velocity (rolling slip incl. the contact point's spin + axial
 * velocity), the 1e-14 added component-wise before the norm — not with the element velocity.
 *
 * Build: see oracle/Makefile (gcc -O2 -ffp-contract=off: Numba does not
 * contract a*b+c either).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#include "../include/softrod.h"

#ifdef ORACLE_LIBM_JITTER
/* CONTROL BUILD (oracle/Makefile: jitter): the results of libm's transcendental functions — sin, cos, acos, pow,
 * exp, the calls PyElastica's substep makes through NumPy / Numba — are moved by one unit in the last place in half
 * of the calls (deterministically, by a hash of the result).  A second, equally valid evaluation of the same algorithm
 * under ANOTHER libm: glibc, Numba's LLVM intrinsics and the GPU's device library round these functions differently
 * in the last bit, which the FMA control (same glibc, contraction only) does not show.  tools/ensemble_parity.py
 * calibrates its paired-divergence band on the larger of the two controls. */
static inline double oracle_ulp_jitter(double x)
{
    uint64_t u;
    if (!(x == x) || x == 0.0) return x;
    memcpy(&u, &x, 8);
    const uint64_t h = (u ^ (u >> 29)) * 0x9E3779B97F4A7C15ull;
    if ((h >> 62) == 0) u += 1;
    else if ((h >> 62) == 1) u -= 1;
    memcpy(&x, &u, 8);
    return x;
}
#define sin(x) oracle_ulp_jitter((sin)(x))
#define cos(x) oracle_ulp_jitter((cos)(x))
#define acos(x) oracle_ulp_jitter((acos)(x))
#define pow(x, y) oracle_ulp_jitter((pow)(x, y))
#define exp(x) oracle_ulp_jitter((exp)(x))
#endif

#define NMAX 256 /* max elements per rod the oracle supports */

typedef struct oracle_rod {
    softrod_config cfg;
    int n; /* elements */
    /* state (PyElastica: position_collection (3,n+1), director_collection
     * (3,3,n) rows = d1,d2,d3 in lab frame, velocity, omega (material frame)) */
    double x[3][NMAX + 1], v[3][NMAX + 1];
    double Q[3][3][NMAX], w[3][NMAX];
    double time;
    /* constants from CosseratRod.straight_rod (elastica/rod/factory_function.py) */
    double mass[NMAX + 1];
    double rest_len[NMAX], rest_vor[NMAX], volume[NMAX];
    double J[3][NMAX], invJ[3][NMAX];      /* diagonal mass second moment   */
    double shear[3][NMAX];                 /* diag(ac*G*A, ac*G*A, E*A)     */
    double bend[3][NMAX];                  /* Voronoi-averaged diag(EI,EI,GI3) */
    double rest_sigma[3][NMAX], rest_kappa[3][NMAX];
    double damp_t;                         /* exp(-nu dt)                   */
    double damp_r[3][NMAX];                /* exp(-nu dt m_e invJ)          */
    /* caches (as of the last force evaluation) */
    double len[NMAX], tang[3][NMAX], dil[NMAX], vdil[NMAX], dil_rate[NMAX];
    double sigma[3][NMAX], kappa[3][NMAX];
    double n_int[3][NMAX], m_int[3][NMAX]; /* internal_stress / internal_couple */
    double f_int[3][NMAX + 1], t_int[3][NMAX];
    double f_ext[3][NMAX + 1], t_ext[3][NMAX];
    /* boundary condition targets */
    double fixed_pos[3], fixed_dir[3][3];
    float prev_action; /* _prev_action, soft_pendulum.py:97-99,165 */
    double point_force; /* point_force[0], soft_pendulum.py:117,166 */
    double radius[NMAX];   /* rod.radius, refreshed with the geometry (volume preserving) */
    /* ArmSingleEnv memory: prev_kappa_state, prev_com_state, _prev_action
     * (octopus/arm_single_env.py:97-99,172-173,190-198) */
    double prev_kappa[NMAX], prev_com[2];
    float prev_action7[7];
    /* MovingBaseController, soft_pendulum_3d/build.py:15-20 */
    double ctrl_pos[3], ctrl_vel[3];
    float prev_action2[2]; /* SoftPendulum3DEnv._prev_action */
    /* SoftArmTracking: the two MuscleTorquesWithVaryingBetaSplines (normal, binormal)
     * muscle_torques_with_bspline.py:98-126 — points_array, points_cached[1,1:-1],
     * initial_call_flag, torque_magnitude_cache; the interpolant's piecewise-cubic form */
    double pts_input[2][SOFTROD_MAX_CTRL], pts_cached[2][SOFTROD_MAX_CTRL];
    int pts_init[2];
    double torque_mag[2][NMAX];
    double spline_breaks[SOFTROD_MAX_SPLINE_PIECES + 1];
    double spline_coef[SOFTROD_MAX_SPLINE_PIECES][SOFTROD_MAX_CTRL][4];
    /* CosseratRod.straight_rod(base_radius=<array>): per-element rest radii of a tapered rod
     * (octopus/arm_push_env.py:160-179); has_profile = 0: the uniform cfg.base_radius */
    double radius_profile[NMAX];
    int has_profile;
    /* ControllableFixConstraint: effective reduction ratio of each sucker (controller.flag ?
     * controller.reduction_ratio : 0), octopus/controllable_constraint.py:45-69 */
    double sucker_ratio[SOFTROD_MAX_SUCKERS];
    int sucker_index[SOFTROD_MAX_SUCKERS];   /* SuckerController.index (Python indexing; set_action rewrites it) */
    /* COOMM muscle layers (ApplyMuscles; octopus/build.py:295-338): geometry / strength tables, the
     * activations apply_activation wrote, and the per-substep work arrays of coomm's Muscle objects */
    double m_ratio[SOFTROD_MAX_MUSCLES][3][NMAX], m_strength[SOFTROD_MAX_MUSCLES][NMAX];
    double m_act[SOFTROD_MAX_MUSCLES][NMAX];
    double m_force[SOFTROD_MAX_MUSCLES][NMAX], m_length[SOFTROD_MAX_MUSCLES][NMAX];   /* diagnostics */
    float prev_action_push[2];   /* ArmPushEnv._prev_action (discrete: the index in [0]) */
    int round_state_f32;   /* diagnostic (tools/episode_parity.py --fp32-proxy): the dynamic state is
                              rounded to float32 after every substep — float32 STORAGE with float64
                              arithmetic, a lower bound on what a float32 stepper loses */
    int run_substeps;      /* >= 0: substeps the env_step functions really run (fixture replay: the
                              epilogue alone on an injected state); < 0: cfg.n_substeps */
    long tick;             /* soft_arm_tracking.py:222 */
    double arm_target[3];  /* wsol[tick] */
} oracle_rod;

/* ------------------------------------------------------------------------- */
/* CosseratRod.straight_rod -> elastica/rod/factory_function.py allocate()    */
/* call site: build.py:54-61                                                   */
/* ------------------------------------------------------------------------- */
/* AnalyticalLinearDamper(time_step=...): the stepper's dt unless the build hands the damper another one
 * (build_muscle_octopus.py:101-106: 7e-5 under a stepper run at 5e-5, crawl_env.py:63) */
static double damper_dt(const softrod_config* c) { return c->damper_time_step > 0.0 ? c->damper_time_step : c->dt; }

static void straight_rod(oracle_rod* r, const double start[3],
                         const double direction[3], const double normal_in[3])
{
    const softrod_config* c = &r->cfg;
    const int n = r->n;
    double end[3], normal[3];
    for (int i = 0; i < 3; ++i) end[i] = start[i] + direction[i] * c->base_length;
    /* np.linspace(start, end, n+1): start + k*step with step=(end-start)/n,
     * last point forced to `end`. */
    for (int i = 0; i < 3; ++i) {
        const double step = (end[i] - start[i]) / (double)n;
        for (int k = 0; k <= n; ++k) r->x[i][k] = start[i] + (double)k * step;
        r->x[i][n] = end[i];
    }
    double nn = sqrt(normal_in[0] * normal_in[0] + normal_in[1] * normal_in[1] +
                     normal_in[2] * normal_in[2]);
    for (int i = 0; i < 3; ++i) normal[i] = normal_in[i] / nn;

    for (int k = 0; k < n; ++k) {
        double d[3];
        for (int i = 0; i < 3; ++i) d[i] = r->x[i][k + 1] - r->x[i][k];
        const double l = sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        r->rest_len[k] = l;
        double t[3] = { d[0] / l, d[1] / l, d[2] / l };
        /* directors rows: d1 = normal, d2 = tangent x normal, d3 = tangent */
        for (int i = 0; i < 3; ++i) r->Q[0][i][k] = normal[i];
        r->Q[1][0][k] = t[1] * normal[2] - t[2] * normal[1];
        r->Q[1][1][k] = t[2] * normal[0] - t[0] * normal[2];
        r->Q[1][2][k] = t[0] * normal[1] - t[1] * normal[0];
        for (int i = 0; i < 3; ++i) r->Q[2][i][k] = t[i];
    }
    /* radius[:] = base_radius broadcasts a scalar or takes an array of n_elements radii */
    double be[3][NMAX];
    for (int k = 0; k < n; ++k) {
        const double radius = r->has_profile ? r->radius_profile[k] : c->base_radius;
        const double A0 = M_PI * radius * radius;
        const double I1 = A0 * A0 / (4.0 * M_PI);
        const double I[3] = { I1, I1, 2.0 * I1 };
        for (int i = 0; i < 3; ++i) {
            r->J[i][k] = I[i] * (c->density * r->rest_len[k]);
            r->invJ[i][k] = 1.0 / r->J[i][k];
        }
        r->shear[0][k] = c->alpha_c * c->shear_modulus * A0;
        r->shear[1][k] = c->alpha_c * c->shear_modulus * A0;
        r->shear[2][k] = c->youngs_modulus * A0;
        r->volume[k] = M_PI * (radius * radius) * r->rest_len[k];
        /* element bend matrix, then rest-length-weighted average onto Voronoi */
        be[0][k] = c->youngs_modulus * I[0];
        be[1][k] = c->youngs_modulus * I[1];
        be[2][k] = c->shear_modulus * I[2];
    }
    for (int k = 0; k < n - 1; ++k) {
        for (int i = 0; i < 3; ++i)
            r->bend[i][k] = (be[i][k + 1] * r->rest_len[k + 1] + be[i][k] * r->rest_len[k]) /
                            (r->rest_len[k + 1] + r->rest_len[k]);
        r->rest_vor[k] = 0.5 * (r->rest_len[k + 1] + r->rest_len[k]);
    }
    for (int k = 0; k <= n; ++k) r->mass[k] = 0.0;
    for (int k = 0; k < n; ++k) {
        r->mass[k] += 0.5 * c->density * r->volume[k];
        r->mass[k + 1] += 0.5 * c->density * r->volume[k];
    }
    for (int i = 0; i < 3; ++i) {
        for (int k = 0; k <= n; ++k) { r->v[i][k] = 0.0; r->f_ext[i][k] = 0.0; r->f_int[i][k] = 0.0; }
        for (int k = 0; k < n; ++k) {
            r->w[i][k] = 0.0; r->t_ext[i][k] = 0.0; r->t_int[i][k] = 0.0;
            r->rest_sigma[i][k] = 0.0; r->rest_kappa[i][k] = 0.0;
        }
    }
    /* AnalyticalLinearDamper.__init__ (elastica/dissipation.py), build.py:108-113.
     * `damping_constant=` keyword -> per-unit-mass protocol. */
    r->damp_t = exp(-c->damping_constant * damper_dt(c));
    for (int k = 0; k < n; ++k) {
        double me = 0.5 * (r->mass[k + 1] + r->mass[k]);
        if (k == 0) me += 0.5 * r->mass[0];
        if (k == n - 1) me += 0.5 * r->mass[n];
        for (int i = 0; i < 3; ++i)     /* damper_protocol 1: `uniform_damping_constant=`, exp(-nu dt) on every rate */
            r->damp_r[i][k] = c->damper_protocol == 1 ? r->damp_t : exp(-c->damping_constant * damper_dt(c) * me * r->invJ[i][k]);
    }
    /* constraint targets: ConstraintBase is handed position[..., idx] and
     * directors[..., idx] at finalize (build.py:81-85) */
    for (int i = 0; i < 3; ++i) {
        r->fixed_pos[i] = r->x[i][0];
        for (int j = 0; j < 3; ++j) r->fixed_dir[i][j] = r->Q[i][j][0];
    }
    r->time = 0.0;
    r->point_force = 0.0;
    for (int i = 0; i < 3; ++i) { r->ctrl_pos[i] = 0.0; r->ctrl_vel[i] = 0.0; }
    /* SuckerController: reduction_ratio as configured, switched on after finalize (arm_push_env.py:188,222) */
    for (int j = 0; j < SOFTROD_MAX_SUCKERS; ++j) {
        r->sucker_ratio[j] = (j < c->n_suckers) ? c->sucker_reduction_ratio : 0.0;
        r->sucker_index[j] = c->sucker_index[j];
    }
}

/* ------------------------------------------------------------------------- */
/* elastica/rod/cosserat_rod.py kernels                                        */
/* ------------------------------------------------------------------------- */

/* _compute_geometry_from_state + _compute_all_dilatations */
static void compute_all_dilatations(oracle_rod* r)
{
    const int n = r->n;
    for (int k = 0; k < n; ++k) {
        double d0 = r->x[0][k + 1] - r->x[0][k];
        double d1 = r->x[1][k + 1] - r->x[1][k];
        double d2 = r->x[2][k + 1] - r->x[2][k];
        r->len[k] = sqrt(d0 * d0 + d1 * d1 + d2 * d2) + r->cfg.eps_length;
        r->tang[0][k] = d0 / r->len[k];
        r->tang[1][k] = d1 / r->len[k];
        r->tang[2][k] = d2 / r->len[k];
        r->radius[k] = sqrt(r->volume[k] / r->len[k] / M_PI); /* used by plane contact only */
        r->dil[k] = r->len[k] / r->rest_len[k];
    }
    for (int k = 0; k < n - 1; ++k) {
        const double vl = 0.5 * (r->len[k + 1] + r->len[k]);
        r->vdil[k] = vl / r->rest_vor[k];
    }
}

/* _compute_shear_stretch_strains + _compute_internal_shear_stretch_stresses_from_model */
static void compute_shear_stress(oracle_rod* r)
{
    const int n = r->n;
    compute_all_dilatations(r);
    for (int k = 0; k < n; ++k) {
        for (int i = 0; i < 3; ++i) {
            double qt = r->Q[i][0][k] * r->tang[0][k];
            qt += r->Q[i][1][k] * r->tang[1][k];
            qt += r->Q[i][2][k] * r->tang[2][k];
            r->sigma[i][k] = r->dil[k] * qt - (i == 2 ? 1.0 : 0.0);
        }
        for (int i = 0; i < 3; ++i)
            r->n_int[i][k] = r->shear[i][k] * (r->sigma[i][k] - r->rest_sigma[i][k]);
    }
}

/* _compute_internal_forces: Q^T n / e then the two-point difference kernel */
static void compute_internal_forces(oracle_rod* r)
{
    const int n = r->n;
    double cs[3][NMAX];
    compute_shear_stress(r);
    for (int k = 0; k < n; ++k)
        for (int i = 0; i < 3; ++i) {
            double s = 0.0;
            for (int j = 0; j < 3; ++j) s += r->Q[j][i][k] * r->n_int[j][k];
            cs[i][k] = s / r->dil[k];
        }
    for (int i = 0; i < 3; ++i) {
        r->f_int[i][0] = cs[i][0];
        for (int k = 1; k < n; ++k) r->f_int[i][k] = cs[i][k] - cs[i][k - 1];
        r->f_int[i][n] = -cs[i][n - 1];
    }
}

/* elastica/_rotations.py _inv_rotate + _compute_bending_twist_strains */
static void compute_bending_twist_strains(oracle_rod* r)
{
    const int n = r->n;
    for (int k = 0; k < n - 1; ++k) {
        double R[3][3]; /* Q_{k+1} Q_k^T */
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double s = r->Q[i][0][k + 1] * r->Q[j][0][k];
                s += r->Q[i][1][k + 1] * r->Q[j][1][k];
                s += r->Q[i][2][k + 1] * r->Q[j][2][k];
                R[i][j] = s;
            }
        double vec[3] = { R[2][1] - R[1][2], R[0][2] - R[2][0], R[1][0] - R[0][1] };
        const double trace = (R[0][0] + R[1][1]) + R[2][2];
        const double theta = acos(0.5 * trace - 0.5 - r->cfg.acos_shift);
        const double f = -0.5 * theta / sin(theta + r->cfg.eps_sin);
        for (int i = 0; i < 3; ++i) r->kappa[i][k] = (vec[i] * f) / r->rest_vor[k];
    }
}

/* _compute_dilatation_rate (expanded dot-product form, as PyElastica writes it) */
static void compute_dilatation_rate(oracle_rod* r)
{
    const int n = r->n;
    double rv[NMAX + 1];
    for (int k = 0; k <= n; ++k)
        rv[k] = (r->x[0][k] * r->v[0][k] + r->x[1][k] * r->v[1][k]) + r->x[2][k] * r->v[2][k];
    for (int k = 0; k < n; ++k) {
        const double rp1v = (r->x[0][k + 1] * r->v[0][k] + r->x[1][k + 1] * r->v[1][k]) +
                            r->x[2][k + 1] * r->v[2][k];
        const double rvp1 = (r->x[0][k] * r->v[0][k + 1] + r->x[1][k] * r->v[1][k + 1]) +
                            r->x[2][k] * r->v[2][k + 1];
        r->dil_rate[k] = (rv[k] + rv[k + 1] - rvp1 - rp1v) / r->len[k] / r->rest_len[k];
    }
}

/* _compute_internal_torques */
static void compute_internal_torques(oracle_rod* r)
{
    const int n = r->n;
    double c2d[3][NMAX], c3d[3][NMAX]; /* Voronoi-domain quantities */
    compute_bending_twist_strains(r);
    for (int k = 0; k < n - 1; ++k)
        for (int i = 0; i < 3; ++i)
            r->m_int[i][k] = r->bend[i][k] * (r->kappa[i][k] - r->rest_kappa[i][k]);
    compute_dilatation_rate(r);
    for (int k = 0; k < n - 1; ++k) {
        const double e3 = 1.0 / (r->vdil[k] * r->vdil[k] * r->vdil[k]);
        const double* kp[3] = { &r->kappa[0][k], &r->kappa[1][k], &r->kappa[2][k] };
        const double* mm[3] = { &r->m_int[0][k], &r->m_int[1][k], &r->m_int[2][k] };
        for (int i = 0; i < 3; ++i) c2d[i][k] = r->m_int[i][k] * e3;
        c3d[0][k] = ((*kp[1]) * (*mm[2]) - (*kp[2]) * (*mm[1])) * r->rest_vor[k] * e3;
        c3d[1][k] = ((*kp[2]) * (*mm[0]) - (*kp[0]) * (*mm[2])) * r->rest_vor[k] * e3;
        c3d[2][k] = ((*kp[0]) * (*mm[1]) - (*kp[1]) * (*mm[0])) * r->rest_vor[k] * e3;
    }
    for (int k = 0; k < n; ++k) {
        double d2[3], d3[3];
        for (int i = 0; i < 3; ++i) {
            /* difference kernel / trapezoidal kernel, Voronoi -> element */
            if (k == 0) { d2[i] = c2d[i][0]; d3[i] = 0.5 * c3d[i][0]; }
            else if (k == n - 1) { d2[i] = -c2d[i][n - 2]; d3[i] = 0.5 * c3d[i][n - 2]; }
            else { d2[i] = c2d[i][k] - c2d[i][k - 1]; d3[i] = 0.5 * (c3d[i][k] + c3d[i][k - 1]); }
        }
        double qt[3], ssc[3], jwe[3], lt[3], ud[3];
        for (int i = 0; i < 3; ++i) {
            double s = r->Q[i][0][k] * r->tang[0][k];
            s += r->Q[i][1][k] * r->tang[1][k];
            s += r->Q[i][2][k] * r->tang[2][k];
            qt[i] = s;
        }
        ssc[0] = (qt[1] * r->n_int[2][k] - qt[2] * r->n_int[1][k]) * r->rest_len[k];
        ssc[1] = (qt[2] * r->n_int[0][k] - qt[0] * r->n_int[2][k]) * r->rest_len[k];
        ssc[2] = (qt[0] * r->n_int[1][k] - qt[1] * r->n_int[0][k]) * r->rest_len[k];
        for (int i = 0; i < 3; ++i) jwe[i] = (r->J[i][k] * r->w[i][k]) / r->dil[k];
        lt[0] = jwe[1] * r->w[2][k] - jwe[2] * r->w[1][k];
        lt[1] = jwe[2] * r->w[0][k] - jwe[0] * r->w[2][k];
        lt[2] = jwe[0] * r->w[1][k] - jwe[1] * r->w[0][k];
        for (int i = 0; i < 3; ++i) ud[i] = jwe[i] * r->dil_rate[k] / r->dil[k];
        for (int i = 0; i < 3; ++i)
            r->t_int[i][k] = d2[i] + d3[i] + ssc[i] + lt[i] + ud[i];
    }
}

/* elastica/_rotations.py _get_rotation_matrix(1.0, prefac*omega) and
 * elastica/timestepper/symplectic_steppers.py overload_operator_kinematic_numba */
static void kinematic_step(oracle_rod* r, double prefac)
{
    const int n = r->n;
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k <= n; ++k) r->x[i][k] += prefac * r->v[i][k];
    for (int k = 0; k < n; ++k) {
        /* _get_rotation_matrix(scale = prefac, axis_collection = omega): the axis is normalised
         * with the UNSCALED |omega| + 1e-14, the angle is scaled afterwards (theta *= scale) */
        double v0 = r->w[0][k], v1 = r->w[1][k], v2 = r->w[2][k];
        double theta = sqrt(v0 * v0 + v1 * v1 + v2 * v2);
        v0 /= theta + r->cfg.eps_rot_axis;
        v1 /= theta + r->cfg.eps_rot_axis;
        v2 /= theta + r->cfg.eps_rot_axis;
        theta *= prefac;
        const double up = sin(theta), usq = 1.0 - cos(theta);
        double R[3][3];
        R[0][0] = 1.0 - usq * (v1 * v1 + v2 * v2);
        R[1][1] = 1.0 - usq * (v0 * v0 + v2 * v2);
        R[2][2] = 1.0 - usq * (v0 * v0 + v1 * v1);
        R[0][1] = up * v2 + usq * v0 * v1;
        R[1][0] = -up * v2 + usq * v0 * v1;
        R[0][2] = -up * v1 + usq * v0 * v2;
        R[2][0] = up * v1 + usq * v0 * v2;
        R[1][2] = up * v0 + usq * v1 * v2;
        R[2][1] = -up * v0 + usq * v1 * v2;
        double Qn[3][3];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double s = R[i][0] * r->Q[0][j][k];
                s += R[i][1] * r->Q[1][j][k];
                s += R[i][2] * r->Q[2][j][k];
                Qn[i][j] = s;
            }
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) r->Q[i][j][k] = Qn[i][j];
    }
}

/* constrain_values: PendulumBoundaryConditions (build.py:71-74) or OneEndFixedBC */
static void constrain_values(oracle_rod* r)
{
    if (r->cfg.features & SOFTROD_FEAT_PENDULUM_BC) {
        r->x[1][0] = r->fixed_pos[1];
        r->x[2][0] = r->fixed_pos[2];
        for (int j = 0; j < 3; ++j) {
            r->Q[0][j][0] = r->fixed_dir[0][j];
            r->Q[2][j][0] = r->fixed_dir[2][j]; /* row 1 deliberately untouched */
        }
    }
    if (r->cfg.features & SOFTROD_FEAT_FIXED_BC) {
        for (int i = 0; i < 3; ++i) {
            r->x[i][0] = r->fixed_pos[i];
            for (int j = 0; j < 3; ++j) r->Q[i][j][0] = r->fixed_dir[i][j];
        }
    }
    /* MovingBaseConstraint.constrain_values, soft_pendulum_3d/build.py:31-34 */
    if (r->cfg.features & SOFTROD_FEAT_MOVING_BASE_BC) {
        for (int i = 0; i < 3; ++i) r->x[i][0] = r->ctrl_pos[i];
        r->x[2][0] = r->fixed_pos[2];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) r->Q[i][j][0] = r->fixed_dir[i][j];
    }
}

/* constrain_rates (build.py:76-79) */
static void constrain_rates(oracle_rod* r)
{
    if (r->cfg.features & SOFTROD_FEAT_PENDULUM_BC) {
        r->v[1][0] = 0.0; r->v[2][0] = 0.0;
        r->w[0][0] = 0.0; r->w[2][0] = 0.0;
    }
    if (r->cfg.features & SOFTROD_FEAT_FIXED_BC)
        for (int i = 0; i < 3; ++i) { r->v[i][0] = 0.0; r->w[i][0] = 0.0; }
    /* MovingBaseConstraint.constrain_rates, soft_pendulum_3d/build.py:36-39 */
    if (r->cfg.features & SOFTROD_FEAT_MOVING_BASE_BC) {
        for (int i = 0; i < 3; ++i) r->v[i][0] = r->ctrl_vel[i];
        r->v[2][0] = 0.0;
        for (int i = 0; i < 3; ++i) r->w[i][0] = 0.0;
    }
    /* ControllableFixConstraint.nb_compute_constrain_rates, octopus/controllable_constraint.py:63-69
     * (a controller that is off is an effective ratio of 0: x * (1 - 0) = x) */
    if (r->cfg.features & SOFTROD_FEAT_SUCKER_CONSTRAINT)
        for (int j = 0; j < r->cfg.n_suckers; ++j) {
            /* velocity_collection[..., index] has n + 1 columns, omega_collection[..., index] n: a negative
             * index counts from each array's own end (arm_push_env.py:262 sets index = -1) */
            const int idx = r->sucker_index[j];
            const int node = idx >= 0 ? idx : r->n + 1 + idx, elem = idx >= 0 ? idx : r->n + idx;
            const double keep = 1.0 - r->sucker_ratio[j];
            for (int i = 0; i < 3; ++i) { r->v[i][node] *= keep; r->w[i][elem] *= keep; }
        }
}

/* ------------------------------------------------------------------------- */
/* RodPlaneContactWithAnisotropicFriction.apply_contact (pyelastica 1.0.0      */
/* elastica/contact_forces.py -> _contact_functions.py: the normal-force        */
/* kernel `_calculate_contact_forces_rod_plane` (Gazzola et al. 2018 eq. 4.8)  */
/* and `anisotropic_friction`), registered at octopus/build.py:274-283.         */
/* RECALLED, not on disk: second-largest parity risk after the stepper.        */
/* ------------------------------------------------------------------------- */
static void node_to_element_force(const oracle_rod* r, double out[3][NMAX])
{
    const int n = r->n;
    for (int i = 0; i < 3; ++i) {
        for (int k = 0; k < n; ++k) {
            const double a = r->f_int[i][k] + r->f_ext[i][k];
            const double b = r->f_int[i][k + 1] + r->f_ext[i][k + 1];
            out[i][k] = 0.0;
            out[i][k] += 0.5 * (a + b);
        }
        out[i][0] += 0.5 * (r->f_int[i][0] + r->f_ext[i][0]);
        out[i][n - 1] += 0.5 * (r->f_int[i][n] + r->f_ext[i][n]);
    }
}

static void elements_to_nodes(oracle_rod* r, const double e[3][NMAX])
{
    const int n = r->n;
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < n; ++k) {
            r->f_ext[i][k] += 0.5 * e[i][k];
            r->f_ext[i][k + 1] += 0.5 * e[i][k];
        }
}

static double sign_of(double x) { return (x > 0.0) - (x < 0.0); }

/* find_slipping_elements: 1 below the threshold, linear ramp to 0 at 2x threshold */
static double slip_function(const double vec[3], double thr)
{
    const double a = sqrt(vec[0] * vec[0] + vec[1] * vec[1] + vec[2] * vec[2]);
    if (fabs(a) > thr) {
        double m = a / thr - 1.0;
        if (m > 1.0) m = 1.0;
        return fabs(1.0 - m);
    }
    return 1.0;
}

static void plane_contact(oracle_rod* r)
{
    const softrod_config* c = &r->cfg;
    const int n = r->n;
    const double* nrm = c->plane_normal;
    double fel[3][NMAX], vel[3][NMAX], resp_mag[NMAX], e_tot[3][NMAX];
    int no_contact[NMAX];

    /* ---- normal force ---- */
    node_to_element_force(r, fel);
    for (int k = 0; k < n; ++k) {
        double fn = nrm[0] * fel[0][k] + nrm[1] * fel[1][k] + nrm[2] * fel[2][k];
        double along[3] = { nrm[0] * fn, nrm[1] * fn, nrm[2] * fn };
        if (fn > 0.0) along[0] = along[1] = along[2] = 0.0;
        double resp[3] = { -along[0], -along[1], -along[2] };
        double xe[3], dist = 0.0;
        for (int i = 0; i < 3; ++i) {
            xe[i] = 0.5 * (r->x[i][k] + r->x[i][k + 1]);
            dist += nrm[i] * (xe[i] - c->plane_origin[i]);
        }
        double pen = dist - r->radius[k];
        if (pen > 0.0) pen = 0.0;
        double vn = 0.0;
        for (int i = 0; i < 3; ++i) {
            vel[i][k] = (r->mass[k + 1] * r->v[i][k + 1] + r->mass[k] * r->v[i][k]);
            vel[i][k] /= (r->mass[k + 1] + r->mass[k]);
            vn += nrm[i] * vel[i][k];
        }
        no_contact[k] = (dist - r->radius[k]) > c->surface_tol;
        for (int i = 0; i < 3; ++i) {
            const double elastic = -c->contact_k * (nrm[i] * pen);
            const double damping = -c->contact_nu * (nrm[i] * vn);
            double tot = resp[i] + elastic + damping;
            if (no_contact[k]) { resp[i] = 0.0; tot = 0.0; }
            e_tot[i][k] = tot;
        }
        resp_mag[k] = sqrt(resp[0] * resp[0] + resp[1] * resp[1] + resp[2] * resp[2]);
    }
    elements_to_nodes(r, e_tot);

    /* ---- kinetic friction (axial, rolling) ---- */
    double axial[3][NMAX], roll[3][NMAX], arm[3][NMAX], slip_ax[NMAX], slip_roll[NMAX];
    for (int k = 0; k < n; ++k) {
        double tn = 0.0, tp[3];
        for (int i = 0; i < 3; ++i) tn += nrm[i] * r->tang[i][k];
        for (int i = 0; i < 3; ++i) tp[i] = r->tang[i][k] - nrm[i] * tn;
        const double tpm = sqrt(tp[0] * tp[0] + tp[1] * tp[1] + tp[2] * tp[2]);
        for (int i = 0; i < 3; ++i) axial[i][k] = (1.0 / (tpm + 1e-14)) * tp[i];
        double ve[3] = { vel[0][k], vel[1][k], vel[2][k] };
        const double vax = ve[0] * axial[0][k] + ve[1] * axial[1][k] + ve[2] * axial[2][k];
        double vax_vec[3] = { vax * axial[0][k], vax * axial[1][k], vax * axial[2][k] };
        const double sgn = sign_of(vax);
        const double kmu = 0.5 * (c->kinetic_mu[0] * (1 + sgn) + c->kinetic_mu[1] * (1 - sgn));
        slip_ax[k] = slip_function(vax_vec, c->slip_velocity_tol);
        /* rolling direction = axial x normal ; torque arm = -normal * radius */
        roll[0][k] = axial[1][k] * nrm[2] - axial[2][k] * nrm[1];
        roll[1][k] = axial[2][k] * nrm[0] - axial[0][k] * nrm[2];
        roll[2][k] = axial[0][k] * nrm[1] - axial[1][k] * nrm[0];
        for (int i = 0; i < 3; ++i) arm[i][k] = -nrm[i] * r->radius[k];
        const double vroll = ve[0] * roll[0][k] + ve[1] * roll[1][k] + ve[2] * roll[2][k];
        /* rotation velocity Q^T (omega x (Q arm)) */
        double qa[3], wq[3], rot[3];
        for (int i = 0; i < 3; ++i)
            qa[i] = r->Q[i][0][k] * arm[0][k] + r->Q[i][1][k] * arm[1][k] + r->Q[i][2][k] * arm[2][k];
        wq[0] = r->w[1][k] * qa[2] - r->w[2][k] * qa[1];
        wq[1] = r->w[2][k] * qa[0] - r->w[0][k] * qa[2];
        wq[2] = r->w[0][k] * qa[1] - r->w[1][k] * qa[0];
        for (int i = 0; i < 3; ++i)
            rot[i] = r->Q[0][i][k] * wq[0] + r->Q[1][i][k] * wq[1] + r->Q[2][i][k] * wq[2];
        const double vrot = rot[0] * roll[0][k] + rot[1] * roll[1][k] + rot[2] * roll[2][k];
        const double sroll = vroll + vrot;
        double sroll_vec[3] = { sroll * roll[0][k], sroll * roll[1][k], sroll * roll[2][k] };
        slip_roll[k] = slip_function(sroll_vec, c->slip_velocity_tol);
        /* unitized_total_velocity = slip_velocity_along_rolling_direction + velocity_along_axial_direction;
         * unitized_total_velocity /= _batch_norm(unitized_total_velocity + 1e-14): the total SLIP
         * velocity in the plane (the rolling part includes the contact point's spin velocity), and
         * the 1e-14 is added to every component of the vector before its norm is taken */
        double u[3] = { sroll_vec[0] + vax_vec[0], sroll_vec[1] + vax_vec[1], sroll_vec[2] + vax_vec[2] };
        {
            const double t0 = u[0] + 1e-14, t1 = u[1] + 1e-14, t2 = u[2] + 1e-14;
            const double vm = sqrt(t0 * t0 + t1 * t1 + t2 * t2);
            u[0] /= vm; u[1] /= vm; u[2] /= vm;
        }
        const double uax = u[0] * axial[0][k] + u[1] * axial[1][k] + u[2] * axial[2][k];
        const double uro = u[0] * roll[0][k] + u[1] * roll[1][k] + u[2] * roll[2][k];
        double fk_ax[3], fk_ro[3];
        for (int i = 0; i < 3; ++i) {
            fk_ax[i] = -((1.0 - slip_ax[k]) * kmu * resp_mag[k] * uax * axial[i][k]);
            fk_ro[i] = -((1.0 - slip_roll[k]) * c->kinetic_mu[2] * resp_mag[k] * uro * roll[i][k]);
            if (no_contact[k]) { fk_ax[i] = 0.0; fk_ro[i] = 0.0; }
            e_tot[i][k] = fk_ax[i];
        }
        /* torque = Q (arm x F_roll) */
        double cr[3] = { arm[1][k] * fk_ro[2] - arm[2][k] * fk_ro[1],
                         arm[2][k] * fk_ro[0] - arm[0][k] * fk_ro[2],
                         arm[0][k] * fk_ro[1] - arm[1][k] * fk_ro[0] };
        for (int i = 0; i < 3; ++i)
            r->t_ext[i][k] += r->Q[i][0][k] * cr[0] + r->Q[i][1][k] * cr[1] + r->Q[i][2][k] * cr[2];
        for (int i = 0; i < 3; ++i) vel[i][k] = fk_ro[i]; /* reuse as scratch for the scatter */
    }
    elements_to_nodes(r, e_tot);                 /* axial kinetic */
    {
        double tmp[3][NMAX];
        for (int i = 0; i < 3; ++i) for (int k = 0; k < n; ++k) tmp[i][k] = vel[i][k];
        elements_to_nodes(r, tmp);               /* rolling kinetic */
    }

    /* ---- static friction (axial, rolling) on the updated total forces ---- */
    node_to_element_force(r, fel);
    double fs_ro[3][NMAX];
    for (int k = 0; k < n; ++k) {
        const double fax = fel[0][k] * axial[0][k] + fel[1][k] * axial[1][k] + fel[2][k] * axial[2][k];
        const double sg = sign_of(fax);
        const double smu = 0.5 * (c->static_mu[0] * (1 + sg) + c->static_mu[1] * (1 - sg));
        double maxf = slip_ax[k] * smu * resp_mag[k];
        const double mag = fabs(fax) < maxf ? fabs(fax) : maxf;
        for (int i = 0; i < 3; ++i) {
            double f = -(mag * sg * axial[i][k]);
            if (no_contact[k]) f = 0.0;
            e_tot[i][k] = f;
        }
        /* rolling: total torques in the lab frame, Q^T (tau_int + tau_ext) */
        double tt[3], tl[3];
        for (int i = 0; i < 3; ++i) tl[i] = r->t_int[i][k] + r->t_ext[i][k];
        for (int i = 0; i < 3; ++i)
            tt[i] = r->Q[0][i][k] * tl[0] + r->Q[1][i][k] * tl[1] + r->Q[2][i][k] * tl[2];
        const double tax = tt[0] * axial[0][k] + tt[1] * axial[1][k] + tt[2] * axial[2][k];
        const double fro = fel[0][k] * roll[0][k] + fel[1][k] * roll[1][k] + fel[2][k] * roll[2][k];
        const double noslip = -((r->radius[k] * fro - 2.0 * tax) / 3.0 / r->radius[k]);
        maxf = slip_roll[k] * c->static_mu[2] * resp_mag[k];
        const double sg2 = sign_of(noslip);
        const double mag2 = fabs(noslip) < maxf ? fabs(noslip) : maxf;
        for (int i = 0; i < 3; ++i) {
            double f = mag2 * sg2 * roll[i][k];
            if (no_contact[k]) f = 0.0;
            fs_ro[i][k] = f;
        }
    }
    elements_to_nodes(r, e_tot);                 /* axial static */
    elements_to_nodes(r, fs_ro);                 /* rolling static */
    for (int k = 0; k < n; ++k) {
        double cr[3] = { arm[1][k] * fs_ro[2][k] - arm[2][k] * fs_ro[1][k],
                         arm[2][k] * fs_ro[0][k] - arm[0][k] * fs_ro[2][k],
                         arm[0][k] * fs_ro[1][k] - arm[1][k] * fs_ro[0][k] };
        for (int i = 0; i < 3; ++i)
            r->t_ext[i][k] += r->Q[i][0][k] * cr[0] + r->Q[i][1][k] * cr[1] + r->Q[i][2][k] * cr[2];
    }
}

/* synchronize(): forcing in registration order — GravityForces (build.py:88-91)
 * then PendulumPointForces which ASSIGNS (build.py:100-101) */
/* MuscleTorquesWithVaryingBetaSplines.apply_torques, muscle_torques_with_bspline.py:128-176,
 * for the two instances SoftArmTrackingEnv.reset registers (soft_arm_tracking.py:352-383:
 * "normal" then "binormal").  `my_spline(cumulative_lengths)` is evaluated in the
 * piecewise-cubic form of the interpolant (oracle_set_spline_table). */
static void spline_muscle_torques(oracle_rod* r)
{
    const int n = r->n, nc = r->cfg.n_ctrl, np = r->cfg.n_spline_pieces;
    for (int d = 0; d < 2; ++d) {
        int same = 1;                                     /* np.array_equal(points_cached, points_array) :137 */
        for (int j = 0; j < nc; ++j) same = same && (r->pts_cached[d][j] == r->pts_input[d][j]);
        if (!same || !r->pts_init[d]) {
            r->pts_init[d] = 1;
            for (int j = 0; j < nc; ++j) {                /* filter_activation, :221-225 */
                const double diff = r->pts_input[d][j] - r->pts_cached[d][j];
                r->pts_cached[d][j] += sign_of(diff) * fmin(r->cfg.max_activation_rate, fabs(diff));
            }
            double cum = 0.0;
            for (int k = 0; k < n; ++k) {                 /* np.cumsum(system.lengths), :153 */
                cum += r->len[k];
                int p = 0;
                while (p + 1 < np && cum >= r->spline_breaks[p + 1]) ++p;
                const double ds = cum - r->spline_breaks[p];
                double val = 0.0;
                for (int j = 0; j < nc; ++j) {
                    const double* c = r->spline_coef[p][j];
                    val += r->pts_cached[d][j] * (((c[3] * ds + c[2]) * ds + c[1]) * ds + c[0]);
                }
                r->torque_mag[d][k] = r->cfg.muscle_torque_scale * val;   /* :156-158 */
            }
        }
        for (int k = 0; k < n; ++k) r->t_ext[d][k] += r->torque_mag[d][k];   /* compute_muscle_torques :199-201 */
    }
}

/* ------------------------------------------------------------------------- */
/* coomm.actuations.muscles.muscle.ApplyMuscles.apply_torques (COOMM, git pin   */
/* uv.lock:173-175, NOT on disk): RECALLED from coomm/actuations/muscles/        */
/* muscle.py (Muscle.__call__, MuscleForce, LongitudinalMuscle, TransverseMuscle) */
/* and coomm/actuations/actuation.py (ContinuousActuation, ApplyActuations,      */
/* internal_load_to_equivalent_external_load), anchored on the published model   */
/* (Chang et al., Proc. R. Soc. A 479:20220593, 2023, section 2(c): muscle force  */
/* F = u sigma_max A f_l(l), f_l(l) = max{3.06 l^3 - 13.64 l^2 + 18.01 l - 6.44,   */
/* 0}, applied along the muscle tangent at the offset x_m) and on the reference's */
/* call sites octopus/build.py:295-338, arm_push_env.py:197-212,247-274.          */
/* PARITY UNPINNED: every recalled detail is a switch of softrod_config.          */
/* ------------------------------------------------------------------------- */
static void apply_muscles(oracle_rod* r)
{
    const softrod_config* c = &r->cfg;
    const int n = r->n;
    double kav[3][NMAX];            /* average2D(kappa): Voronoi -> elements, half weights at both ends */
    double fi[3][NMAX], ce[3][NMAX];   /* internal force (local frame, elements), x_m x force (elements) */
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < n; ++k) {
            kav[i][k] = 0.0;
            fi[i][k] = 0.0;
            ce[i][k] = 0.0;
        }
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < n - 1; ++k) {
            kav[i][k] += 0.5 * r->kappa[i][k];
            kav[i][k + 1] += 0.5 * r->kappa[i][k];
        }
    for (int m = 0; m < c->n_muscles; ++m) {
        for (int k = 0; k < n; ++k) {
            /* Muscle.__call__: position, strain, tangent, length */
            const double rad = c->muscle_position_current_radius ? r->radius[k]
                             : sqrt(r->volume[k] / r->rest_len[k] / M_PI);
            double pos[3], nu[3];
            for (int i = 0; i < 3; ++i) pos[i] = rad * r->m_ratio[m][i][k];
            const double sh[3] = { r->sigma[0][k], r->sigma[1][k], r->sigma[2][k] + 1.0 };
            nu[0] = sh[0] + (kav[1][k] * pos[2] - kav[2][k] * pos[1]);
            nu[1] = sh[1] + (kav[2][k] * pos[0] - kav[0][k] * pos[2]);
            nu[2] = sh[2] + (kav[0][k] * pos[1] - kav[1][k] * pos[0]);
            const double nrm = sqrt(nu[0] * nu[0] + nu[1] * nu[1] + nu[2] * nu[2]);
            const double tg[3] = { nu[0] / nrm, nu[1] / nrm, nu[2] / nrm };
            double len = nrm;       /* muscle_rest_length = 1: the normalised length is the length */
            if (c->muscle_kind[m] == SOFTROD_MUSCLE_TRANSVERSE && c->muscle_tm_length_law == 0)
                len = 1.0 / sqrt(nrm);
            /* MuscleForce: force-length weight (clipped at zero), force */
            double w = 0.0;
            for (int p = c->muscle_fl_degree; p >= 0; --p) w = w * len + c->muscle_fl_coef[p];
            if (w < 0.0) w = 0.0;
            const double F = r->m_act[m][k] * r->m_strength[m][k] * w;
            r->m_force[m][k] = F;
            r->m_length[m][k] = len;
            /* ContinuousActuation: internal force along the muscle tangent, couple x_m x force */
            const double fm[3] = { F * tg[0], F * tg[1], F * tg[2] };
            for (int i = 0; i < 3; ++i) fi[i][k] += fm[i];
            ce[0][k] += pos[1] * fm[2] - pos[2] * fm[1];
            ce[1][k] += pos[2] * fm[0] - pos[0] * fm[2];
            ce[2][k] += pos[0] * fm[1] - pos[1] * fm[0];
        }
    }
    /* force_induced_couple: quadrature_kernel(x_m x f)[:, 1:-1] — element values averaged onto the Voronoi vertices */
    double cv[3][NMAX];
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < n - 1; ++k) cv[i][k] = 0.5 * (ce[i][k] + ce[i][k + 1]);
    /* internal_load_to_equivalent_external_load */
    const int pyel = c->muscle_equiv_load_form == 1;
    double cs[3][NMAX];
    for (int k = 0; k < n; ++k)
        for (int i = 0; i < 3; ++i) {
            double q = 0.0;
            for (int j = 0; j < 3; ++j) q += r->Q[j][i][k] * fi[j][k];
            cs[i][k] = pyel ? q / r->dil[k] : q;
        }
    for (int i = 0; i < 3; ++i) {                      /* difference_kernel */
        r->f_ext[i][0] += cs[i][0];
        for (int k = 1; k < n; ++k) r->f_ext[i][k] += cs[i][k] - cs[i][k - 1];
        r->f_ext[i][n] += -cs[i][n - 1];
    }
    double c2[3][NMAX], c3[3][NMAX];
    for (int k = 0; k < n - 1; ++k) {
        const double e3 = pyel ? 1.0 / (r->vdil[k] * r->vdil[k] * r->vdil[k]) : 1.0;
        for (int i = 0; i < 3; ++i) c2[i][k] = cv[i][k] * e3;
        c3[0][k] = (r->kappa[1][k] * cv[2][k] - r->kappa[2][k] * cv[1][k]) * r->rest_vor[k] * e3;
        c3[1][k] = (r->kappa[2][k] * cv[0][k] - r->kappa[0][k] * cv[2][k]) * r->rest_vor[k] * e3;
        c3[2][k] = (r->kappa[0][k] * cv[1][k] - r->kappa[1][k] * cv[0][k]) * r->rest_vor[k] * e3;
    }
    for (int k = 0; k < n; ++k) {
        double qt[3];
        for (int i = 0; i < 3; ++i) {
            double q = r->Q[i][0][k] * r->tang[0][k];
            q += r->Q[i][1][k] * r->tang[1][k];
            q += r->Q[i][2][k] * r->tang[2][k];
            qt[i] = pyel ? q : q * r->dil[k];
        }
        const double sc[3] = { (qt[1] * fi[2][k] - qt[2] * fi[1][k]) * r->rest_len[k],
                               (qt[2] * fi[0][k] - qt[0] * fi[2][k]) * r->rest_len[k],
                               (qt[0] * fi[1][k] - qt[1] * fi[0][k]) * r->rest_len[k] };
        for (int i = 0; i < 3; ++i) {
            double d2, d3;                             /* difference / quadrature kernels, Voronoi -> elements */
            if (k == 0) { d2 = c2[i][0]; d3 = 0.5 * c3[i][0]; }
            else if (k == n - 1) { d2 = -c2[i][n - 2]; d3 = 0.5 * c3[i][n - 2]; }
            else { d2 = c2[i][k] - c2[i][k - 1]; d3 = 0.5 * (c3[i][k] + c3[i][k - 1]); }
            r->t_ext[i][k] += d2 + d3 + sc[i];
        }
    }
}

static void apply_forcing(oracle_rod* r)
{
    const int n = r->n;
    if (r->cfg.features & SOFTROD_FEAT_GRAVITY)
        for (int i = 0; i < 3; ++i)
            for (int k = 0; k <= n; ++k) r->f_ext[i][k] += r->cfg.gravity[i] * r->mass[k];
    if (r->cfg.features & SOFTROD_FEAT_POINT_FORCE_NODE0_X) r->f_ext[0][0] = r->point_force;
    if (r->cfg.features & SOFTROD_FEAT_TIP_FORCE)
        for (int i = 0; i < 3; ++i) r->f_ext[i][n] += r->cfg.tip_force[i];
    if (r->cfg.features & SOFTROD_FEAT_SPLINE_MUSCLE_TORQUES) spline_muscle_torques(r);
    if (r->cfg.features & SOFTROD_FEAT_COOMM_MUSCLES) apply_muscles(r);
}

/* _update_accelerations + overload_operator_dynamic_numba (v += dt*a) */
static void dynamic_step(oracle_rod* r, double dt)
{
    const int n = r->n;
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k <= n; ++k) {
            const double a = (r->f_int[i][k] + r->f_ext[i][k]) / r->mass[k];
            r->v[i][k] += dt * a;
        }
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < n; ++k) {
            const double al = (r->invJ[i][k] * (r->t_int[i][k] + r->t_ext[i][k])) * r->dil[k];
            r->w[i][k] += dt * al;
        }
}

/* AnalyticalLinearDamper.dampen_rates (elastica/dissipation.py) */
static void dampen_rates(oracle_rod* r)
{
    const int n = r->n;
    if (!(r->cfg.features & SOFTROD_FEAT_ANALYTICAL_DAMPER)) return;
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k <= n; ++k) r->v[i][k] = r->v[i][k] * r->damp_t;
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k < n; ++k) r->w[i][k] = r->w[i][k] * pow(r->damp_r[i][k], r->dil[k]);
}

/* LaplaceDissipationFilter.dampen_rates -> nb_filter_rate (elastica/dissipation.py),
 * registered at soft_pendulum_3d/build.py:82-85: applied to velocity (n+1 entries)
 * and omega (n entries) */
static void filter_rate(double* rate, int m, int order)
{
    double f[NMAX + 1], g[NMAX + 1];
    for (int k = 0; k < m; ++k) f[k] = rate[k];
    for (int it = 0; it < order; ++it) {
        for (int k = 1; k < m - 1; ++k) g[k] = (-f[k + 1] - f[k - 1] + 2.0 * f[k]) / 4.0;
        for (int k = 1; k < m - 1; ++k) f[k] = g[k];
        f[0] = 0.0;
        f[m - 1] = 0.0;
    }
    for (int k = 0; k < m; ++k) rate[k] = rate[k] - f[k];
}

/* exported for tests: the bare filter on one array */
void oracle_filter_rate(double* rate, int m, int order) { filter_rate(rate, m, order); }

static void laplace_filter(oracle_rod* r)
{
    if (!(r->cfg.features & SOFTROD_FEAT_LAPLACE_FILTER)) return;
    for (int i = 0; i < 3; ++i) filter_rate(r->v[i], r->n + 1, r->cfg.filter_order);
    for (int i = 0; i < 3; ++i) filter_rate(r->w[i], r->n, r->cfg.filter_order);
}

/* PositionVerlet().step(simulator, time, dt) — elastica/timestepper/
 * symplectic_steppers.py SymplecticStepperMethods.do_step; call site
 * soft_pendulum.py:184 */
static void position_verlet_step(oracle_rod* r)
{
    const double dt = r->cfg.dt;
    const int n = r->n;
    kinematic_step(r, 0.5 * dt);
    if (r->cfg.time_two_half_adds) r->time += 0.5 * dt;
    constrain_values(r);
    compute_internal_forces(r);
    compute_internal_torques(r);
    /* synchronize(): operators in registration order — add_forcing_to(GravityForces)
     * precedes detect_contact_between(...) in build_arm (octopus/build.py:236-283) */
    const int has_contact = (r->cfg.features & SOFTROD_FEAT_PLANE_CONTACT_ANISO) != 0;
    if (has_contact && r->cfg.contact_before_forcing) plane_contact(r);
    apply_forcing(r);
    if (has_contact && !r->cfg.contact_before_forcing) plane_contact(r);
    dynamic_step(r, dt);
    /* _feature_group_constrain_rates: Damping registers before Constraints for
     * the mixin order of BaseSimulator (soft_pendulum.py:34-42); the two
     * commute exactly for this env (zeros vs. scaling). */
    if (r->cfg.damp_before_constrain) {
        dampen_rates(r);   /* dampers in registration order: analytical, then Laplace */
        laplace_filter(r);
        constrain_rates(r);
    } else {
        constrain_rates(r);
        dampen_rates(r);
        laplace_filter(r);
    }
    kinematic_step(r, 0.5 * dt);
    if (r->cfg.time_two_half_adds) r->time += 0.5 * dt; else r->time += dt;
    constrain_values(r);
    for (int i = 0; i < 3; ++i) {
        for (int k = 0; k <= n; ++k) r->f_ext[i][k] = 0.0;
        for (int k = 0; k < n; ++k) r->t_ext[i][k] = 0.0;
    }
    if (r->round_state_f32) {
        for (int i = 0; i < 3; ++i) {
            for (int k = 0; k <= n; ++k) { r->x[i][k] = (double)(float)r->x[i][k]; r->v[i][k] = (double)(float)r->v[i][k]; }
            for (int k = 0; k < n; ++k) {
                r->w[i][k] = (double)(float)r->w[i][k];
                for (int j = 0; j < 3; ++j) r->Q[i][j][k] = (double)(float)r->Q[i][j][k];
            }
        }
    }
}

static int substeps_to_run(const oracle_rod* r)
{
    return r->run_substeps >= 0 ? r->run_substeps : r->cfg.n_substeps;
}
void oracle_set_run_substeps(oracle_rod* r, int n) { r->run_substeps = n; }
void oracle_set_round_state_f32(oracle_rod* r, int on) { r->round_state_f32 = on; }

/* ------------------------------------------------------------------------- */
/* env epilogue: soft_pendulum.py:149-161 (get_state) and :196-251             */
/* ------------------------------------------------------------------------- */
static double py_mod(double a, double b)
{
    double m = fmod(a, b);
    if (m != 0.0 && ((m < 0.0) != (b < 0.0))) m += b;
    return m;
}

static double wrapped_theta(const oracle_rod* r)
{
    const int n = r->n;
    double tm[2] = { 0.0, 0.0 };
    for (int k = 0; k < n; ++k) { tm[0] += r->tang[0][k]; tm[1] += r->tang[1][k]; }
    tm[0] /= (double)n; tm[1] /= (double)n;
    double theta = atan(tm[0] / tm[1]);
    return py_mod(theta + M_PI, 2.0 * M_PI) - M_PI;
}

static void get_state(const oracle_rod* r, float obs[4])
{
    obs[0] = (float)r->x[0][0];
    obs[1] = (float)r->v[0][0];
    obs[2] = r->prev_action;
    obs[3] = (float)wrapped_theta(r);
}

/* ------------------------------------------------------------------------- */
/* exported API (loaded with ctypes by tests/ and bench.py cpu_baseline)       */
/* ------------------------------------------------------------------------- */
oracle_rod* oracle_create(const softrod_config* cfg)
{
    if (!cfg || cfg->struct_size != sizeof(softrod_config)) return NULL;
    if (cfg->n_elem < 2 || cfg->n_elem > NMAX) return NULL;
    oracle_rod* r = (oracle_rod*)calloc(1, sizeof(oracle_rod));
    if (!r) return NULL;
    r->cfg = *cfg;
    r->n = cfg->n_elem;
    r->run_substeps = -1;
    return r;
}

void oracle_destroy(oracle_rod* r) { free(r); }

void oracle_reset_straight(oracle_rod* r, const double start[3], const double direction[3],
                           const double normal[3])
{
    straight_rod(r, start, direction, normal);
    /* CosseratRod.__init__ evaluates strains once at allocation, so
     * rod.tangents is valid for the reset observation (soft_pendulum.py:145) */
    compute_shear_stress(r);
    compute_bending_twist_strains(r);
}

/* build.py:46-52 */
void oracle_reset_pendulum(oracle_rod* r, double theta)
{
    const double start[3] = { 0.0, 0.0, 0.0 };
    const double direction[3] = { 1.0 * cos(theta), 1.0 * sin(theta), 0.0 };
    const double normal[3] = { 1.0 * sin(theta), -1.0 * cos(theta), 0.0 };
    oracle_reset_straight(r, start, direction, normal);
}

/* re-evaluate the strain caches at the CURRENT configuration (tests: energy audits);
 * the stepping path itself leaves them stale by half a substep, as PyElastica does */
void oracle_refresh_strains(oracle_rod* r)
{
    compute_shear_stress(r);
    compute_bending_twist_strains(r);
}

void oracle_set_prev_action(oracle_rod* r, float a) { r->prev_action = a; }
void oracle_observe(const oracle_rod* r, float obs[4]) { get_state(r, obs); }
double oracle_time(const oracle_rod* r) { return r->time; }

void oracle_substeps(oracle_rod* r, float action, int nsub)
{
    r->point_force = (double)action;
    for (int s = 0; s < nsub; ++s) position_verlet_step(r);
}

/* SoftPendulumEnv.step, soft_pendulum.py:176-251 */
void oracle_env_step(oracle_rod* r, float action, float obs[4], double* reward,
                     uint8_t* terminated, uint8_t* truncated)
{
    const int n = r->n;
    r->prev_action = action;          /* :165 */
    r->point_force = (double)action;  /* :166 (float32 value held in float64) */
    for (int s = 0; s < substeps_to_run(r); ++s) position_verlet_step(r);
    int invalid = 0;
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k <= n; ++k)
            if (isnan(r->x[i][k]) || isnan(r->v[i][k])) invalid = 1;
    double survive = 0.0, forward = 0.0;
    *terminated = 0;
    if (invalid) { *terminated = 1; survive = -50.0; }
    else {
        const double dist = fabs(r->x[0][0]);
        const double th = wrapped_theta(r);
        forward = dist * 10 + th * th;
    }
    *truncated = (r->time > r->cfg.final_time) ? 1 : 0;
    *reward = forward - 0.0 + survive;
    get_state(r, obs);
}

/* ---- SoftPendulum3D-v0: soft_pendulum_3d/soft_pendulum_3d.py ---- */
static double tilt_angle(const oracle_rod* r) /* :88-91 */
{
    const int n = r->n;
    double tm[3] = { 0.0, 0.0, 0.0 };
    for (int k = 0; k < n; ++k) for (int i = 0; i < 3; ++i) tm[i] += r->tang[i][k];
    for (int i = 0; i < 3; ++i) tm[i] /= (double)n;
    const double nrm = sqrt(tm[0] * tm[0] + tm[1] * tm[1] + tm[2] * tm[2]);
    double tz = tm[2] / nrm;
    if (tz < -1.0) tz = -1.0;
    if (tz > 1.0) tz = 1.0;
    return acos(tz);
}

static void get_state3d(const oracle_rod* r, float obs[9]) /* :93-98 */
{
    for (int i = 0; i < 3; ++i) { obs[i] = (float)r->x[i][0]; obs[3 + i] = (float)r->v[i][0]; }
    obs[6] = r->prev_action2[0];
    obs[7] = r->prev_action2[1];
    obs[8] = (float)tilt_angle(r);
}

void oracle_observe3d(const oracle_rod* r, float obs[9]) { get_state3d(r, obs); }
void oracle_clear_prev_action3d(oracle_rod* r) { r->prev_action2[0] = r->prev_action2[1] = 0.0f; }

/* build_soft_pendulum_3d, soft_pendulum_3d/build.py:51-64 */
void oracle_reset_pendulum3d(oracle_rod* r, double tilt)
{
    const double start[3] = { 0.0, 0.0, 0.0 };
    const double direction[3] = { sin(tilt), 0.0, cos(tilt) };
    const double normal[3] = { 0.0, 1.0, 0.0 };
    oracle_reset_straight(r, start, direction, normal);
    oracle_clear_prev_action3d(r); /* soft_pendulum_3d.py:68 */
}

/* SoftPendulum3DEnv.step, soft_pendulum_3d.py:115-174 (action validity is checked by
 * the caller, :116-117) */
void oracle_env_step3d(oracle_rod* r, const float action[2], float obs[9], double* reward,
                       uint8_t* terminated, uint8_t* truncated, double* tilt_out)
{
    const int n = r->n;
    /* set_action, :99-113 */
    double next[3] = { r->ctrl_pos[0], r->ctrl_pos[1], r->ctrl_pos[2] };
    for (int i = 0; i < 2; ++i) {
        const float disp = (float)r->cfg.base_step * action[i]; /* float32 product */
        double v = next[i] + (double)disp;
        if (v < -r->cfg.base_limit) v = -r->cfg.base_limit; /* np.clip */
        if (v > r->cfg.base_limit) v = r->cfg.base_limit;
        next[i] = v;
    }
    const double step_time = (double)r->cfg.n_substeps * r->cfg.dt;
    for (int i = 0; i < 3; ++i) {
        const double actual = next[i] - r->ctrl_pos[i];
        r->ctrl_pos[i] = next[i];
        r->ctrl_vel[i] = actual / step_time;
    }
    r->prev_action2[0] = action[0];
    r->prev_action2[1] = action[1];
    for (int s = 0; s < substeps_to_run(r); ++s) position_verlet_step(r);
    int invalid = 0;
    for (int i = 0; i < 3; ++i)
        for (int k = 0; k <= n; ++k)
            if (isnan(r->x[i][k]) || isnan(r->v[i][k])) invalid = 1;
    const double tilt = tilt_angle(r);
    const double bd = sqrt(r->ctrl_pos[0] * r->ctrl_pos[0] + r->ctrl_pos[1] * r->ctrl_pos[1]);
    /* 1e-3 * np.dot(action, action): float32 under NumPy 2 promotion rules */
    const float ctl = 1e-3f * (action[0] * action[0] + action[1] * action[1]);
    double rew = -(tilt * tilt + 0.1 * (bd * bd) + (double)ctl);
    *terminated = invalid ? 1 : 0;
    *truncated = (r->time >= r->cfg.final_time) ? 1 : 0;
    if (invalid) rew = -50.0;
    *reward = rew;
    *tilt_out = tilt;
    get_state3d(r, obs);
}

/* ---- OctoArmSingle-v0: octopus/arm_single_env.py ---- */
static void center_of_mass(const oracle_rod* r, double com[3]) /* compute_position_center_of_mass */
{
    double msum = 0.0;
    for (int k = 0; k <= r->n; ++k) msum += r->mass[k];
    for (int i = 0; i < 3; ++i) {
        double s = 0.0;
        for (int k = 0; k <= r->n; ++k) s += r->mass[k] * r->x[i][k];
        com[i] = s / msum;
    }
}

/* get_state, :186-219.  Mutates prev_kappa_state / prev_com_state like the reference.
 * The 7x7 reshape of the reference (:193-194) is generalised to np.array_split-style
 * bins so that n_elem - 1 need not be 49 (identical for 49). */
static void get_state_arm(oracle_rod* r, float obs[25])
{
    const softrod_config* c = &r->cfg;
    const int nv = r->n - 1;
    double mk[7], mr[7];
    int lo = 0;
    for (int b = 0; b < 7; ++b) {
        const int sz = nv / 7 + (b < nv % 7 ? 1 : 0);
        double sk = 0.0, sr = 0.0;
        for (int k = lo; k < lo + sz; ++k) {
            sk += r->kappa[0][k];
            sr += r->kappa[0][k] - r->prev_kappa[k];
        }
        mk[b] = sk / (double)sz;
        mr[b] = sr / (double)sz;
        lo += sz;
    }
    for (int k = 0; k < nv; ++k) r->prev_kappa[k] = r->kappa[0][k];
    double com[3];
    center_of_mass(r, com);
    const double cr0 = com[0] - r->prev_com[0], cr1 = com[1] - r->prev_com[1];
    r->prev_com[0] = com[0]; r->prev_com[1] = com[1];
    for (int b = 0; b < 7; ++b) {
        obs[b] = (float)((mk[b] - c->kappa_range[0]) / (c->kappa_range[1] - c->kappa_range[0]));
        obs[7 + b] = (float)((mr[b] - c->kappa_rate_range[0]) /
                             (c->kappa_rate_range[1] - c->kappa_rate_range[0]));
    }
    obs[14] = (float)cr0; obs[15] = (float)cr1;
    for (int i = 0; i < 7; ++i) obs[16 + i] = r->prev_action7[i];
    obs[23] = (float)c->target[0]; obs[24] = (float)c->target[1];
}

/* ArmSingleEnv.reset, :135-183 (build_arm: octopus/build.py:220-292) */
void oracle_reset_arm(oracle_rod* r, float obs[25])
{
    const double start[3] = { 0.0, 0.0, 0.0 };
    const double direction[3] = { 1.0, 0.0, 0.0 };
    const double normal[3] = { 0.0, 0.0, 1.0 };
    oracle_reset_straight(r, start, direction, normal);
    for (int k = 0; k < r->n - 1; ++k) r->prev_kappa[k] = r->kappa[0][k];
    double com[3];
    center_of_mass(r, com);
    r->prev_com[0] = com[0]; r->prev_com[1] = com[1];
    get_state_arm(r, obs);
}

/* ArmSingleEnv.step, :237-316.  rest_kappa0: the interp1d output of set_action
 * (:226-235), computed by the caller with scipy exactly as the reference does. */
void oracle_env_step_arm(oracle_rod* r, const float action[7], const double* rest_kappa0,
                         float obs[25], double* reward, uint8_t* terminated, uint8_t* truncated)
{
    const softrod_config* c = &r->cfg;
    const int n = r->n;
    for (int i = 0; i < 7; ++i) r->prev_action7[i] = action[i];
    for (int k = 0; k < n - 1; ++k) r->rest_kappa[0][k] = rest_kappa0[k];
    for (int s = 0; s < substeps_to_run(r); ++s) position_verlet_step(r);
    /* control penalty: float32 arithmetic (np.square/mean on the float32 action, and
     * python-float * np.float32 stays float32 under NumPy 2 promotion) */
    float sq = 0.0f;
    for (int i = 0; i < 7; ++i) sq += action[i] * action[i];
    const float pen = (float)c->control_penalty_coeff * (sq / 7.0f);
    int invalid = 0;
    double wn = 0.0;
    for (int i = 0; i < 3; ++i) {
        for (int k = 0; k <= n; ++k)
            if (isnan(r->x[i][k]) || isnan(r->v[i][k])) invalid = 1;
        for (int k = 0; k < n; ++k) wn += r->w[i][k] * r->w[i][k];
    }
    if (sqrt(wn) > 250) invalid = 1;
    double survive = 0.0, forward = 0.0;
    *terminated = 0;
    if (invalid) { *terminated = 1; survive = -1.0; }
    else {
        double com[3];
        center_of_mass(r, com);
        const double dx = com[0] - c->target[0], dy = com[1] - c->target[1];
        const double dist = sqrt(dx * dx + dy * dy);
        forward = exp(-dist / 0.35) - 0.096;
        if (dist < 0.1) { survive = 5.0; *terminated = 1; }
    }
    *truncated = (r->time > c->final_time) ? 1 : 0;
    *reward = forward - (double)pen + survive;
    /* invalid branch: forward_reward is still the Python float 0.0, so `0.0 - np.float32 + (-1.0)`
     * stays float32 under NumPy 2 promotion (:255-259,274-276,296); every other branch has a
     * float64 forward_reward */
    if (invalid) *reward = (double)((0.0f - pen) + (-1.0f));
    get_state_arm(r, obs);
}

/* batched driver for the cpu_baseline leg (OpenMP over rods when built with it) */
void oracle_env_step_batch(oracle_rod** rods, int n_rods, const float* actions, float* obs,
                           double* reward, uint8_t* terminated, uint8_t* truncated)
{
#pragma omp parallel for schedule(static)
    for (int e = 0; e < n_rods; ++e)
        oracle_env_step(rods[e], actions[e], obs + 4 * e, reward + e, terminated + e,
                        truncated + e);
}

/* the same for OctoArmSingle (rest_kappa0: [n_rods][n_elem - 1], the interp1d output per rod) */
void oracle_env_step_arm_batch(oracle_rod** rods, int n_rods, const float* actions, const double* rest_kappa0,
                               float* obs, double* reward, uint8_t* terminated, uint8_t* truncated)
{
#pragma omp parallel for schedule(dynamic, 8)
    for (int e = 0; e < n_rods; ++e)
        oracle_env_step_arm(rods[e], actions + 7 * e, rest_kappa0 + (size_t)(rods[e]->n - 1) * e, obs + 25 * e,
                            reward + e, terminated + e, truncated + e);
}

#define COPY3X(arr, cnt) do { for (int i = 0; i < 3; ++i) for (int k = 0; k < (cnt); ++k) \
        out[i * (cnt) + k] = r->arr[i][k]; return 3 * (cnt); } while (0)
/* field access for tests: name in {x,v,Q,w,tangents,kappa,sigma,mass,f_int,t_int,...} */
int oracle_get(const oracle_rod* r, const char* name, double* out)
{
    const int n = r->n;
#define COPY3(arr, cnt) do { for (int i = 0; i < 3; ++i) for (int k = 0; k < (cnt); ++k) \
        out[i * (cnt) + k] = r->arr[i][k]; return 3 * (cnt); } while (0)
    if (!strcmp(name, "x")) COPY3(x, n + 1);
    if (!strcmp(name, "v")) COPY3(v, n + 1);
    if (!strcmp(name, "w")) COPY3(w, n);
    if (!strcmp(name, "tangents")) COPY3(tang, n);
    if (!strcmp(name, "sigma")) COPY3(sigma, n);
    if (!strcmp(name, "kappa")) COPY3(kappa, n - 1);
    if (!strcmp(name, "n_int")) COPY3(n_int, n);
    if (!strcmp(name, "m_int")) COPY3(m_int, n - 1);
    if (!strcmp(name, "f_int")) COPY3(f_int, n + 1);
    if (!strcmp(name, "t_int")) COPY3(t_int, n);
    if (!strcmp(name, "J")) COPY3(J, n);
    if (!strcmp(name, "shear")) COPY3(shear, n);
    if (!strcmp(name, "bend")) COPY3(bend, n - 1);
    if (!strcmp(name, "damp_r")) COPY3(damp_r, n);
#undef COPY3
    if (!strcmp(name, "Q")) {
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < n; ++k)
            out[(i * 3 + j) * n + k] = r->Q[i][j][k];
        return 9 * n;
    }
    if (!strcmp(name, "mass")) { for (int k = 0; k <= n; ++k) out[k] = r->mass[k]; return n + 1; }
    if (!strcmp(name, "lengths")) { for (int k = 0; k < n; ++k) out[k] = r->len[k]; return n; }
    if (!strcmp(name, "dilatation")) { for (int k = 0; k < n; ++k) out[k] = r->dil[k]; return n; }
    if (!strcmp(name, "rest_lengths")) { for (int k = 0; k < n; ++k) out[k] = r->rest_len[k]; return n; }
    if (!strcmp(name, "damp_t")) { out[0] = r->damp_t; return 1; }
    if (!strcmp(name, "voronoi_dilatation")) { for (int k = 0; k < n - 1; ++k) out[k] = r->vdil[k]; return n - 1; }
    if (!strcmp(name, "muscle_force")) { for (int m = 0; m < SOFTROD_MAX_MUSCLES; ++m) for (int k = 0; k < n; ++k) out[m * n + k] = r->m_force[m][k]; return SOFTROD_MAX_MUSCLES * n; }
    if (!strcmp(name, "muscle_length")) { for (int m = 0; m < SOFTROD_MAX_MUSCLES; ++m) for (int k = 0; k < n; ++k) out[m * n + k] = r->m_length[m][k]; return SOFTROD_MAX_MUSCLES * n; }
    if (!strcmp(name, "muscle_activation")) { for (int m = 0; m < SOFTROD_MAX_MUSCLES; ++m) for (int k = 0; k < n; ++k) out[m * n + k] = r->m_act[m][k]; return SOFTROD_MAX_MUSCLES * n; }
    if (!strcmp(name, "sucker_index")) { for (int j = 0; j < SOFTROD_MAX_SUCKERS; ++j) out[j] = (double)r->sucker_index[j]; return SOFTROD_MAX_SUCKERS; }
    if (!strcmp(name, "sucker_ratio")) { for (int j = 0; j < SOFTROD_MAX_SUCKERS; ++j) out[j] = r->sucker_ratio[j]; return SOFTROD_MAX_SUCKERS; }
    if (!strcmp(name, "rest_kappa")) COPY3X(rest_kappa, n - 1);
    if (!strcmp(name, "radius")) { for (int k = 0; k < n; ++k) out[k] = r->radius[k]; return n; }
    if (!strcmp(name, "f_ext")) COPY3X(f_ext, n + 1);
    if (!strcmp(name, "t_ext")) COPY3X(t_ext, n);
    if (!strcmp(name, "control")) {
        out[0] = r->ctrl_pos[0]; out[1] = r->ctrl_pos[1];
        out[2] = r->ctrl_vel[0]; out[3] = r->ctrl_vel[1];
        return 4;
    }
    if (!strcmp(name, "time")) { out[0] = r->time; return 1; }
    if (!strcmp(name, "prev_kappa")) { for (int k = 0; k < n - 1; ++k) out[k] = r->prev_kappa[k]; return n - 1; }
    if (!strcmp(name, "prev_com")) { out[0] = r->prev_com[0]; out[1] = r->prev_com[1]; return 2; }
    if (!strcmp(name, "prev_action7")) { for (int i = 0; i < 7; ++i) out[i] = (double)r->prev_action7[i]; return 7; }
    if (!strcmp(name, "prev_action2")) { out[0] = r->prev_action2[0]; out[1] = r->prev_action2[1]; return 2; }
    if (!strcmp(name, "prev_action")) { out[0] = r->prev_action; return 1; }
    if (!strcmp(name, "fixed_pos")) { for (int i = 0; i < 3; ++i) out[i] = r->fixed_pos[i]; return 3; }
    if (!strcmp(name, "fixed_dir")) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) out[3 * i + j] = r->fixed_dir[i][j]; return 9; }
    return -1;
}

/* state injection for tests (perturbed initial conditions) */
int oracle_set(oracle_rod* r, const char* name, const double* in)
{
    const int n = r->n;
#define SET3X(arr, cnt) do { for (int i = 0; i < 3; ++i) for (int k = 0; k < (cnt); ++k) \
        r->arr[i][k] = in[i * (cnt) + k]; return 0; } while (0)
#define SET3(arr, cnt) do { for (int i = 0; i < 3; ++i) for (int k = 0; k < (cnt); ++k) \
        r->arr[i][k] = in[i * (cnt) + k]; return 0; } while (0)
    if (!strcmp(name, "x")) SET3(x, n + 1);
    if (!strcmp(name, "v")) SET3(v, n + 1);
    if (!strcmp(name, "w")) SET3(w, n);
    if (!strcmp(name, "rest_kappa")) SET3(rest_kappa, n - 1);
    /* caches and env memory: lets a test put the oracle into exactly the state a recorded
     * reference epilogue saw (tests/golden/ref_*.npz) */
    if (!strcmp(name, "tangents")) SET3(tang, n);
    if (!strcmp(name, "kappa")) SET3(kappa, n - 1);
    if (!strcmp(name, "f_ext")) SET3(f_ext, n + 1);
    /* damper coefficients: lets tools/sweep_switches.py try AnalyticalLinearDamper's other
     * protocol (uniform: the same exp(-nu dt) on every rate) without a config field */
    if (!strcmp(name, "damp_r")) SET3(damp_r, n);
#undef SET3
    if (!strcmp(name, "damp_t")) { r->damp_t = in[0]; return 0; }
    if (!strcmp(name, "sigma")) SET3X(sigma, n);
    if (!strcmp(name, "muscle_activation")) { for (int m = 0; m < SOFTROD_MAX_MUSCLES; ++m) for (int k = 0; k < n; ++k) r->m_act[m][k] = in[m * n + k]; return 0; }
    if (!strcmp(name, "sucker_index")) { for (int j = 0; j < SOFTROD_MAX_SUCKERS; ++j) r->sucker_index[j] = (int)in[j]; return 0; }
    if (!strcmp(name, "prev_action_push")) { r->prev_action_push[0] = (float)in[0]; r->prev_action_push[1] = (float)in[1]; return 0; }
    if (!strcmp(name, "time")) { r->time = in[0]; return 0; }
    if (!strcmp(name, "prev_kappa")) { for (int k = 0; k < n - 1; ++k) r->prev_kappa[k] = in[k]; return 0; }
    if (!strcmp(name, "prev_com")) { r->prev_com[0] = in[0]; r->prev_com[1] = in[1]; return 0; }
    if (!strcmp(name, "prev_action7")) { for (int i = 0; i < 7; ++i) r->prev_action7[i] = (float)in[i]; return 0; }
    if (!strcmp(name, "prev_action2")) { r->prev_action2[0] = (float)in[0]; r->prev_action2[1] = (float)in[1]; return 0; }
    if (!strcmp(name, "prev_action")) { r->prev_action = (float)in[0]; return 0; }
    if (!strcmp(name, "control")) {
        r->ctrl_pos[0] = in[0]; r->ctrl_pos[1] = in[1]; r->ctrl_vel[0] = in[2]; r->ctrl_vel[1] = in[3];
        return 0;
    }
    if (!strcmp(name, "fixed_pos")) { for (int i = 0; i < 3; ++i) r->fixed_pos[i] = in[i]; return 0; }
    if (!strcmp(name, "fixed_dir")) { for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) r->fixed_dir[i][j] = in[3 * i + j]; return 0; }
    if (!strcmp(name, "Q")) {
        for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) for (int k = 0; k < n; ++k)
            r->Q[i][j][k] = in[(i * 3 + j) * n + k];
        return 0;
    }
    return -1;
}

/* Operator probes for the fixtures recorded from the reference's own classes
 * (tests/golden/ref_*.npz): one application of constrain_values + constrain_rates
 * (build.py:71-79, soft_pendulum_3d/build.py:31-39), or of the forcing group on a
 * prefilled external_forces (build.py:88-105: gravity adds, the point force assigns). */
void oracle_constrain_probe(oracle_rod* r) { constrain_values(r); constrain_rates(r); }
void oracle_set_radius_profile(oracle_rod* r, const double* radius)
{
    for (int k = 0; k < r->n; ++k) r->radius_profile[k] = radius[k];
    r->has_profile = 1;
}
void oracle_set_sucker_ratio(oracle_rod* r, const double* ratio)
{
    for (int j = 0; j < SOFTROD_MAX_SUCKERS; ++j) r->sucker_ratio[j] = ratio[j];
}
void oracle_forcing_probe(oracle_rod* r, double point_force)
{
    r->point_force = point_force;
    apply_forcing(r);
}

/* ------------------------------------------------------------------------- */
/* SoftArmTracking-v0 (game_mode 1): soft_arm/soft_arm_tracking.py              */
/* ------------------------------------------------------------------------- */
void oracle_set_spline_table(oracle_rod* r, const double* breaks, const double* coef)
{
    const int np = r->cfg.n_spline_pieces, nc = r->cfg.n_ctrl;
    for (int p = 0; p <= np; ++p) r->spline_breaks[p] = breaks[p];
    for (int p = 0; p < np; ++p)
        for (int j = 0; j < nc; ++j)
            for (int q = 0; q < 4; ++q) r->spline_coef[p][j][q] = coef[(p * nc + j) * 4 + q];
}

/* get_state, :160-207 (float64, as the reference's observation space) */
static void get_state_soft_arm(const oracle_rod* r, double* obs)
{
    const int n = r->n, ns = r->cfg.n_ctrl;       /* number_of_observation_segments = control points */
    const int avg_length = (n - 1) / ns;
    for (int c = 0; c < 2; ++c)
        for (int i = 0; i < ns; ++i) {
            const int lo = avg_length * i, hi = (i == ns - 1) ? (n - 1) : avg_length * (i + 1);
            double s = 0.0;
            for (int k = lo; k < hi; ++k) s += r->kappa[c][k];
            obs[c * ns + i] = (s / (double)(hi - lo)) * r->cfg.base_length / (2.0 * M_PI);
        }
    for (int i = 0; i < 3; ++i) {
        obs[2 * ns + i] = r->x[i][n] / r->cfg.base_length;
        obs[2 * ns + 3 + i] = r->arm_target[i] / 1000.0;
    }
}

void oracle_reset_soft_arm(oracle_rod* r, double* obs)   /* reset, :261-282,386-483 */
{
    const double start[3] = { 0.0, 0.0, 0.0 }, direction[3] = { 0.0, 1.0, 0.0 }, normal[3] = { 0.0, 0.0, 1.0 };
    oracle_reset_straight(r, start, direction, normal);
    r->time = 0.0;
    r->tick = 0;
    for (int d = 0; d < 2; ++d) {
        r->pts_init[d] = 0;
        for (int j = 0; j < SOFTROD_MAX_CTRL; ++j) r->pts_cached[d][j] = r->pts_input[d][j] = 0.0;
        for (int k = 0; k < NMAX; ++k) r->torque_mag[d][k] = 0.0;
    }
    for (int i = 0; i < 3; ++i) r->arm_target[i] = r->cfg.arm_target[i];
    get_state_soft_arm(r, obs);
}

/* test probe: one call of the two muscles' apply_torques with the given control points and
 * element lengths (tests/golden/softarm_vectors.npz holds what the reference's own class gives) */
void oracle_spline_torque_probe(oracle_rod* r, const double* points, const double* lengths,
                                double* torques /* [3][n] */, double* cached /* [2 n_ctrl] */)
{
    const int n = r->n, nc = r->cfg.n_ctrl;
    for (int j = 0; j < nc; ++j) { r->pts_input[0][j] = points[j]; r->pts_input[1][j] = points[nc + j]; }
    for (int k = 0; k < n; ++k) r->len[k] = lengths[k];
    for (int i = 0; i < 3; ++i) for (int k = 0; k < n; ++k) r->t_ext[i][k] = 0.0;
    spline_muscle_torques(r);
    for (int i = 0; i < 3; ++i) for (int k = 0; k < n; ++k) { torques[i * n + k] = r->t_ext[i][k]; r->t_ext[i][k] = 0.0; }
    for (int j = 0; j < nc; ++j) { cached[j] = r->pts_cached[0][j]; cached[nc + j] = r->pts_cached[1][j]; }
}

void oracle_set_arm_target(oracle_rod* r, const double t[3]) { for (int i = 0; i < 3; ++i) r->arm_target[i] = t[i]; }
void oracle_observe_soft_arm(const oracle_rod* r, double* obs) { get_state_soft_arm(r, obs); }

/* step, :209-259.  The target sphere does not interact with the rod (it is appended to the
 * simulator without a connection, :428-436, and its state is overwritten every substep,
 * :223-224), so only its position enters — through the reward and the observation. */
void oracle_env_step_soft_arm(oracle_rod* r, const float* action, double* obs, double* reward,
                              uint8_t* terminated, uint8_t* truncated)
{
    const softrod_config* c = &r->cfg;
    const int n = r->n, nc = c->n_ctrl;
    for (int j = 0; j < nc; ++j) {
        r->pts_input[0][j] = (double)action[j];
        r->pts_input[1][j] = (double)action[nc + j];
    }
    for (int s = 0; s < c->n_substeps; ++s) {
        position_verlet_step(r);
        r->tick += 1;
    }
    double d2 = 0.0;
    for (int i = 0; i < 3; ++i) {
        const double d = (r->arm_target[i] - r->x[i][n]) / 1000.0;
        d2 += d * d;
    }
    const double nrm = sqrt(d2);                  /* -np.square(np.linalg.norm(tip_to_target)) */
    *reward = -(nrm * nrm);
    get_state_soft_arm(r, obs);
    *terminated = 0;
    int invalid = 0;
    for (int i = 0; i < 2 * nc + 6; ++i) invalid = invalid || isnan(obs[i]);
    if (invalid) {
        *reward = -100.0;
        for (int i = 0; i < 2 * nc + 6; ++i)      /* np.nan_to_num */
            obs[i] = isnan(obs[i]) ? 0.0 : (isinf(obs[i]) ? copysign(1.7976931348623157e308, obs[i]) : obs[i]);
        *terminated = 1;
    }
    *truncated = ((double)r->tick * c->dt >= c->final_time) ? 1 : 0;
}

/* ------------------------------------------------------------------------- */
/* COOMM muscle layers + ArmPushEnv (octopus/arm_push_env.py)                  */
/* ------------------------------------------------------------------------- */
void oracle_set_muscle_layers(oracle_rod* r, const double* ratio_position, const double* strength)
{
    const int n = r->n;
    for (int m = 0; m < r->cfg.n_muscles; ++m)
        for (int k = 0; k < n; ++k) {
            for (int i = 0; i < 3; ++i) r->m_ratio[m][i][k] = ratio_position[(m * 3 + i) * n + k];
            r->m_strength[m][k] = strength[m * n + k];
        }
}

/* MuscleForce.apply_activation(activation): a scalar is broadcast over the elements */
void oracle_apply_activation(oracle_rod* r, int m, double activation)
{
    for (int k = 0; k < r->n; ++k) r->m_act[m][k] = activation;
}
/* ... and an array is taken element by element (arm_two_env.py:246-248, reach_env.py:176-179) */
void oracle_apply_activation_array(oracle_rod* r, int m, const double* activation)
{
    for (int k = 0; k < r->n; ++k) r->m_act[m][k] = activation[k];
}

/* test probe: ApplyMuscles on the current caches (sigma, kappa, radius, tangents, dilatations as last
 * evaluated or injected) -> the equivalent external force [3][n+1] and couple [3][n] */
void oracle_muscle_probe(oracle_rod* r, double* force, double* couple)
{
    const int n = r->n;
    for (int i = 0; i < 3; ++i) {
        for (int k = 0; k <= n; ++k) r->f_ext[i][k] = 0.0;
        for (int k = 0; k < n; ++k) r->t_ext[i][k] = 0.0;
    }
    apply_muscles(r);
    for (int i = 0; i < 3; ++i) {
        for (int k = 0; k <= n; ++k) { force[i * (n + 1) + k] = r->f_ext[i][k]; r->f_ext[i][k] = 0.0; }
        for (int k = 0; k < n; ++k) { couple[i * n + k] = r->t_ext[i][k]; r->t_ext[i][k] = 0.0; }
    }
}

/* get_state, arm_push_env.py:225-245: x-positions, x-velocities, then np.eye(2)[previous_action]
 * (discrete) or the previous action (continuous) */
static void get_state_push(const oracle_rod* r, float* obs)
{
    const int n = r->n;
    for (int k = 0; k <= n; ++k) { obs[k] = (float)r->x[0][k]; obs[n + 1 + k] = (float)r->v[0][k]; }
    if (r->cfg.arm_push_mode == 0) {
        const int a = (int)r->prev_action_push[0];
        obs[2 * n + 2] = a == 0 ? 1.0f : 0.0f;
        obs[2 * n + 3] = a == 0 ? 0.0f : 1.0f;
    } else {
        obs[2 * n + 2] = r->prev_action_push[0];
        obs[2 * n + 3] = r->prev_action_push[1];
    }
}

void oracle_observe_push(const oracle_rod* r, float* obs) { get_state_push(r, obs); }

/* ArmPushEnv.reset -> _build, arm_push_env.py:141-224 (the radii and the muscle layers were handed over
 * with oracle_set_radius_profile / oracle_set_muscle_layers, as _build computes them with NumPy) */
void oracle_reset_push(oracle_rod* r, float* obs)
{
    const double start[3] = { 0.0, 0.0, 0.0 }, direction[3] = { 1.0, 0.0, 0.0 }, normal[3] = { 0.0, 1.0, -0.0 };
    oracle_reset_straight(r, start, direction, normal);
    for (int m = 0; m < SOFTROD_MAX_MUSCLES; ++m) oracle_apply_activation(r, m, 0.0);   /* fresh muscle objects */
    get_state_push(r, obs);
}

/* ArmPushEnv.set_action, arm_push_env.py:247-274.  action: the index (discrete, 0 / 1, as a float) or
 * (location, activation) in float32 (continuous). */
static void push_set_action(oracle_rod* r, const float* action)
{
    const softrod_config* c = &r->cfg;
    const int n = r->n;
    if (c->arm_push_mode == 0) {
        if ((int)action[0] == 0) {
            r->sucker_index[0] = 0;
            oracle_apply_activation(r, 0, -0.0 * 1.0);
            oracle_apply_activation(r, 1, 0.0 * 1.0);
            oracle_apply_activation(r, 2, 0.5 * 1.0);
        } else {
            r->sucker_index[0] = -1;
            oracle_apply_activation(r, 0, 0.0);
            oracle_apply_activation(r, 1, 0.0);
            oracle_apply_activation(r, 2, 0.0);
        }
        r->prev_action_push[0] = action[0];
        r->prev_action_push[1] = 0.0f;
    } else {
        /* int(np.clip(location * self.n_elem, 0, self.n_elem - 1)): np.float32 * int stays float32 */
        float loc = action[0] * (float)n;
        if (loc < 0.0f) loc = 0.0f;
        if (loc > (float)(n - 1)) loc = (float)(n - 1);
        r->sucker_index[0] = (int)loc;
        oracle_apply_activation(r, 2, (double)action[1]);
        r->prev_action_push[0] = action[0];
        r->prev_action_push[1] = action[1];
    }
}

/* ArmPushEnv.step after the loop, arm_push_env.py:288-347; prev_cm = prev_cm_pos (:280) */
static void push_epilogue(oracle_rod* r, const double prev_cm[3], float* obs, double* reward,
                          uint8_t* terminated, uint8_t* truncated)
{
    const softrod_config* c = &r->cfg;
    const int n = r->n;
    double cm[3];
    center_of_mass(r, cm);
    /* _isnan_check over position, velocity, director, alpha, omega, cm_pos (:298-309); alpha = J^-1 tau e of the
     * last substep is NaN only where omega became NaN in that substep, so omega covers it */
    int invalid = 0;
    for (int i = 0; i < 3; ++i) {
        for (int k = 0; k <= n; ++k) if (isnan(r->x[i][k]) || isnan(r->v[i][k])) invalid = 1;
        for (int k = 0; k < n; ++k) {
            if (isnan(r->w[i][k])) invalid = 1;
            for (int j = 0; j < 3; ++j) if (isnan(r->Q[i][j][k])) invalid = 1;
        }
    }
    if (isnan(cm[0]) || isnan(cm[1])) invalid = 1;
    double survive = 0.0, forward = 0.0;
    *terminated = 0;
    *truncated = 0;
    if (invalid) { *terminated = 1; survive = -20.0; }
    else forward = sqrt(cm[0] * cm[0] + cm[1] * cm[1]) - sqrt(prev_cm[0] * prev_cm[0] + prev_cm[1] * prev_cm[1]);
    if (r->time > c->final_time) *truncated = 1;
    *reward = forward + survive;
    if (isnan(*reward)) { *terminated = 1; *reward = -20.0; }
    get_state_push(r, obs);
    int bad = 0;
    for (int i = 0; i < 2 * n + 4; ++i) bad = bad || isnan(obs[i]);
    if (bad) {
        *terminated = 1;
        *reward = -20.0;
        for (int i = 0; i < 2 * n + 4; ++i)        /* np.nan_to_num on float32 */
            obs[i] = isnan(obs[i]) ? 0.0f : (isinf(obs[i]) ? copysignf(3.4028234663852886e38f, obs[i]) : obs[i]);
    }
}

/* ArmPushEnv.step, arm_push_env.py:276-347 */
void oracle_env_step_push(oracle_rod* r, const float* action, float* obs, double* reward,
                          uint8_t* terminated, uint8_t* truncated)
{
    push_set_action(r, action);
    /* prev_cm_pos (:280).  run_substeps = 0 (fixture replay: the epilogue alone on an injected state): it is what
     * oracle_set("prev_com") put there */
    double prev_cm[3];
    if (substeps_to_run(r) > 0) { center_of_mass(r, prev_cm); r->prev_com[0] = prev_cm[0]; r->prev_com[1] = prev_cm[1]; }
    else { prev_cm[0] = r->prev_com[0]; prev_cm[1] = r->prev_com[1]; prev_cm[2] = 0.0; }
    for (int s = 0; s < substeps_to_run(r); ++s) position_verlet_step(r);
    push_epilogue(r, prev_cm, obs, reward, terminated, truncated);
}

size_t oracle_config_size(void) { return sizeof(softrod_config); }

#include "octoflat_oracle.inc.c"
