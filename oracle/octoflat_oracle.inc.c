/*
 * octoflat_oracle.inc.c — CPU restatement of OctoFlat-v0 (BASELINE config 5), included at
 * the end of softrod_oracle.c so that it shares the rod kernels above.  TEST INFRASTRUCTURE
 * ONLY, PARITY UNPINNED (see the header of softrod_oracle.c).
 *
 * What is restated, and from where:
 *   build_octopus                        gym_softrobot/envs/octopus/build.py:52-217
 *   FixedJoint2Rigid (forces, torques)   gym_softrobot/utils/custom_elastica/joint.py:20-225  (ON DISK)
 *   BodyBoundaryCondition                gym_softrobot/utils/custom_elastica/constraint.py:8-85 (ON DISK)
 *   intersection (arm-crossing count)    gym_softrobot/utils/intersection.py:12-77             (ON DISK)
 *   FlatEnv reset/get_state/step         gym_softrobot/envs/octopus/flat_env.py:171-408
 *   Cylinder rigid body + RigidBodyBase.update_accelerations, the Connections mixin
 *                                        pyelastica==1.0.0 (elastica/rigidbody/ modules,
 *                                        elastica/modules/connections.py) — recalled, not on disk
 * Operator order inside synchronize() = registration order of build_octopus: the eight
 * joints (:117-132), gravity per arm (:134-140), plane contact per arm (:193-200);
 * constrain_rates: head BC (:109-115), then the arm dampers (:145-151).
 */

#define OCTO_MAX_ARM 16

typedef struct rigid_head {
    double x[3], v[3], Q[3][3], w[3];
    double mass, J[3], invJ[3];
    double f_ext[3], t_ext[3];
    double fixed_z;
    double fixed_x[3], fixed_Q[3][3];      /* OneEndFixedBC on the rigid body (reach_env.py:126-130) */
} rigid_head;

typedef struct oracle_octo {
    softrod_config cfg;
    int n_arm;
    oracle_rod* arm[OCTO_MAX_ARM];
    rigid_head head;
    double angle[OCTO_MAX_ARM]; /* degrees, octopus/build.py:73-74 */
    double time;
    double target[2];
    float prev_action[3 * OCTO_MAX_ARM];
} oracle_octo;

/* Cylinder.__init__ (elastica/rigidbody/cylinder.py), call site octopus/build.py:103-105 */
static void cylinder_init(rigid_head* h, const double start[3], const double direction[3],
                          const double normal[3], double length, double radius, double density)
{
    const double volume = M_PI * radius * radius * length;
    h->mass = volume * density;
    const double area = M_PI * radius * radius;
    const double smoa1 = area * area / (4.0 * M_PI);
    const double smoa[3] = { smoa1, smoa1, 2.0 * smoa1 };
    for (int i = 0; i < 3; ++i) {
        h->J[i] = smoa[i] * density * length;
        h->invJ[i] = 1.0 / h->J[i];
        h->x[i] = start[i] + direction[i] * length / 2;
        h->v[i] = 0.0; h->w[i] = 0.0; h->f_ext[i] = 0.0; h->t_ext[i] = 0.0;
    }
    /* directors rows: normal, binormal = tangent x normal, tangent */
    for (int i = 0; i < 3; ++i) { h->Q[0][i] = normal[i]; h->Q[2][i] = direction[i]; }
    h->Q[1][0] = direction[1] * normal[2] - direction[2] * normal[1];
    h->Q[1][1] = direction[2] * normal[0] - direction[0] * normal[2];
    h->Q[1][2] = direction[0] * normal[1] - direction[1] * normal[0];
    h->fixed_z = h->x[2];
    for (int i = 0; i < 3; ++i) { h->fixed_x[i] = h->x[i]; for (int j = 0; j < 3; ++j) h->fixed_Q[i][j] = h->Q[i][j]; }
}

static void head_kinematic(rigid_head* h, double prefac, double eps)
{
    for (int i = 0; i < 3; ++i) h->x[i] += prefac * h->v[i];
    double v0 = h->w[0], v1 = h->w[1], v2 = h->w[2];      /* _get_rotation_matrix(prefac, omega) */
    double theta = sqrt(v0 * v0 + v1 * v1 + v2 * v2);
    v0 /= theta + eps; v1 /= theta + eps; v2 /= theta + eps;
    theta *= prefac;
    const double up = sin(theta), usq = 1.0 - cos(theta);
    double R[3][3];
    R[0][0] = 1.0 - usq * (v1 * v1 + v2 * v2);
    R[1][1] = 1.0 - usq * (v0 * v0 + v2 * v2);
    R[2][2] = 1.0 - usq * (v0 * v0 + v1 * v1);
    R[0][1] = up * v2 + usq * v0 * v1;  R[1][0] = -up * v2 + usq * v0 * v1;
    R[0][2] = -up * v1 + usq * v0 * v2; R[2][0] = up * v1 + usq * v0 * v2;
    R[1][2] = up * v0 + usq * v1 * v2;  R[2][1] = -up * v0 + usq * v1 * v2;
    double Qn[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            double s = R[i][0] * h->Q[0][j];
            s += R[i][1] * h->Q[1][j];
            s += R[i][2] * h->Q[2][j];
            Qn[i][j] = s;
        }
    memcpy(h->Q, Qn, sizeof(Qn));
}

/* BodyBoundaryCondition.compute_contrain_values, constraint.py:41-58 */
static void head_constrain_values(rigid_head* h)
{
    h->x[2] = h->fixed_z;
    h->Q[2][0] = 0.0; h->Q[2][1] = 0.0; h->Q[2][2] = 1.0;
    for (int i = 0; i < 2; ++i) {
        const double length = sqrt(h->Q[i][0] * h->Q[i][0] + h->Q[i][1] * h->Q[i][1]);
        for (int j = 0; j < 2; ++j) h->Q[i][j] /= length;
        h->Q[i][2] = 0.0;
    }
}

/* compute_constrain_rates, constraint.py:60-85 */
static void head_constrain_rates(rigid_head* h)
{
    h->v[2] = 0.0;
    h->w[0] = 0.0; h->w[1] = 0.0;
}

/* OneEndFixedBC(constrained_position_idx=(0,), constrained_director_idx=(0,)) registered on the rigid body AFTER its
 * BodyBoundaryCondition (reach_env.py:126-130): position[..., 0] and directors[..., 0] back to their finalize() values,
 * velocity[..., 0] = omega[..., 0] = 0 (pyelastica's OneEndFixedBC, recalled) */
static void head_fixed_values(rigid_head* h)
{
    for (int i = 0; i < 3; ++i) { h->x[i] = h->fixed_x[i]; for (int j = 0; j < 3; ++j) h->Q[i][j] = h->fixed_Q[i][j]; }
}
static void head_fixed_rates(rigid_head* h) { for (int i = 0; i < 3; ++i) { h->v[i] = 0.0; h->w[i] = 0.0; } }

/* FixedJoint2Rigid.apply_forces + apply_torques, joint.py:48-123,125-219, for the
 * connection (first_rod = head, idx -1 ; second_rod = arm, idx 0), octopus/build.py:117-132 */
static void joint_apply(oracle_octo* o, int a)
{
    const softrod_config* c = &o->cfg;
    rigid_head* h = &o->head;
    oracle_rod* r = o->arm[a];
    /* z_rotation(rod_one_binormal, angle): R_z(angle deg) applied to the head's d2 row */
    const double th = o->angle[a] / 180.0 * M_PI;
    const double ct = cos(th), st = sin(th);
    const double b[3] = { h->Q[1][0], h->Q[1][1], h->Q[1][2] };
    double dir[3] = { ct * b[0] + (-st) * b[1] + 0.0 * b[2], st * b[0] + ct * b[1] + 0 * b[2],
                      0.0 * b[0] + 0.0 * b[1] + 1.0 * b[2] };
    for (int i = 0; i < 3; ++i) dir[i] = -dir[i];
    double pos[3] = { h->x[0], h->x[1], 0.0 };         /* rigid_rod_pos[2] = 0.0 */
    for (int i = 0; i < 3; ++i) pos[i] += dir[i] * c->head_radius;
    double dvec[3], dist2 = 0.0;
    for (int i = 0; i < 3; ++i) { dvec[i] = r->x[i][0] - pos[i]; dist2 += dvec[i] * dvec[i]; }
    const double dist = sqrt(dist2);
    double nvec[3] = { 0.0, 0.0, 0.0 };
    if (!(dist <= 2.220446049250313e-16 * 1e4))
        for (int i = 0; i < 3; ++i) nvec[i] = dvec[i] / dist;
    double rel = 0.0;
    for (int i = 0; i < 3; ++i) rel += (r->v[i][0] - h->v[i]) * nvec[i];
    for (int i = 0; i < 3; ++i) {
        const double elastic = c->joint_k * dvec[i];
        const double damping = -c->joint_nu * (rel * nvec[i]);
        const double f = elastic + damping;
        h->f_ext[i] += f;
        r->f_ext[i][0] -= f;
    }
    /* torques */
    double link[3], tgt[3], force[3];
    for (int i = 0; i < 3; ++i) {
        link[i] = r->x[i][1] - r->x[i][0];
        tgt[i] = pos[i] + r->rest_len[0] * dir[i];
        force[i] = -c->joint_kt * (r->x[i][1] - tgt[i]);
    }
    const double tq[3] = { link[1] * force[2] - link[2] * force[1], link[2] * force[0] - link[0] * force[2],
                           link[0] * force[1] - link[1] * force[0] };
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) {
            h->t_ext[i] -= h->Q[i][j] * tq[j];
            r->t_ext[i][0] += r->Q[i][j][0] * tq[j];
        }
}

/* RigidBodyBase.update_accelerations + the dynamic step */
static void head_dynamic(rigid_head* h, double dt)
{
    double jw[3], lt[3];
    for (int i = 0; i < 3; ++i) jw[i] = h->J[i] * h->w[i];
    lt[0] = jw[1] * h->w[2] - jw[2] * h->w[1];
    lt[1] = jw[2] * h->w[0] - jw[0] * h->w[2];
    lt[2] = jw[0] * h->w[1] - jw[1] * h->w[0];
    for (int i = 0; i < 3; ++i) {
        const double a = h->f_ext[i] / h->mass;
        const double al = h->invJ[i] * (lt[i] + h->t_ext[i]);
        h->v[i] += dt * a;
        h->w[i] += dt * al;
    }
}

static void octo_substep(oracle_octo* o)
{
    const double dt = o->cfg.dt;
    const int na = o->n_arm;
    for (int a = 0; a < na; ++a) kinematic_step(o->arm[a], 0.5 * dt);
    head_kinematic(&o->head, 0.5 * dt, o->cfg.eps_rot_axis);
    if (o->cfg.time_two_half_adds) o->time += 0.5 * dt;
    head_constrain_values(&o->head);
    if (o->cfg.head_fixed) head_fixed_values(&o->head);
    for (int a = 0; a < na; ++a) { compute_internal_forces(o->arm[a]); compute_internal_torques(o->arm[a]); }
    /* synchronize: joints, gravity, contact (registration order); with the switch
     * contact_before_forcing the contact runs before the forcing group: joints, contact, gravity —
     * the plane's response then sees the joint load but not the weight */
    for (int a = 0; a < na; ++a) joint_apply(o, a);
    if (o->cfg.contact_before_forcing)
        for (int a = 0; a < na; ++a) plane_contact(o->arm[a]);
    for (int a = 0; a < na; ++a) apply_forcing(o->arm[a]);
    if (!o->cfg.contact_before_forcing)
        for (int a = 0; a < na; ++a) plane_contact(o->arm[a]);
    for (int a = 0; a < na; ++a) dynamic_step(o->arm[a], dt);
    head_dynamic(&o->head, dt);
    head_constrain_rates(&o->head);
    if (o->cfg.head_fixed) head_fixed_rates(&o->head);
    /* the arms' own constraints (ControllableFixConstraint of the muscle arms, arm_push_env.py:591-599; none in
     * FlatEnv) and dampers, in registration order (damp_before_constrain: dampen() is registered first there) */
    for (int a = 0; a < na; ++a) {
        if (o->cfg.damp_before_constrain) { dampen_rates(o->arm[a]); constrain_rates(o->arm[a]); }
        else { constrain_rates(o->arm[a]); dampen_rates(o->arm[a]); }
    }
    for (int a = 0; a < na; ++a) kinematic_step(o->arm[a], 0.5 * dt);
    head_kinematic(&o->head, 0.5 * dt, o->cfg.eps_rot_axis);
    o->time += o->cfg.time_two_half_adds ? 0.5 * dt : dt;
    head_constrain_values(&o->head);
    if (o->cfg.head_fixed) head_fixed_values(&o->head);
    for (int a = 0; a < na; ++a) {
        oracle_rod* r = o->arm[a];
        for (int i = 0; i < 3; ++i) {
            for (int k = 0; k <= r->n; ++k) r->f_ext[i][k] = 0.0;
            for (int k = 0; k < r->n; ++k) r->t_ext[i][k] = 0.0;
        }
    }
    for (int i = 0; i < 3; ++i) { o->head.f_ext[i] = 0.0; o->head.t_ext[i] = 0.0; }
}

/* ---- arm crossing: utils/intersection.py:29-77 ---- */
static int solve4(double A[4][4], double b[4], double x[4])
{   /* LU with partial pivoting (what np.linalg.solve / dgesv does) */
    int p[4] = { 0, 1, 2, 3 };
    for (int k = 0; k < 4; ++k) {
        int m = k;
        for (int i = k + 1; i < 4; ++i) if (fabs(A[i][k]) > fabs(A[m][k])) m = i;
        if (A[m][k] == 0.0) return -1; /* singular: numpy raises LinAlgError */
        if (m != k) {
            for (int j = 0; j < 4; ++j) { double t = A[k][j]; A[k][j] = A[m][j]; A[m][j] = t; }
            double t = b[k]; b[k] = b[m]; b[m] = t;
            int q = p[k]; p[k] = p[m]; p[m] = q;
        }
        for (int i = k + 1; i < 4; ++i) {
            const double f = A[i][k] / A[k][k];
            for (int j = k; j < 4; ++j) A[i][j] -= f * A[k][j];
            b[i] -= f * b[k];
        }
    }
    for (int i = 3; i >= 0; --i) {
        double s = b[i];
        for (int j = i + 1; j < 4; ++j) s -= A[i][j] * x[j];
        x[i] = s / A[i][i];
    }
    return 0;
}

static int count_intersections(const oracle_rod* r1, const oracle_rod* r2)
{
    const int n1 = r1->n, n2 = r2->n;
    int count = 0;
    for (int i = 0; i < n1; ++i)
        for (int j = 0; j < n2; ++j) {
            const double x1a = r1->x[0][i], x1b = r1->x[0][i + 1], y1a = r1->x[1][i], y1b = r1->x[1][i + 1];
            const double x2a = r2->x[0][j], x2b = r2->x[0][j + 1], y2a = r2->x[1][j], y2b = r2->x[1][j + 1];
            const int c1 = fmin(x1a, x1b) <= fmax(x2a, x2b);
            const int c2 = fmax(x1a, x1b) >= fmin(x2a, x2b);
            const int c3 = fmin(y1a, y1b) <= fmax(y2a, y2b);
            const int c4 = fmax(y1a, y1b) >= fmin(y2a, y2b);
            if (!(c1 && c2 && c3 && c4)) continue;
            double A[4][4] = { { x1b - x1a, 0.0, -1.0, 0.0 }, { 0.0, x2b - x2a, -1.0, 0.0 },
                               { y1b - y1a, 0.0, 0.0, -1.0 }, { 0.0, y2b - y2a, 0.0, -1.0 } };
            double b[4] = { -x1a, -x2a, -y1a, -y2a }, T[4];
            if (solve4(A, b, T) != 0) continue;
            if (T[0] >= 0 && T[1] >= 0 && T[0] <= 1 && T[1] <= 1) ++count;
        }
    return count;
}

/* FlatEnv.get_state (centralized), flat_env.py:231-286: individual (n_arm, 56), shared (13,) */
static void octo_get_state(const oracle_octo* o, float* individual, float* shared)
{
    const int na = o->n_arm, n = o->arm[0]->n, nk = o->cfg.n_knots;
    const int width = (n - 1) + (n + 1) * 4 + nk;
    const double cx = o->head.x[0], cy = o->head.x[1];
    for (int a = 0; a < na; ++a) {
        const oracle_rod* r = o->arm[a];
        float* row = individual + (size_t)a * width;
        int q = 0;
        for (int k = 0; k < n - 1; ++k) row[q++] = (float)r->kappa[0][k];
        for (int k = 0; k <= n; ++k) row[q++] = (float)(r->x[0][k] - cx);
        for (int k = 0; k <= n; ++k) row[q++] = (float)(r->x[1][k] - cy);
        for (int k = 0; k <= n; ++k) row[q++] = (float)r->v[0][k];
        for (int k = 0; k <= n; ++k) row[q++] = (float)r->v[1][k];
        for (int i = 0; i < nk; ++i) row[q++] = o->prev_action[a * nk + i];
    }
    shared[0] = (float)(o->target[0] - o->head.x[0]);
    shared[1] = (float)(o->target[1] - o->head.x[1]);
    shared[2] = (float)o->head.v[0];
    shared[3] = (float)o->head.v[1];
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) shared[4 + 3 * i + j] = (float)o->head.Q[i][j];
}

/* ---- exported API ---- */
oracle_octo* oracle_octo_create(const softrod_config* cfg)
{
    if (!cfg || cfg->struct_size != sizeof(softrod_config)) return NULL;
    const int mocto = cfg->env_kind == SOFTROD_ENV_CRAWL || cfg->env_kind == SOFTROD_ENV_ARM_TWO || cfg->env_kind == SOFTROD_ENV_REACH;
    if (cfg->n_arm < 1 || cfg->n_arm > OCTO_MAX_ARM || cfg->n_knots < 1 ||
        (!mocto && cfg->n_arm * cfg->n_knots > 3 * OCTO_MAX_ARM)) return NULL;
    oracle_octo* o = (oracle_octo*)calloc(1, sizeof(oracle_octo));
    if (!o) return NULL;
    o->cfg = *cfg;
    o->n_arm = cfg->n_arm;
    softrod_config arm_cfg = *cfg;
    arm_cfg.features &= ~(unsigned)SOFTROD_FEAT_OCTO_HEAD;
    for (int a = 0; a < o->n_arm; ++a) {
        o->arm[a] = oracle_create(&arm_cfg);
        if (!o->arm[a]) return NULL;
    }
    return o;
}

void oracle_octo_destroy(oracle_octo* o)
{
    if (!o) return;
    for (int a = 0; a < o->n_arm; ++a) oracle_destroy(o->arm[a]);
    free(o);
}

/* FlatEnv.reset, flat_env.py:171-229.  arm_pos / arm_dir [n_arm][3] are the
 * scipy Rotation.from_euler("z", angle, degrees=True).apply(...) results of
 * octopus/build.py:76-80, computed by the caller exactly as the reference does;
 * target is `(2 - 0.5) * np_random.random(2) + 0.5` (:221). */
void oracle_octo_reset(oracle_octo* o, const double* arm_pos, const double* arm_dir, const double target[2],
                       float* individual, float* shared)
{
    const softrod_config* c = &o->cfg;
    const double normal[3] = { 0.0, 0.0, 1.0 };
    const double rotation_angle = 360 / (double)o->n_arm;
    for (int a = 0; a < o->n_arm; ++a) {
        o->angle[a] = rotation_angle * a;
        oracle_reset_straight(o->arm[a], arm_pos + 3 * a, arm_dir + 3 * a, normal);
    }
    const double start[3] = { 0.0, 0.0, -c->base_radius };
    const double direction[3] = { 0.0, 0.0, 1.0 }, hn[3] = { 0.0, 1.0, 0.0 };
    cylinder_init(&o->head, start, direction, hn, c->base_radius * 2, c->head_radius, c->head_density);
    /* finalize(): constraints are applied once at t = 0 */
    head_constrain_values(&o->head);
    head_constrain_rates(&o->head);
    o->time = 0.0;
    o->target[0] = target[0]; o->target[1] = target[1];
    octo_get_state(o, individual, shared);
}

/* FlatEnv.step, flat_env.py:315-408.  rest_kappa0 [n_arm][n_elem-1] is the interp1d output
 * of set_action (:288-311), computed by the caller with scipy exactly as the reference. */
/* FlatEnv.step after the substep loop, flat_env.py:330-408; `before` = xposbefore (:321) */
static void octo_epilogue(oracle_octo* o, const double before[2], float* individual, float* shared,
                          double* reward, uint8_t* terminated, uint8_t* truncated)
{
    const softrod_config* c = &o->cfg;
    const int na = o->n_arm, n = o->arm[0]->n;
    int invalid = 0;
    for (int a = 0; a < na; ++a)
        for (int i = 0; i < 3; ++i)
            for (int k = 0; k <= n; ++k)
                if (isnan(o->arm[a]->x[i][k]) || isnan(o->arm[a]->v[i][k])) invalid = 1;
    int crossing = 0;
    for (int i = 0; i < na - 1; ++i) /* pairs (i-1, i) with i-1 = -1 wrapping to the last arm */
        crossing += count_intersections(o->arm[(i - 1 + na) % na], o->arm[i]);
    const double tx = o->target[0] - o->head.x[0], ty = o->target[1] - o->head.x[1];
    const double dist = sqrt(tx * tx + ty * ty);
    double survive = 0.0, forward = 0.0;
    *terminated = 0;
    if (invalid) { *terminated = 1; survive = -50.0; }
    else {
        survive = -0.02 * crossing;
        const double bx = o->target[0] - before[0], by = o->target[1] - before[1];
        forward = (dist - sqrt(bx * bx + by * by)) / ((double)c->n_substeps * c->dt);
        if (dist < 0.1) { survive = 100.0; *terminated = 1; }
    }
    *truncated = (o->time > c->final_time) ? 1 : 0;
    double rew = forward - 0.0 + survive - 0.0;
    if (*terminated) rew -= dist - 0.1;
    *reward = rew;
    octo_get_state(o, individual, shared);
}

void oracle_octo_env_step(oracle_octo* o, const float* action, const double* rest_kappa0,
                          float* individual, float* shared, double* reward, uint8_t* terminated,
                          uint8_t* truncated)
{
    const softrod_config* c = &o->cfg;
    const int na = o->n_arm, n = o->arm[0]->n;
    for (int i = 0; i < na * c->n_knots; ++i) o->prev_action[i] = action[i];
    for (int a = 0; a < na; ++a)
        for (int k = 0; k < n - 1; ++k) o->arm[a]->rest_kappa[0][k] = rest_kappa0[a * (n - 1) + k];
    const double before[2] = { o->head.x[0], o->head.x[1] };
    for (int s = 0; s < c->n_substeps; ++s) octo_substep(o);
    octo_epilogue(o, before, individual, shared, reward, terminated, truncated);
}

/* batched driver (OpenMP over envs when built with it): full-batch parity at BASELINE configs[4]'s per-GPU share.
 * obs rows: the n_arm "individual" rows, then "shared" (13). */
void oracle_octo_env_step_batch(oracle_octo** envs, int n_envs, const float* actions, const double* rest_kappa0,
                                float* obs, double* reward, uint8_t* terminated, uint8_t* truncated)
{
#pragma omp parallel for schedule(dynamic, 4)
    for (int e = 0; e < n_envs; ++e) {
        oracle_octo* o = envs[e];
        const int na = o->n_arm, n = o->arm[0]->n, nk = o->cfg.n_knots;
        const int wi = (n - 1) + 4 * (n + 1) + nk, w = na * wi + 13;
        oracle_octo_env_step(o, actions + (size_t)na * nk * e, rest_kappa0 + (size_t)na * (n - 1) * e,
                             obs + (size_t)w * e, obs + (size_t)w * e + (size_t)na * wi, reward + e, terminated + e,
                             truncated + e);
    }
}

/* The epilogue alone, on whatever state the arms / head / clock hold now, with the
 * pre-loop head position given: replays the fixtures recorded from the reference's own
 * FlatEnv.step (tests/golden/ref_octoflat.npz). */
void oracle_octo_epilogue_probe(oracle_octo* o, const float* action, const double before[2],
                                float* individual, float* shared, double* reward, uint8_t* terminated,
                                uint8_t* truncated)
{
    for (int i = 0; i < o->n_arm * o->cfg.n_knots; ++i) o->prev_action[i] = action[i];
    octo_epilogue(o, before, individual, shared, reward, terminated, truncated);
}
void oracle_octo_set_target(oracle_octo* o, const double t[2]) { o->target[0] = t[0]; o->target[1] = t[1]; }

void oracle_octo_substeps(oracle_octo* o, int n) { for (int s = 0; s < n; ++s) octo_substep(o); }
double oracle_octo_time(const oracle_octo* o) { return o->time; }
oracle_rod* oracle_octo_arm(oracle_octo* o, int a) { return (a >= 0 && a < o->n_arm) ? o->arm[a] : NULL; }
int oracle_octo_crossings(oracle_octo* o)
{
    int c = 0;
    for (int i = 0; i < o->n_arm - 1; ++i)
        c += count_intersections(o->arm[(i - 1 + o->n_arm) % o->n_arm], o->arm[i]);
    return c;
}
/* test probes: ONE joint, or the head constraint, evaluated on the current state
 * (tests/golden/octo_operator_vectors.npz holds what the reference's own joint.py /
 * constraint.py give for the same inputs).  out: head force, head torque, arm node-0 force,
 * arm element-0 torque. */
void oracle_octo_joint_probe(oracle_octo* o, int a, double angle_deg, double* out)
{
    oracle_rod* r = o->arm[a];
    for (int i = 0; i < 3; ++i) {
        o->head.f_ext[i] = 0.0; o->head.t_ext[i] = 0.0; r->f_ext[i][0] = 0.0; r->t_ext[i][0] = 0.0;
    }
    const double keep = o->angle[a];
    o->angle[a] = angle_deg;
    joint_apply(o, a);
    o->angle[a] = keep;
    for (int i = 0; i < 3; ++i) {
        out[i] = o->head.f_ext[i]; out[3 + i] = o->head.t_ext[i];
        out[6 + i] = r->f_ext[i][0]; out[9 + i] = r->t_ext[i][0];
        o->head.f_ext[i] = 0.0; o->head.t_ext[i] = 0.0; r->f_ext[i][0] = 0.0; r->t_ext[i][0] = 0.0;
    }
}
void oracle_octo_head_constrain_probe(oracle_octo* o)
{
    head_constrain_values(&o->head);
    head_constrain_rates(&o->head);
}
/* state injection for the windowed parity tests: x[3], v[3], Q[9], w[3] ; time */
void oracle_octo_set_head(oracle_octo* o, const double* in)
{
    for (int i = 0; i < 3; ++i) { o->head.x[i] = in[i]; o->head.v[i] = in[3 + i]; o->head.w[i] = in[15 + i]; }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) o->head.Q[i][j] = in[6 + 3 * i + j];
}
void oracle_octo_set_time(oracle_octo* o, double t) { o->time = t; }
/* head state: x[3], v[3], Q[9], w[3], mass, J[3] -> 22 doubles */
void oracle_octo_head(const oracle_octo* o, double* out)
{
    for (int i = 0; i < 3; ++i) { out[i] = o->head.x[i]; out[3 + i] = o->head.v[i]; out[15 + i] = o->head.w[i]; }
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) out[6 + 3 * i + j] = o->head.Q[i][j];
    out[18] = o->head.mass;
    for (int i = 0; i < 3; ++i) out[19 + i] = o->head.J[i];
}


/* ------------------------------------------------------------------------- */
/* ArmPullWeightEnv (octopus/arm_push_env.py:516-618, OctoArmPullWeight-v0):    */
/* ArmPushEnv's arm and step() with a rigid Cylinder joined to node 0.          */
/* PARITY UNPINNED (the COOMM muscle law; softrod_oracle.c apply_muscles).      */
/* ------------------------------------------------------------------------- */
/* reset -> _build (:520-618): the arm as ArmPushEnv's; Cylinder(start, e_z, e_y, 2 radius_base, 0.015, 700) (:552-567);
 * BodyBoundaryCondition on it (:569-575); FixedJoint2Rigid(head idx -1, arm idx 0, angle 0) (:577-589) */
void oracle_pull_reset(oracle_octo* o, float* obs)
{
    const softrod_config* c = &o->cfg;
    const double start[3] = { 0.0, 0.0, 0.0 }, direction[3] = { 1.0, 0.0, 0.0 }, normal[3] = { 0.0, 1.0, -0.0 };
    oracle_rod* r = o->arm[0];
    oracle_reset_straight(r, start, direction, normal);
    for (int m = 0; m < SOFTROD_MAX_MUSCLES; ++m) oracle_apply_activation(r, m, 0.0);
    o->angle[0] = c->joint_angle0;
    const double hd[3] = { 0.0, 0.0, 1.0 }, hn[3] = { 0.0, 1.0, 0.0 };
    const double hs[3] = { c->head_center[0] - hd[0] * c->head_length / 2, c->head_center[1] - hd[1] * c->head_length / 2,
                           c->head_center[2] - hd[2] * c->head_length / 2 };
    cylinder_init(&o->head, hs, hd, hn, c->head_length, c->head_radius, c->head_density);
    head_constrain_values(&o->head);
    head_constrain_rates(&o->head);
    o->time = 0.0;
    r->time = 0.0;
    get_state_push(r, obs);
}

void oracle_pull_observe(const oracle_octo* o, float* obs) { get_state_push(o->arm[0], obs); }

/* ArmPullWeightEnv.step = ArmPushEnv.step (:276-347) over the two-body simulator */
void oracle_env_step_pull(oracle_octo* o, const float* action, float* obs, double* reward,
                          uint8_t* terminated, uint8_t* truncated)
{
    oracle_rod* r = o->arm[0];
    push_set_action(r, action);
    double prev_cm[3];
    center_of_mass(r, prev_cm);
    for (int s = 0; s < o->cfg.n_substeps; ++s) octo_substep(o);
    r->time = o->time;
    push_epilogue(r, prev_cm, obs, reward, terminated, truncated);
}


/* ------------------------------------------------------------------------- */
/* The muscle octopus (octopus/build_muscle_octopus.py): build_octopus_muscles  */
/* (:66-179) / build_two_arms (:182-291) under CrawlEnv / ArmTwoEnv / ReachEnv. */
/* The body only: set_action / get_state / step's bookkeeping are NumPy in      */
/* tests/oracle_backend.py, line by line after the env files.  PARITY UNPINNED  */
/* (the COOMM muscle law; softrod_oracle.c apply_muscles).                      */
/* ------------------------------------------------------------------------- */
/* arm_pos / arm_dir [n_arm][3]: Rot.from_euler("z", angle, degrees=True).apply(...) of :87-93, by the caller; normal e_z;
 * Cylinder((0, 0, -2 r0), e_z, e_y, 2 r0, head_radius, head_density) (:108-114) as head_center / head_length;
 * FixedJoint2Rigid(angle=angles[arm_i]) (:127-141): joint_angle0 + a * joint_angle_step; fresh muscle objects and
 * SuckerControllers (crawl_env.py:146-155: index, reduction_ratio 1.0, turned on after finalize) */
void oracle_mocto_reset(oracle_octo* o, const double* arm_pos, const double* arm_dir)
{
    const softrod_config* c = &o->cfg;
    const double normal[3] = { 0.0, 0.0, 1.0 };
    for (int a = 0; a < o->n_arm; ++a) {
        oracle_rod* r = o->arm[a];
        o->angle[a] = c->joint_angle0 + c->joint_angle_step * (double)a;
        oracle_reset_straight(r, arm_pos + 3 * a, arm_dir + 3 * a, normal);
        for (int m = 0; m < SOFTROD_MAX_MUSCLES; ++m) oracle_apply_activation(r, m, 0.0);
        for (int j = 0; j < SOFTROD_MAX_SUCKERS; ++j) {
            r->sucker_index[j] = c->sucker_index[j];
            r->sucker_ratio[j] = (j < c->n_suckers) ? c->sucker_reduction_ratio : 0.0;
        }
        r->time = 0.0;
    }
    const double hd[3] = { 0.0, 0.0, 1.0 }, hn[3] = { 0.0, 1.0, 0.0 };
    const double hs[3] = { c->head_center[0] - hd[0] * c->head_length / 2, c->head_center[1] - hd[1] * c->head_length / 2,
                           c->head_center[2] - hd[2] * c->head_length / 2 };
    cylinder_init(&o->head, hs, hd, hn, c->head_length, c->head_radius, c->head_density);
    head_constrain_values(&o->head);
    head_constrain_rates(&o->head);
    if (c->head_fixed) { head_fixed_values(&o->head); head_fixed_rates(&o->head); }
    o->time = 0.0;
}

/* one env.step's stepping (crawl_env.py:251-253): n_substeps of the whole body */
void oracle_mocto_step(oracle_octo* o)
{
    for (int s = 0; s < o->cfg.n_substeps; ++s) octo_substep(o);
    for (int a = 0; a < o->n_arm; ++a) o->arm[a]->time = o->time;
}
