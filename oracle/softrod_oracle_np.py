"""NumPy restatement (fp64) of the hot path — second, independent oracle: the SoftPendulum-v0
substep (round 1), and since round 4 the two pieces the C oracle held as its ONLY transcription:
RodPlaneContactWithAnisotropicFriction (BASELINE configs[2], OctoArmSingle-v0) and the rigid
octopus head — Cylinder, FixedJoint2Rigid, BodyBoundaryCondition (configs[4], OctoFlat-v0).
Those two were written from PyElastica's array-style kernels as recalled (elastica/
_contact_functions.py, contact_utils.py, rigidbody/cylinder.py, rigidbody/rigid_body.py; the
friction model of Gazzola et al. 2018 §4 / Methods) and from the reference's OWN files for the
joint and the head constraint (gym_softrobot/utils/custom_elastica/joint.py:20-225,
constraint.py:8-85; assembly octopus/build.py:52-217) — NOT from softrod_oracle.c, which
tests/test_oracle_np_contact_head.py holds against this file to 1e-12.

TEST INFRASTRUCTURE ONLY: imported by tests/ and tools/make_golden.py, never by
gym_softrobot_amd/.  PARITY UNPINNED for the PyElastica arithmetic (see the header
of softrod_oracle.c for why); this file exists so that the C oracle is checked
against a second transcription written in PyElastica's own array style
((3, n) blocks, whole-array kernels) rather than per-element loops.

Anchors in the reference:
  assembly   gym_softrobot/envs/soft_pendulum/build.py:29-115
  hot loop   gym_softrobot/envs/soft_pendulum/soft_pendulum.py:183-184
  epilogue   gym_softrobot/envs/soft_pendulum/soft_pendulum.py:149-161,196-251
PyElastica modules restated (pyelastica==1.0.0, uv.lock:845-846; not on disk):
  elastica/rod/factory_function.py, rod/cosserat_rod.py, _rotations.py,
  _calculus.py, timestepper/symplectic_steppers.py, external_forces.py,
  dissipation.py.
"""
from __future__ import annotations

import numpy as np


def _difference(a):
    """two-point difference kernel: (3, m) -> (3, m+1)."""
    out = np.empty((3, a.shape[1] + 1))
    out[:, 0] = a[:, 0]
    out[:, 1:-1] = a[:, 1:] - a[:, :-1]
    out[:, -1] = -a[:, -1]
    return out


def _trapezoidal(a):
    out = np.empty((3, a.shape[1] + 1))
    out[:, 0] = 0.5 * a[:, 0]
    out[:, 1:-1] = 0.5 * (a[:, 1:] + a[:, :-1])
    out[:, -1] = 0.5 * a[:, -1]
    return out


def _matvec(A, v):  # (3,3,n),(3,n)->(3,n)
    return np.einsum("ijk,jk->ik", A, v)


def _cross(a, b):
    return np.cross(a, b, axis=0)


class NumpyRod:
    """One rod; cfg is any object with the attribute names of softrod_config."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.n = int(cfg.n_elem)

    # -- CosseratRod.straight_rod (build.py:54-61) ---------------------------------
    def reset_straight(self, start, direction, normal):
        c, n = self.cfg, self.n
        start = np.asarray(start, float)
        direction = np.asarray(direction, float)
        normal = np.asarray(normal, float)
        end = start + direction * c.base_length
        self.x = np.stack([np.linspace(start[i], end[i], n + 1) for i in range(3)])
        diff = self.x[:, 1:] - self.x[:, :-1]
        self.rest_len = np.sqrt(np.einsum("ik,ik->k", diff, diff))
        tang = diff / self.rest_len
        normal = normal / np.linalg.norm(normal)
        ncol = np.repeat(normal[:, None], n, axis=1)
        self.Q = np.zeros((3, 3, n))
        self.Q[0] = ncol
        self.Q[1] = _cross(tang, ncol)
        self.Q[2] = tang
        radius = np.full(n, c.base_radius)
        A0 = np.pi * radius * radius
        I1 = A0 * A0 / (4.0 * np.pi)
        I0 = np.array([I1, I1, 2.0 * I1])  # (3, n)
        self.J = I0 * (c.density * self.rest_len)  # diagonal entries
        self.invJ = 1.0 / self.J
        G = c.shear_modulus
        self.shear = np.array([c.alpha_c * G * A0, c.alpha_c * G * A0, c.youngs_modulus * A0])
        be = np.array([c.youngs_modulus * I0[0], c.youngs_modulus * I0[1], G * I0[2]])
        rl = self.rest_len
        self.bend = (be[:, 1:] * rl[1:] + be[:, :-1] * rl[:-1]) / (rl[1:] + rl[:-1])
        self.volume = np.pi * radius**2 * rl
        self.mass = np.zeros(n + 1)
        self.mass[:-1] += 0.5 * c.density * self.volume
        self.mass[1:] += 0.5 * c.density * self.volume
        self.rest_vor = 0.5 * (rl[1:] + rl[:-1])
        self.v = np.zeros((3, n + 1))
        self.w = np.zeros((3, n))
        self.rest_sigma = np.zeros((3, n))
        self.rest_kappa = np.zeros((3, n - 1))
        self.f_ext = np.zeros((3, n + 1))
        self.t_ext = np.zeros((3, n))
        # AnalyticalLinearDamper.__init__ (build.py:108-113)
        self.damp_t = np.exp(-c.damping_constant * c.dt)
        me = 0.5 * (self.mass[1:] + self.mass[:-1])
        me[0] += 0.5 * self.mass[0]
        me[-1] += 0.5 * self.mass[-1]
        self.damp_r = np.exp(-c.damping_constant * c.dt * me * self.invJ)
        if getattr(c, "damper_protocol", 0) == 1:      # uniform protocol: the same exp(-nu dt) on every rate
            self.damp_r = np.full_like(self.damp_r, self.damp_t)
        self.fixed_pos = self.x[:, 0].copy()
        self.fixed_dir = self.Q[:, :, 0].copy()
        self.time = np.float64(0.0)
        self.point_force = 0.0
        self.prev_action = np.float32(0.0)
        self._shear_stress()
        self._kappa()

    def reset_pendulum(self, theta):  # build.py:46-52
        direction = np.array([1.0 * np.cos(theta), 1.0 * np.sin(theta), 0.0])
        normal = np.array([1.0 * np.sin(theta), -1.0 * np.cos(theta), 0.0])
        self.reset_straight(np.zeros(3), direction, normal)

    # -- cosserat_rod.py kernels ---------------------------------------------------
    def _shear_stress(self):
        c = self.cfg
        d = self.x[:, 1:] - self.x[:, :-1]
        self.len = np.sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]) + c.eps_length
        self.tang = d / self.len
        self.radius = np.sqrt(self.volume / self.len / np.pi)      # _compute_geometry_from_state
        self.dil = self.len / self.rest_len
        self.vdil = 0.5 * (self.len[1:] + self.len[:-1]) / self.rest_vor
        z = np.array([0.0, 0.0, 1.0]).reshape(3, 1)
        self.sigma = self.dil * _matvec(self.Q, self.tang) - z
        self.n_int = self.shear * (self.sigma - self.rest_sigma)

    def _kappa(self):
        c = self.cfg
        Q = self.Q
        R = np.einsum("imk,jmk->ijk", Q[:, :, 1:], Q[:, :, :-1])  # Q_{k+1} Q_k^T
        vec = np.array([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
        trace = R[0, 0] + R[1, 1] + R[2, 2]
        theta = np.arccos(0.5 * trace - 0.5 - c.acos_shift)
        self.kappa = vec * (-0.5 * theta / np.sin(theta + c.eps_sin)) / self.rest_vor

    def _forces_and_torques(self):
        self._shear_stress()
        cs = np.einsum("jik,jk->ik", self.Q, self.n_int) / self.dil
        self.f_int = _difference(cs)
        self._kappa()
        self.m_int = self.bend * (self.kappa - self.rest_kappa)
        x, v = self.x, self.v
        rv = np.einsum("ik,ik->k", x, v)
        rp1v = np.einsum("ik,ik->k", x[:, 1:], v[:, :-1])
        rvp1 = np.einsum("ik,ik->k", x[:, :-1], v[:, 1:])
        self.dil_rate = (rv[:-1] + rv[1:] - rvp1 - rp1v) / self.len / self.rest_len
        e3 = 1.0 / self.vdil**3
        c2d = _difference(self.m_int * e3)
        c3d = _trapezoidal(_cross(self.kappa, self.m_int) * self.rest_vor * e3)
        ssc = _cross(_matvec(self.Q, self.tang), self.n_int) * self.rest_len
        jwe = self.J * self.w / self.dil
        lt = _cross(jwe, self.w)
        ud = jwe * self.dil_rate / self.dil
        self.t_int = c2d + c3d + ssc + lt + ud

    # -- symplectic_steppers.py ------------------------------------------------------
    def _kinematic(self, prefac):
        c = self.cfg
        self.x = self.x + prefac * self.v
        ax = self.w                    # _get_rotation_matrix(prefac, omega): unscaled axis norm,
        theta = np.sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2])
        u = ax / (theta + c.eps_rot_axis)
        theta = theta * prefac         # then the angle is scaled
        up, usq = np.sin(theta), 1.0 - np.cos(theta)
        R = np.empty((3, 3, self.n))
        R[0, 0] = 1.0 - usq * (u[1] * u[1] + u[2] * u[2])
        R[1, 1] = 1.0 - usq * (u[0] * u[0] + u[2] * u[2])
        R[2, 2] = 1.0 - usq * (u[0] * u[0] + u[1] * u[1])
        R[0, 1] = up * u[2] + usq * u[0] * u[1]
        R[1, 0] = -up * u[2] + usq * u[0] * u[1]
        R[0, 2] = -up * u[1] + usq * u[0] * u[2]
        R[2, 0] = up * u[1] + usq * u[0] * u[2]
        R[1, 2] = up * u[0] + usq * u[1] * u[2]
        R[2, 1] = -up * u[0] + usq * u[1] * u[2]
        self.Q = np.einsum("imk,mjk->ijk", R, self.Q)

    def _constrain_values(self):
        f = self.cfg.features
        if f & 4:  # PENDULUM_BC, build.py:71-74
            self.x[1:, 0] = self.fixed_pos[1:]
            self.Q[0, :, 0] = self.fixed_dir[0, :]
            self.Q[2, :, 0] = self.fixed_dir[2, :]
        if f & 16:  # FIXED_BC
            self.x[:, 0] = self.fixed_pos
            self.Q[:, :, 0] = self.fixed_dir

    def _constrain_rates(self):
        f = self.cfg.features
        if f & 4:  # build.py:76-79
            self.v[1:, 0] = 0
            self.w[0, 0] = 0
            self.w[2, 0] = 0
        if f & 16:
            self.v[:, 0] = 0
            self.w[:, 0] = 0

    # -- one PositionVerlet substep in phases (a system of several bodies interleaves them) --------
    def forcing(self):
        """add_forcing_to(...) operators in registration order (build.py:88-105, octopus/build.py:134-141)."""
        c = self.cfg
        g = np.asarray(list(c.gravity), float)
        if c.features & 1:
            self.f_ext += g[:, None] * self.mass[None, :]
        if c.features & 2:
            self.f_ext[0, 0] = self.point_force  # assigns: build.py:101
        if c.features & 32:
            self.f_ext[:, -1] += np.asarray(list(c.tip_force), float)

    def contact(self):
        """detect_contact_between(rod, plane).using(RodPlaneContactWithAnisotropicFriction, ...)
        (octopus/build.py:193-200,274-283)."""
        c = self.cfg
        if c.features & 256:
            rod_plane_contact_with_anisotropic_friction(
                self, np.asarray(list(c.plane_origin), float), np.asarray(list(c.plane_normal), float),
                c.surface_tol, c.slip_velocity_tol, c.contact_k, c.contact_nu,
                np.asarray(list(c.kinetic_mu), float), np.asarray(list(c.static_mu), float))

    def dynamic(self, dt):
        acc = (self.f_int + self.f_ext) / self.mass
        alpha = (self.invJ * (self.t_int + self.t_ext)) * self.dil
        self.v = self.v + dt * acc
        self.w = self.w + dt * alpha

    def dampen(self):
        if self.cfg.features & 8:
            self.v = self.v * self.damp_t
            self.w = self.w * np.power(self.damp_r, self.dil)

    def rates_operators(self):
        """constrain_rates and dampen_rates in the order the switch says (0: constrain first)."""
        if self.cfg.damp_before_constrain:
            self.dampen()
            self._constrain_rates()
        else:
            self._constrain_rates()
            self.dampen()

    def zero_external(self):
        self.f_ext[:] = 0.0
        self.t_ext[:] = 0.0

    def substep(self):
        c = self.cfg
        dt = c.dt
        self._kinematic(0.5 * dt)
        if c.time_two_half_adds:
            self.time = self.time + 0.5 * dt
        self._constrain_values()
        self._forces_and_torques()
        if c.contact_before_forcing:
            self.contact()
            self.forcing()
        else:
            self.forcing()
            self.contact()
        self.dynamic(dt)
        self.rates_operators()
        self._kinematic(0.5 * dt)
        self.time = self.time + (0.5 * dt if c.time_two_half_adds else dt)
        self._constrain_values()
        self.zero_external()

    # -- env epilogue (soft_pendulum.py) ---------------------------------------------
    def theta(self):
        tm = np.mean(self.tang, axis=1)
        th = np.arctan(tm[0] / tm[1])
        return ((th + np.pi) % (2 * np.pi)) - np.pi

    def get_state(self):  # :149-161
        return np.hstack(
            [self.x[0, 0], self.v[0, 0], np.array([self.prev_action]), self.theta()]
        ).astype(np.float32)

    def env_step(self, action):  # :176-251
        a32 = np.float32(action)
        self.prev_action = a32
        self.point_force = float(a32)
        for _ in range(int(self.cfg.n_substeps)):
            self.substep()
        invalid = bool(np.isnan(np.concatenate([self.x, self.v])).any())
        terminated, survive, forward = False, 0.0, 0.0
        if invalid:
            terminated, survive = True, -50.0
        else:
            th = self.theta()
            forward = np.abs(self.x[0, 0]) * 10 + th**2
        truncated = bool(self.time > self.cfg.final_time)
        return self.get_state(), forward - 0.0 + survive, terminated, truncated


# ===============================================================================================
# RodPlaneContactWithAnisotropicFriction — elastica/_contact_functions.py
# (_calculate_contact_forces_rod_plane, ..._with_anisotropic_friction) and contact_utils.py, in
# PyElastica's whole-array style.  Friction model: Gazzola et al. 2018, Methods "Surface friction":
# kinetic Coulomb friction along the in-plane axial direction (forward / backward coefficients by the
# sign of the axial velocity) and along the rolling direction (sideways coefficient; slip = rolling
# velocity + spin of the contact point), blended out by the slip function below the slip-velocity
# threshold, where static friction takes over (axial: up to mu_s N against the pushing force;
# rolling: the no-slip force (r F_roll - 2 T_axial) / (3 r)), with the rolling friction's torque
# r_contact x F about the axis.
# ===============================================================================================
def _node_to_element_force(f):          # node_to_element_mass_or_force
    out = 0.5 * (f[:, :-1] + f[:, 1:])
    out[:, 0] += 0.5 * f[:, 0]
    out[:, -1] += 0.5 * f[:, -1]
    return out


def _node_to_element_position(x):
    return 0.5 * (x[:, 1:] + x[:, :-1])


def _node_to_element_velocity(mass, v):
    return (mass[1:] * v[:, 1:] + mass[:-1] * v[:, :-1]) / (mass[1:] + mass[:-1])


def _elements_to_nodes_inplace(e, nodes):
    nodes[:, :-1] += 0.5 * e
    nodes[:, 1:] += 0.5 * e


def _norm(a):
    return np.sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2])


def _dot(a, b):
    return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]


def find_slipping_elements(velocity_slip, velocity_threshold):
    """1 below the threshold (static regime), falling linearly to 0 at twice the threshold."""
    mag = _norm(velocity_slip)
    slip = np.ones(velocity_slip.shape[1])
    pts = np.fabs(mag) > velocity_threshold
    slip[pts] = np.fabs(1.0 - np.minimum(1.0, mag[pts] / velocity_threshold - 1.0))
    return slip


def rod_plane_contact(rod, origin, normal, surface_tol, k, nu):
    """_calculate_contact_forces_rod_plane -> (|plane response| per element, no-contact mask)."""
    nrm = normal[:, None]
    total = _node_to_element_force(rod.f_int + rod.f_ext)
    along = normal @ total                                   # force component along the normal
    response = -(nrm * along)
    response[:, along > 0.0] = 0.0                           # pushed away from the plane: no response
    elem_x = _node_to_element_position(rod.x)
    distance = normal @ (elem_x - origin[:, None])
    penetration = np.minimum(distance - rod.radius, 0.0)
    elastic = -k * (nrm * penetration)
    elem_v = _node_to_element_velocity(rod.mass, rod.v)
    damping = -nu * (nrm * (normal @ elem_v))
    total_response = response + elastic + damping
    no_contact = (distance - rod.radius) > surface_tol
    response[:, no_contact] = 0.0
    total_response[:, no_contact] = 0.0
    _elements_to_nodes_inplace(total_response, rod.f_ext)
    return _norm(response), no_contact


def rod_plane_contact_with_anisotropic_friction(rod, origin, normal, surface_tol, slip_tol, k, nu,
                                                kinetic_mu, static_mu):
    """kinetic_mu / static_mu = [forward, backward, sideways] (octopus/build.py:178-186)."""
    mag, no_contact = rod_plane_contact(rod, origin, normal, surface_tol, k, nu)
    nrm = normal[:, None]
    tang = rod.tang
    in_plane = tang - nrm * (normal @ tang)
    axial = in_plane / (_norm(in_plane) + 1e-14)
    elem_v = _node_to_element_velocity(rod.mass, rod.v)
    v_axial_mag = _dot(elem_v, axial)
    v_axial = v_axial_mag * axial
    sgn = np.sign(v_axial_mag)
    mu_k = 0.5 * (kinetic_mu[0] * (1 + sgn) + kinetic_mu[1] * (1 - sgn))
    slip_axial = find_slipping_elements(v_axial, slip_tol)
    # rolling
    rolling = _cross(axial, np.repeat(nrm, rod.n, axis=1))
    arm = -nrm * rod.radius                                  # from the axis to the contact point
    v_roll = _dot(elem_v, rolling)
    Qt = np.transpose(rod.Q, (1, 0, 2))
    spin = _matvec(Qt, _cross(rod.w, _matvec(rod.Q, arm)))    # lab-frame velocity of the contact point
    slip_roll_mag = v_roll + _dot(spin, rolling)
    slip_roll = slip_roll_mag * rolling
    slip_rolling = find_slipping_elements(slip_roll, slip_tol)
    unit = slip_roll + v_axial                               # total slip velocity, unitised
    unit = unit / _norm(unit + 1e-14)
    # kinetic, axial
    f = -((1.0 - slip_axial) * mu_k * mag * _dot(unit, axial) * axial)
    f[:, no_contact] = 0.0
    _elements_to_nodes_inplace(f, rod.f_ext)
    # kinetic, rolling
    f = -((1.0 - slip_rolling) * kinetic_mu[2] * mag * _dot(unit, rolling) * rolling)
    f[:, no_contact] = 0.0
    _elements_to_nodes_inplace(f, rod.f_ext)
    rod.t_ext += _matvec(rod.Q, _cross(arm, f))
    # static, axial: min(mu_s N, pushing force) against the push
    total = _node_to_element_force(rod.f_int + rod.f_ext)
    push = _dot(total, axial)
    psgn = np.sign(push)
    mu_s = 0.5 * (static_mu[0] * (1 + psgn) + static_mu[1] * (1 - psgn))
    fmax = slip_axial * mu_s * mag
    f = -(np.minimum(np.fabs(push), fmax) * psgn * axial)
    f[:, no_contact] = 0.0
    _elements_to_nodes_inplace(f, rod.f_ext)
    # static, rolling: the force that keeps the contact point from slipping
    torques = _matvec(Qt, rod.t_int + rod.t_ext)
    t_axial = _dot(torques, axial)
    f_roll = _dot(total, rolling)
    noslip = -((rod.radius * f_roll - 2.0 * t_axial) / 3.0 / rod.radius)
    fmax = slip_rolling * static_mu[2] * mag
    f = np.minimum(np.fabs(noslip), fmax) * np.sign(noslip) * rolling
    f[:, no_contact] = 0.0
    _elements_to_nodes_inplace(f, rod.f_ext)
    rod.t_ext += _matvec(rod.Q, _cross(arm, f))


# ===============================================================================================
# The rigid head: Cylinder (elastica/rigidbody/cylinder.py, rigid_body.py), the reference's
# FixedJoint2Rigid (utils/custom_elastica/joint.py:20-225) and BodyBoundaryCondition
# (utils/custom_elastica/constraint.py:8-85), assembled as build_octopus does (octopus/build.py:52-217).
# ===============================================================================================
class NumpyCylinder:
    """Cylinder(start, direction, normal, base_length, base_radius, density): one "node" at the
    centre of mass, director rows (normal, direction x normal, direction); inertia as PyElastica
    allocates it — diag(I1, I1, 2 I1) rho L with I1 = A^2 / 4 pi, i.e. I_axis = m r^2 / 2 and
    I_transverse = m r^2 / 4 (the thin-disc value, no L^2 term)."""

    def __init__(self, start, direction, normal, base_length, base_radius, density):
        start, direction, normal = (np.asarray(a, float) for a in (start, direction, normal))
        self.x = (start + direction * base_length / 2).reshape(3, 1)
        self.v = np.zeros((3, 1))
        self.w = np.zeros((3, 1))
        self.Q = np.zeros((3, 3, 1))
        self.Q[0, :, 0] = normal
        self.Q[1, :, 0] = np.cross(direction, normal)
        self.Q[2, :, 0] = direction
        area = np.pi * base_radius * base_radius
        self.mass = np.pi * base_radius * base_radius * base_length * density
        i1 = area * area / (4.0 * np.pi)
        self.J = np.array([i1, i1, 2.0 * i1]) * density * base_length      # diagonal
        self.invJ = 1.0 / self.J
        self.radius, self.length = base_radius, base_length
        self.f_ext = np.zeros((3, 1))
        self.t_ext = np.zeros((3, 1))
        self.fixed_position = self.x[:, 0].copy()           # BodyBoundaryCondition(constrained_position_idx=(0,))

    def kinematic(self, prefac, eps_rot_axis):
        self.x = self.x + prefac * self.v
        w = self.w[:, 0]
        theta = np.sqrt(w @ w)
        u = w / (theta + eps_rot_axis)
        theta = theta * prefac
        s, c1 = np.sin(theta), 1.0 - np.cos(theta)
        K = np.array([[0.0, -u[2], u[1]], [u[2], 0.0, -u[0]], [-u[1], u[0], 0.0]])
        R = np.eye(3) - s * K + c1 * (K @ K)                 # exp(-theta [u]x): the rods' rotation, in matrix form
        self.Q[:, :, 0] = R @ self.Q[:, :, 0]

    def dynamic(self, dt):
        """update_accelerations: a = F / m; alpha = J^-1 ((J w) x w + T), all in the body frame."""
        acc = self.f_ext / self.mass
        jw = self.J[:, None] * self.w
        alpha = self.invJ[:, None] * (_cross(jw, self.w) + self.t_ext)
        self.v = self.v + dt * acc
        self.w = self.w + dt * alpha

    def constrain_values(self):                              # constraint.py:41-59
        self.x[2, 0] = self.fixed_position[2]
        Q = self.Q
        Q[2, 0, 0], Q[2, 1, 0], Q[2, 2, 0] = 0.0, 0.0, 1.0
        for i in range(2):
            length = np.sqrt(Q[i, 0, 0] ** 2 + Q[i, 1, 0] ** 2)
            Q[i, 0, 0] /= length
            Q[i, 1, 0] /= length
            Q[i, 2, 0] = 0.0

    def constrain_rates(self):                               # constraint.py:61-85
        self.v[2, :] = 0.0
        self.w[:2, :] = 0.0

    def zero_external(self):
        self.f_ext[:] = 0.0
        self.t_ext[:] = 0.0


def fixed_joint_to_rigid(head, arm, k, nu, kt, angle_deg, radius):
    """FixedJoint2Rigid.apply_forces then apply_torques (joint.py:47-219) for index_one = -1 (the
    head's only node), index_two = 0 (the arm's base node and element)."""
    pos = head.x[:, 0].copy()
    pos[2] = 0.0                                             # :50-51
    th = angle_deg / 180.0 * np.pi                           # z_rotation, :7-17
    R = np.array([[np.cos(th), -np.sin(th), 0.0], [np.sin(th), np.cos(th), 0.0], [0.0, 0.0, 1.0]])
    binormal = head.Q[1, :, 0]
    conn_dir = -(R @ binormal)                               # :91
    pos = pos + conn_dir * radius                            # :94-95
    dist_vec = arm.x[:, 0] - pos
    dist = np.sqrt(dist_vec @ dist_vec)
    unit = np.zeros(3) if dist <= np.finfo(np.float64).eps * 1e4 else dist_vec / dist
    elastic = k * dist_vec
    rel_v = arm.v[:, 0] - head.v[:, 0]
    damping = -nu * ((rel_v @ unit) * unit)
    force = elastic + damping
    head.f_ext[:, 0] += force                                # :123
    arm.f_ext[:, 0] -= force                                 # :124
    # torques, :174-219
    link = arm.x[:, 1] - arm.x[:, 0]
    target = pos + arm.rest_len[0] * conn_dir
    restoring = -kt * (arm.x[:, 1] - target)
    torque = np.cross(link, restoring)
    head.t_ext[:, 0] -= head.Q[:, :, 0] @ torque
    arm.t_ext[:, 0] += arm.Q[:, :, 0] @ torque


class NumpyOctopus:
    """build_octopus (octopus/build.py:52-217): n_arm rods around a Cylinder head, one
    FixedJoint2Rigid per arm, BodyBoundaryCondition on the head, gravity + damper + plane contact on
    every arm (the head carries neither).  cfg: the attribute names of softrod_config."""

    def __init__(self, cfg):
        self.cfg = cfg
        self.n_arm = int(cfg.n_arm)
        self.arms = [NumpyRod(cfg) for _ in range(self.n_arm)]
        self.angles = [360 / self.n_arm * a for a in range(self.n_arm)]
        self.head = None
        self.time = np.float64(0.0)

    def reset(self, arm_pos, arm_dir):
        c = self.cfg
        for a, rod in enumerate(self.arms):
            rod.reset_straight(arm_pos[a], arm_dir[a], np.array([0.0, 0.0, 1.0]))     # :79-87
        r0 = c.base_radius
        self.head = NumpyCylinder(np.array([0.0, 0.0, -r0]), np.array([0.0, 0.0, 1.0]), np.array([0.0, 1.0, 0.0]),
                                  2 * r0, c.head_radius, c.head_density)             # :93-105
        self.time = np.float64(0.0)

    def _connections(self):
        c = self.cfg
        for a, rod in enumerate(self.arms):
            fixed_joint_to_rigid(self.head, rod, c.joint_k, c.joint_nu, c.joint_kt, self.angles[a], c.head_radius)

    def substep(self):
        c = self.cfg
        dt = c.dt
        for rod in self.arms:
            rod._kinematic(0.5 * dt)
        self.head.kinematic(0.5 * dt, c.eps_rot_axis)
        if c.time_two_half_adds:
            self.time = self.time + 0.5 * dt
        self.head.constrain_values()
        for rod in self.arms:
            rod._forces_and_torques()
        # synchronize: joints, gravity, contact (registration order, octopus/build.py:117-200); with the
        # switch the contact runs before the forcing group: joints, contact, gravity
        self._connections()
        if c.contact_before_forcing:
            for rod in self.arms:
                rod.contact()
        for rod in self.arms:
            rod.forcing()
        if not c.contact_before_forcing:
            for rod in self.arms:
                rod.contact()
        for rod in self.arms:
            rod.dynamic(dt)
        self.head.dynamic(dt)
        for rod in self.arms:
            rod.rates_operators()
        self.head.constrain_rates()
        for rod in self.arms:
            rod._kinematic(0.5 * dt)
        self.head.kinematic(0.5 * dt, c.eps_rot_axis)
        self.time = self.time + (0.5 * dt if c.time_two_half_adds else dt)
        self.head.constrain_values()
        for rod in self.arms:
            rod.zero_external()
        self.head.zero_external()


# =================================================================================================
# COOMM muscle layers (SOFTROD_FEAT_COOMM_MUSCLES) — second transcription, in COOMM's own whole-array
# style as recalled (coomm/actuations/muscles/muscle.py: Muscle.__call__, MuscleForce, TransverseMuscle;
# coomm/actuations/actuation.py: ContinuousActuation, internal_load_to_equivalent_external_load;
# coomm/_rod_tool.py: average2D, sigma_to_shear).  COOMM (git pin /root/reference/uv.lock:173-175) is NOT on disk:
# PARITY UNPINNED, like softrod_oracle.c's apply_muscles, which tests/test_muscles.py holds against this to 1e-12.
# Call sites in the reference: gym_softrobot/envs/octopus/build.py:295-338, arm_push_env.py:197-212,247-274.
# =================================================================================================
def average2D(vector_collection):
    """Voronoi (3, n-1) -> elements (3, n): each Voronoi value goes half to each neighbouring element."""
    out = np.zeros((3, vector_collection.shape[1] + 1))
    out[:, :-1] += 0.5 * vector_collection
    out[:, 1:] += 0.5 * vector_collection
    return out


def force_length_weight_poly(muscle_length, coef):
    """sum_k coef[k] l^k, clipped at zero from below (Chang et al. 2023: max{3.06 l^3 - 13.64 l^2 + 18.01 l - 6.44, 0})."""
    w = np.zeros_like(muscle_length)
    for power, c in enumerate(coef):
        w += c * muscle_length ** power
    return np.where(w < 0.0, 0.0, w)


def muscle_equivalent_loads(Q, sigma, kappa, tangents, radius, rest_radius, rest_lengths, rest_voronoi_lengths,
                            dilatation, voronoi_dilatation, layers, fl_coef, form=0, current_radius=True, tm_law=0):
    """ApplyMuscles for one rod.  layers: list of dicts {kind: 0 longitudinal / 1 transverse, ratio (3, n),
    strength (n,), activation (n,)}.  -> (external force (3, n+1), external couple (3, n), per-layer force (m, n))."""
    n = sigma.shape[1]
    shear = sigma + np.array([[0.0], [0.0], [1.0]])                      # sigma_to_shear
    kappa_e = average2D(kappa)
    internal_force = np.zeros((3, n))
    couple_e = np.zeros((3, n))
    forces = []
    for layer in layers:
        muscle_position = (radius if current_radius else rest_radius) * layer["ratio"]
        muscle_strain = shear + _cross(kappa_e, muscle_position)
        norm = _norm(muscle_strain)
        muscle_tangent = muscle_strain / norm
        muscle_length = norm
        if layer["kind"] == 1 and tm_law == 0:
            muscle_length = 1.0 / np.sqrt(norm)
        weight = force_length_weight_poly(muscle_length, fl_coef)
        muscle_force = layer["activation"] * layer["strength"] * weight
        forces.append(muscle_force)
        f = muscle_force * muscle_tangent
        internal_force += f
        couple_e += _cross(muscle_position, f)
    internal_couple = 0.5 * (couple_e[:, :-1] + couple_e[:, 1:])       # quadrature_kernel(...)[:, 1:-1]
    QT = np.transpose(Q, (1, 0, 2))
    if form == 0:       # F = D^h(Q^T f); tau = D^h(c) + A^h(kappa x c D^) + (e Q t) x f l^
        ext_force = _difference(_matvec(QT, internal_force))
        ext_couple = (_difference(internal_couple) + _trapezoidal(_cross(kappa, internal_couple) * rest_voronoi_lengths)
                      + _cross(_matvec(Q, tangents * dilatation), internal_force) * rest_lengths)
    else:               # PyElastica's own internal-load form applied to (f, c)
        e3 = 1.0 / voronoi_dilatation ** 3
        ext_force = _difference(_matvec(QT, internal_force) / dilatation)
        ext_couple = (_difference(internal_couple * e3)
                      + _trapezoidal(_cross(kappa, internal_couple) * rest_voronoi_lengths * e3)
                      + _cross(_matvec(Q, tangents), internal_force) * rest_lengths)
    return ext_force, ext_couple, np.array(forces)
