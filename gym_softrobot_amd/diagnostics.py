"""Diagnostic taps: the data the reference's callbacks collect, taken
from the resident state after each env.step instead of from inside PyElastica's stepper.

`RodCallBack` (gym_softrobot/utils/custom_elastica/callback_func.py:23-41) fires when
`current_step % step_skip == 0`, i.e. once per env.step, and appends copies of the rod's
time, radius, dilatation, voronoi_dilatation, position, director, velocity, omega, sigma and
kappa to `callback_params` (soft_pendulum.py:117-126 wires it to `rod_parameters_dict`).
`RodRecorder.record()` appends the same fields for the chosen envs of a batch.

WHICH INSTANT.  `system.sigma / kappa / dilatation / voronoi_dilatation / radius` are PyElastica's
cached arrays: they were last written by the force evaluation of the final substep, i.e. at the
MID-substep configuration (after the first kinematic half step and constrain_values), not at the
end-of-step state that `position_collection` etc. hold.  The kernels keep the
state, not those caches, so the mid-substep configuration is rebuilt here exactly: the closing
half step of PositionVerlet is x_end = x_mid + (dt/2) v_end, Q_end = R((dt/2) omega_end) Q_mid with
the END rates (nothing touches v, omega after the rate update), hence
    x_mid = x_end - (dt/2) v_end,   Q_mid = R((dt/2) omega_end)^T Q_end,
followed by the boundary condition's constrain_values (which the reference applied at that
instant: build.py:71-74, soft_pendulum_3d/build.py:31-34), and the strains are evaluated there with
the formulas of the step kernels.  The tangents this yields are the row the kernel itself caches at
the force evaluation (softrod_state_view.tangents) to rounding — tests/test_gpu_parity.py checks
that.  Rendering itself (matplotlib / POV-Ray) stays out of scope; this is what its inputs would be.
"""
from __future__ import annotations

from collections import defaultdict
from typing import Dict, List, Sequence

import numpy as np


def half_step_rotation(omega: np.ndarray, h: float, eps_rot_axis: float = 1e-14) -> np.ndarray:
    """R (3, 3, n) of the kinematic update Q <- R Q for the step h (PyElastica _get_rotation_matrix:
    axis = omega / (|omega| + eps), angle = h |omega|; the TRANSPOSED Rodrigues matrix)."""
    th = np.sqrt((omega * omega).sum(axis=0))
    a = omega / (th + eps_rot_axis)
    ang = th * h
    up, usq = np.sin(ang), 1.0 - np.cos(ang)
    R = np.empty((3, 3, omega.shape[1]))
    R[0, 0] = 1.0 - usq * (a[1] * a[1] + a[2] * a[2])
    R[1, 1] = 1.0 - usq * (a[0] * a[0] + a[2] * a[2])
    R[2, 2] = 1.0 - usq * (a[0] * a[0] + a[1] * a[1])
    R[0, 1] = up * a[2] + usq * a[0] * a[1]
    R[1, 0] = -up * a[2] + usq * a[0] * a[1]
    R[0, 2] = -up * a[1] + usq * a[0] * a[2]
    R[2, 0] = up * a[1] + usq * a[0] * a[2]
    R[1, 2] = up * a[0] + usq * a[1] * a[2]
    R[2, 1] = -up * a[0] + usq * a[1] * a[2]
    return R


def mid_substep_configuration(x, v, Q, w, dt: float, eps_rot_axis: float = 1e-14):
    """(x_mid, Q_mid): the configuration at which the LAST substep evaluated its forces (module
    docstring), before the boundary condition's constrain_values."""
    R = half_step_rotation(w, 0.5 * dt, eps_rot_axis)
    return x - 0.5 * dt * v, np.einsum("jik,jlk->ilk", R, Q)       # R^T Q


def constrain_values_host(features: int, x, Q, fixed_pos, fixed_dir, base_xy=None):
    """The boundary condition's constrain_values on node 0 / element 0, in place (what the kernels'
    constrain_values_n does; build.py:71-74, soft_pendulum_3d/build.py:31-34, OneEndFixedBC)."""
    from . import _capi

    if features & _capi.FEAT_PENDULUM_BC:
        x[1, 0], x[2, 0] = fixed_pos[1], fixed_pos[2]
        Q[0, :, 0], Q[2, :, 0] = fixed_dir[0], fixed_dir[2]       # row 1 untouched
    if features & _capi.FEAT_FIXED_BC:
        x[:, 0] = fixed_pos
        Q[:, :, 0] = fixed_dir
    if features & _capi.FEAT_MOVING_BASE_BC:
        x[0, 0], x[1, 0], x[2, 0] = base_xy[0], base_xy[1], fixed_pos[2]
        Q[:, :, 0] = fixed_dir


def rod_strains(x: np.ndarray, Q: np.ndarray, rest_length: float, base_radius,
                acos_shift: float = 1e-10, eps_sin: float = 1e-14) -> Dict[str, np.ndarray]:
    """x (3, n+1), Q (3, 3, n) -> lengths, dilatation, voronoi_dilatation, radius, sigma, kappa.
    base_radius: the rest radius, a scalar or one per element (a tapered rod)."""
    d = x[:, 1:] - x[:, :-1]
    lengths = np.sqrt((d * d).sum(axis=0)) + 1e-14
    tangents = d / lengths
    dilatation = lengths / rest_length
    vor = 0.5 * (lengths[1:] + lengths[:-1])
    voronoi_dilatation = vor / rest_length            # uniform rod: rest Voronoi length = rest length
    radius = base_radius * np.sqrt(rest_length / lengths)      # volume-preserving
    sigma = dilatation * np.einsum("ijk,jk->ik", Q, tangents)
    sigma[2] -= 1.0
    # kappa = -log(Q_{k+1} Q_k^T) / D  (_inv_rotate)
    R = np.einsum("ijk,ljk->ilk", Q[:, :, 1:], Q[:, :, :-1])
    vec = np.stack([R[2, 1] - R[1, 2], R[0, 2] - R[2, 0], R[1, 0] - R[0, 1]])
    trace = R[0, 0] + R[1, 1] + R[2, 2]
    theta = np.arccos(np.clip(0.5 * trace - 0.5 - acos_shift, -1.0, 1.0))
    kappa = vec * (-0.5 * theta / np.sin(theta + eps_sin)) / rest_length
    return {"lengths": lengths, "dilatation": dilatation, "voronoi_dilatation": voronoi_dilatation,
            "radius": radius, "sigma": sigma, "kappa": kappa, "tangents": tangents}


class RodRecorder:
    """Collects RodCallBack's fields for `env_indices` of a batch; one dict of lists per env,
    keyed like the reference's `rod_parameters_dict`."""

    FIELDS = ("time", "radius", "dilatation", "voronoi_dilatation", "position", "director", "velocity",
              "omega", "sigma", "kappa")

    def __init__(self, backend, env_indices: Sequence[int] = (0,)):
        self.backend = backend
        self.env_indices = [int(i) for i in env_indices]
        cfg = backend.cfg
        self.rest_length = float(cfg.base_length) / int(cfg.n_elem)
        self.base_radius = float(cfg.base_radius)
        prof = getattr(backend, "_tables", {}).get("radius_profile")
        if prof is not None:                                    # tapered rod: per-element rest radii
            self.base_radius = np.frombuffer(prof, np.float64).copy()
        self.acos_shift = float(cfg.acos_shift)
        self.eps_sin = float(cfg.eps_sin)
        self.eps_rot_axis = float(cfg.eps_rot_axis)
        self.dt = float(cfg.dt)
        self.features = int(cfg.features)
        self.params: List[Dict[str, list]] = [defaultdict(list) for _ in self.env_indices]
        self.last_mid_tangents: List[np.ndarray] = []           # of the latest record(): test access

    def _bc(self):
        """fixed_position (k, 3), fixed_directors (k, 3, 3), moving-base position (k, 2) of the recorded envs."""
        st = self.backend.state()
        idx = self.env_indices
        bc = st["bc_targets"][:, idx].cpu().numpy()             # (12, k)
        ctrl = st["control"][:2, idx].cpu().numpy()             # (2, k)
        return bc[:3].T, bc[3:].T.reshape(len(idx), 3, 3), ctrl.T

    def record(self) -> None:
        snap = self.backend.rod_snapshot(self.env_indices)
        fpos, fdir, base = self._bc()
        self.last_mid_tangents = []
        for k, p in enumerate(self.params):
            x, Q = snap["x"][k], snap["Q"][k]
            xm, Qm = mid_substep_configuration(x, snap["v"][k], Q, snap["w"][k], self.dt, self.eps_rot_axis)
            constrain_values_host(self.features, xm, Qm, fpos[k], fdir[k], base[k])
            s = rod_strains(xm, Qm, self.rest_length, self.base_radius, self.acos_shift, self.eps_sin)
            self.last_mid_tangents.append(s["tangents"])
            p["time"].append(float(snap["time"][k]))
            p["radius"].append(s["radius"])
            p["dilatation"].append(s["dilatation"])
            p["voronoi_dilatation"].append(s["voronoi_dilatation"])
            p["position"].append(x.copy())
            p["director"].append(Q.copy())
            p["velocity"].append(snap["v"][k].copy())
            p["omega"].append(snap["w"][k].copy())
            p["sigma"].append(s["sigma"])
            p["kappa"].append(s["kappa"])


class OctoRecorder:
    """FlatEnv's taps (octopus/flat_env.py:188-206): one RodCallBack dict per arm
    (`rod_parameters_dict_list`, when config_generate_video) and the head's
    RigidCylinderCallBack dict (`head_dict`: time, step, position, velocity —
    callback_func.py:4-20 — when config_save_head_data), for ONE env of the batch, sampled
    once per env.step like the reference's `current_step % step_skip == 0`."""

    def __init__(self, backend, env_index: int = 0, rods: bool = True, head: bool = True):
        self.backend = backend
        self.env_index = int(env_index)
        cfg = backend.cfg
        self.n_arm = int(cfg.n_arm)
        self.rest_length = float(cfg.base_length) / int(cfg.n_elem)
        self.base_radius = float(cfg.base_radius)
        self.acos_shift = float(cfg.acos_shift)
        self.eps_sin = float(cfg.eps_sin)
        self.n_substeps = int(cfg.n_substeps)
        self.eps_rot_axis = float(cfg.eps_rot_axis)
        self.dt = float(cfg.dt)
        self.rod_parameters_dict_list = [defaultdict(list) for _ in range(self.n_arm)] if rods else None
        self.head_dict = defaultdict(list) if head else None
        self._steps = 0

    def record(self) -> None:
        st = self.backend.octo_state_numpy()
        e = self.env_index
        self._steps += 1
        t = float(st["time"][e])
        if self.rod_parameters_dict_list is not None:
            for a, p in enumerate(self.rod_parameters_dict_list):
                x, Q = st["x"][e, a], st["Q"][e, a]
                # the arms carry no boundary condition of their own (the joints hold them): the
                # mid-substep configuration is the plain back half step
                xm, Qm = mid_substep_configuration(x, st["v"][e, a], Q, st["w"][e, a], self.dt, self.eps_rot_axis)
                s = rod_strains(xm, Qm, self.rest_length, self.base_radius, self.acos_shift, self.eps_sin)
                p["time"].append(t)
                p["radius"].append(s["radius"])
                p["dilatation"].append(s["dilatation"])
                p["voronoi_dilatation"].append(s["voronoi_dilatation"])
                p["position"].append(x.copy())
                p["director"].append(Q.copy())
                p["velocity"].append(st["v"][e, a].copy())
                p["omega"].append(st["w"][e, a].copy())
                p["sigma"].append(s["sigma"])
                p["kappa"].append(s["kappa"])
        if self.head_dict is not None:
            self.head_dict["time"].append(t)
            self.head_dict["step"].append(self._steps * self.n_substeps)
            self.head_dict["position"].append(st["head_x"][e].reshape(3, 1).copy())
            self.head_dict["velocity"].append(st["head_v"][e].reshape(3, 1).copy())
