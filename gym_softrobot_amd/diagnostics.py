"""Diagnostic taps: the data the reference's callbacks collect, taken
from the resident state after each env.step instead of from inside PyElastica's stepper.

`RodCallBack` (gym_softrobot/utils/custom_elastica/callback_func.py:23-41) fires when
`current_step % step_skip == 0`, i.e. once per env.step, and appends copies of the rod's
time, radius, dilatation, voronoi_dilatation, position, director, velocity, omega, sigma and
kappa to `callback_params` (soft_pendulum.py:117-126 wires it to `rod_parameters_dict`).
`RodRecorder.record()` appends the same fields for the chosen envs of a batch; the strains
are recomputed here from positions and directors with the formulas of the step kernels
(they are functions of the state, which is all the kernel keeps).  Rendering itself
(matplotlib / POV-Ray) stays out of scope; this is what its inputs would be.
"""
from __future__ import annotations

from collections import defaultdict
from typing import Dict, List, Sequence

import numpy as np


def rod_strains(x: np.ndarray, Q: np.ndarray, rest_length: float, base_radius: float,
                acos_shift: float = 1e-10, eps_sin: float = 1e-14) -> Dict[str, np.ndarray]:
    """x (3, n+1), Q (3, 3, n) -> lengths, dilatation, voronoi_dilatation, radius, sigma, kappa."""
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
        self.acos_shift = float(cfg.acos_shift)
        self.eps_sin = float(cfg.eps_sin)
        self.params: List[Dict[str, list]] = [defaultdict(list) for _ in self.env_indices]

    def record(self) -> None:
        snap = self.backend.rod_snapshot(self.env_indices)
        for k, p in enumerate(self.params):
            x, Q = snap["x"][k], snap["Q"][k]
            s = rod_strains(x, Q, self.rest_length, self.base_radius, self.acos_shift, self.eps_sin)
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
                s = rod_strains(x, Q, self.rest_length, self.base_radius, self.acos_shift, self.eps_sin)
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
