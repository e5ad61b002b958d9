"""Golden vectors for the two OctoFlat operators whose source IS on disk, from the reference's
own code:
    FixedJoint2Rigid.apply_forces / apply_torques   gym_softrobot/utils/custom_elastica/joint.py
    BodyBoundaryCondition.constrain_values / rates  gym_softrobot/utils/custom_elastica/constraint.py

Both files are plain NumPy inside `@njit` static methods, but they cannot be imported as they
stand: numba and elastica are not installed.  This script registers two import shims that
contain NO arithmetic — `numba.njit` returning the function unchanged, and an `elastica` module
whose `FreeJoint.__init__(k, nu)` stores its two arguments and whose `ConstraintBase` / `Damping`
are empty classes — loads the two reference files by path (read-only; nothing is copied) and
calls the reference's methods on duck-typed systems (SimpleNamespace with the *_collection
arrays they read), wired as build_octopus wires them (octopus/build.py:109-132: first_rod = the
rigid head with index -1, second_rod = the arm with index 0).  Inputs and outputs are committed
as tests/golden/octo_operator_vectors.npz; tests/test_oracle_golden.py replays them through the
oracle's joint and head-constraint code.

    python tools/make_octo_operator_golden.py
"""
import importlib.util
import sys
import types
from pathlib import Path
from types import SimpleNamespace

import numpy as np
from scipy.spatial.transform import Rotation as Rot

ROOT = Path(__file__).resolve().parents[1]
REF = Path("/root/reference/gym_softrobot/utils/custom_elastica")


def load(name):
    nb = types.ModuleType("numba")
    nb.njit = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))
    sys.modules.setdefault("numba", nb)
    el = types.ModuleType("elastica")

    class FreeJoint:                      # elastica.FreeJoint.__init__ keeps k and nu, nothing else is used
        def __init__(self, k, nu):
            self.k, self.nu = k, nu

    class ConstraintBase:
        def __init__(self, **kwargs):
            pass

    el.FreeJoint, el.ConstraintBase, el.Damping = FreeJoint, ConstraintBase, type("Damping", (), {})
    sys.modules.setdefault("elastica", el)
    spec = importlib.util.spec_from_file_location("ref_" + name, REF / (name + ".py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def main():
    joint = load("joint")
    constraint = load("constraint")
    rng = np.random.default_rng(77)
    J = {k: [] for k in ("head_x", "head_v", "head_Q", "arm_x", "arm_v", "arm_Q", "angle", "rest_len",
                         "head_f", "head_t", "arm_f", "arm_t")}
    k_, nu_, kt_, radius = 1e6, 1e-3, 1.0, 0.04
    for case in range(24):
        angle = 45.0 * (case % 8)
        phi = rng.uniform(-0.6, 0.6)                                     # the head has turned about z
        hQ = np.array([[0.0, 1.0, 0.0], [-1.0, 0.0, 0.0], [0.0, 0.0, 1.0]]) @ Rot.from_euler("z", phi).as_matrix().T
        hx = np.array([rng.uniform(-0.1, 0.1), rng.uniform(-0.1, 0.1), 0.0])
        hv = np.array([rng.uniform(-0.2, 0.2), rng.uniform(-0.2, 0.2), 0.0])
        head = SimpleNamespace(position_collection=hx[:, None].copy(), velocity_collection=hv[:, None].copy(),
                               director_collection=hQ[:, :, None].copy(), external_forces=np.zeros((3, 1)),
                               external_torques=np.zeros((3, 1)))
        # an arm whose first two nodes sit near (not on) their joint position
        d = Rot.from_euler("z", angle + np.degrees(phi), degrees=True).apply([1.0, 0.0, 0.0])
        rest = 0.035
        x0 = hx * [1, 1, 0] + d * radius + rng.normal(0, 2e-4 if case % 3 else 0.0, 3)
        x1 = x0 + rest * (d + rng.normal(0, 0.05, 3))
        ax = np.stack([x0, x1, x1 + rest * d], axis=1)
        av = rng.normal(0, 0.1, (3, 3))
        aQ = np.repeat((Rot.from_rotvec(rng.normal(0, 0.3, 3)).as_matrix())[:, :, None], 2, axis=2)
        arm = SimpleNamespace(position_collection=ax.copy(), velocity_collection=av.copy(),
                              director_collection=aQ.copy(), rest_lengths=np.full(2, rest),
                              external_forces=np.zeros((3, 3)), external_torques=np.zeros((3, 2)))
        j = joint.FixedJoint2Rigid(k=k_, nu=nu_, kt=kt_, angle=angle, radius=radius)
        j.apply_forces(head, -1, arm, 0)
        j.apply_torques(head, -1, arm, 0)
        for key, val in (("head_x", hx), ("head_v", hv), ("head_Q", hQ), ("arm_x", ax), ("arm_v", av),
                         ("arm_Q", aQ), ("angle", angle), ("rest_len", rest),
                         ("head_f", head.external_forces[:, 0]), ("head_t", head.external_torques[:, 0]),
                         ("arm_f", arm.external_forces[:, 0]), ("arm_t", arm.external_torques[:, 0])):
            J[key].append(np.array(val, dtype=np.float64))
    C = {k: [] for k in ("x_in", "Q_in", "v_in", "w_in", "x_out", "Q_out", "v_out", "w_out")}
    for case in range(12):
        Q = Rot.from_rotvec(rng.normal(0, 0.2, 3)).as_matrix() @ np.array([[0.0, 1, 0], [-1, 0, 0], [0, 0, 1]])
        x, v, w = rng.normal(0, 0.1, (3, 1)), rng.normal(0, 0.1, (3, 1)), rng.normal(0, 1.0, (3, 1))
        body = SimpleNamespace(position_collection=x.copy(), director_collection=Q[:, :, None].copy(),
                               velocity_collection=v.copy(), omega_collection=w.copy(),
                               acceleration_collection=np.zeros((3, 1)), alpha_collection=np.zeros((3, 1)))
        bc = constraint.BodyBoundaryCondition(np.zeros(3), np.zeros((3, 3)))
        bc.fixed_position = np.array([0.0, 0.0, 0.0])        # what finalize() hands over: the head's start (z = 0)
        bc.constrain_values(body, 0.0)
        bc.constrain_rates(body, 0.0)
        for key, val in (("x_in", x[:, 0]), ("Q_in", Q), ("v_in", v[:, 0]), ("w_in", w[:, 0]),
                         ("x_out", body.position_collection[:, 0]), ("Q_out", body.director_collection[:, :, 0]),
                         ("v_out", body.velocity_collection[:, 0]), ("w_out", body.omega_collection[:, 0])):
            C[key].append(np.array(val, dtype=np.float64))
    out = ROOT / "tests" / "golden" / "octo_operator_vectors.npz"
    np.savez(out, **{"joint_" + k: np.stack(v) for k, v in J.items()},
             **{"bc_" + k: np.stack(v) for k, v in C.items()},
             joint_params=np.array([k_, nu_, kt_, radius]))
    print("wrote", out)


if __name__ == "__main__":
    main()
