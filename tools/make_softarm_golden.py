"""Golden vectors for SoftArmTracking-v0's actuation, from the reference's own code:
    MuscleTorquesWithVaryingBetaSplines.apply_torques
        gym_softrobot/utils/custom_elastica/muscle_torque/muscle_torques_with_bspline.py:128-225

The file is NumPy + scipy.interpolate.make_interp_spline (installed here); it needs numba and
elastica only for `@njit` and the empty base class `NoForces`.  As in
tools/make_octo_operator_golden.py, two import shims that hold NO arithmetic are registered
(`njit` returning the function unchanged, `elastica.external_forces.NoForces` an empty class),
the reference file is loaded by path (read-only; nothing is copied) and its methods are called
on duck-typed systems (SimpleNamespace with `lengths` and `external_torques`), wired as
SoftArmTrackingEnv.reset wires them (soft_arm/soft_arm_tracking.py:352-383: base_length 1000,
4 control points, scale 10 * 50 * 2e6, directions "normal" and "binormal", rate limit inf).

Each case is a short sequence of calls with changing control points and element lengths — the
operator is stateful (`points_cached`, the torque profile cached until the control points
change) — and records the external torques it leaves.  Also recorded: the piecewise-cubic form
of the interpolant's cardinal functions (scipy's BSpline -> PPoly), which is what the kernel and
the oracle evaluate; and the reset observation of the env (soft_arm_tracking.py:160-207 on the
straight rod of :268-282, NumPy only).

    python tools/make_softarm_golden.py     -> tests/golden/softarm_vectors.npz
"""
import importlib.util
import sys
import types
from pathlib import Path
from types import SimpleNamespace

import numpy as np
from scipy.interpolate import PPoly, make_interp_spline

ROOT = Path(__file__).resolve().parents[1]
REF = Path("/root/reference/gym_softrobot/utils/custom_elastica/muscle_torque/muscle_torques_with_bspline.py")

BASE_LENGTH, N_CTRL, N_ELEM = 1000.0, 4, 40
ALPHA = 10 * 50 * 2e6


def load():
    nb = types.ModuleType("numba")
    nb.njit = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))
    sys.modules.setdefault("numba", nb)
    el = types.ModuleType("elastica")
    ef = types.ModuleType("elastica.external_forces")
    ef.NoForces = type("NoForces", (), {"__init__": lambda self: None})
    el.external_forces = ef
    sys.modules.setdefault("elastica", el)
    sys.modules.setdefault("elastica.external_forces", ef)
    spec = importlib.util.spec_from_file_location("ref_muscle_bspline", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def cardinal_table():
    """The interpolant is linear in the control values: S(s) = sum_j y_j phi_j(s).  phi_j as
    piecewise cubics: breaks[P + 1], coef[P][N_CTRL][4] in powers of (s - breaks[p])."""
    x = np.linspace(0.0, BASE_LENGTH, N_CTRL + 2)
    pieces = None
    for j in range(N_CTRL):
        y = np.zeros(N_CTRL + 2)
        y[1 + j] = 1.0
        pp = PPoly.from_spline(make_interp_spline(x, y))
        # PPoly.from_spline keeps the repeated end knots as zero-length pieces: drop them
        keep = np.nonzero(np.diff(pp.x) > 0)[0]
        if pieces is None:
            breaks = np.concatenate([pp.x[keep], pp.x[keep[-1] + 1:keep[-1] + 2]])
            pieces = np.zeros((len(keep), N_CTRL, 4))
        pieces[:, j, :] = pp.c[::-1, keep].T          # ascending powers
    return breaks, pieces


def main():
    mod = load()
    rng = np.random.default_rng(2024)
    out = {}
    breaks, coef = cardinal_table()
    out["spline_breaks"], out["spline_coef"] = breaks, coef
    seqs_pts, seqs_len, seqs_tq, seqs_cached = [], [], [], []
    for case in range(6):
        pts_n, pts_b = [], []
        fn = mod.MuscleTorquesWithVaryingBetaSplines(
            base_length=BASE_LENGTH, number_of_control_points=N_CTRL, points_func_array=pts_n,
            muscle_torque_scale=ALPHA, direction="normal", step_skip=10**9, max_rate_of_change_of_activation=np.inf)
        fb = mod.MuscleTorquesWithVaryingBetaSplines(
            base_length=BASE_LENGTH, number_of_control_points=N_CTRL, points_func_array=pts_b,
            muscle_torque_scale=ALPHA, direction="binormal", step_skip=10**9,
            max_rate_of_change_of_activation=np.inf)
        P, Ln, T, Cc = [], [], [], []
        action = np.zeros(2 * N_CTRL)
        for call in range(8):
            if call % 3 == 0:                       # the env sets new control points every 50 substeps
                action = rng.uniform(-1, 1, 2 * N_CTRL).astype(np.float32).astype(np.float64)
                if case == 0 and call == 0:
                    action[:] = 0.0                 # first call with all-zero points: still builds the spline
            pts_n[:] = action[:N_CTRL]
            pts_b[:] = action[N_CTRL:]
            lengths = (BASE_LENGTH / N_ELEM) * (1.0 + 0.02 * rng.standard_normal(N_ELEM))
            rod = SimpleNamespace(lengths=lengths, external_torques=np.zeros((3, N_ELEM)))
            fn.apply_torques(rod, time=0.0)
            fb.apply_torques(rod, time=0.0)
            P.append(action.copy()); Ln.append(lengths.copy()); T.append(rod.external_torques.copy())
            Cc.append(np.concatenate([fn.points_cached[1, 1:-1], fb.points_cached[1, 1:-1]]))
        seqs_pts.append(P); seqs_len.append(Ln); seqs_tq.append(T); seqs_cached.append(Cc)
    out["seq_points"] = np.array(seqs_pts)          # [case][call][8]
    out["seq_lengths"] = np.array(seqs_len)         # [case][call][40]
    out["seq_torques"] = np.array(seqs_tq)          # [case][call][3][40]
    out["seq_cached"] = np.array(seqs_cached)       # [case][call][8]

    # reset observation (soft_arm_tracking.py:160-207): straight rod along +y, kappa = 0, mode 1
    tip = np.array([0.0, BASE_LENGTH, 0.0])
    target = np.array([500.0, 500.0, 500.0])
    out["reset_obs"] = np.concatenate([np.zeros(N_CTRL), np.zeros(N_CTRL), tip / BASE_LENGTH, target / 1000])
    # the segments get_state averages the Voronoi curvatures over (:170-186)
    avg_length = int((N_ELEM - 1) / N_CTRL)
    seg = [(int(np.rint(avg_length * i)), int(np.rint(avg_length * (i + 1)))) for i in range(N_CTRL - 1)]
    seg.append((int(np.rint(avg_length * (N_CTRL - 1))), N_ELEM - 1))
    out["obs_segments"] = np.array(seg)
    np.savez(ROOT / "tests" / "golden" / "softarm_vectors.npz", **out)
    print("breaks", breaks, "coef", coef.shape, "segments", seg)
    print("wrote", ROOT / "tests" / "golden" / "softarm_vectors.npz")


if __name__ == "__main__":
    main()
