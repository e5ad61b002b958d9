"""Golden vectors for the arm-crossing count from the REFERENCE'S OWN function.

gym_softrobot/utils/intersection.py is pure NumPy apart from its `@njit(cache=True)`
decorators; numba is not installed here.  `njit` does not change what a function computes, so
this script registers a module named `numba` whose `njit` returns the function unchanged, loads
the reference file by path (read-only, nothing is copied) and evaluates `intersection` on
arm-like polylines.  The vectors (inputs + number of intersections) are committed as
tests/golden/intersection_vectors.npz; the oracle's and the GPU kernel's crossing counts are
tested against them.

    python tools/make_intersection_golden.py
"""
import importlib.util
import sys
import types
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
REF = Path("/root/reference/gym_softrobot/utils/intersection.py")


def load_reference_intersection():
    shim = types.ModuleType("numba")
    shim.njit = lambda *a, **k: (a[0] if a and callable(a[0]) else (lambda f: f))
    sys.modules.setdefault("numba", shim)
    spec = importlib.util.spec_from_file_location("ref_intersection", REF)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.intersection


def arm(rng, base, heading, n=10, length=0.35, curl=8.0):
    """A planar polyline of n+1 nodes: constant segment length, random-walk curvature."""
    ang = heading + np.cumsum(rng.uniform(-curl, curl, n) * length / n)
    seg = np.stack([np.cos(ang), np.sin(ang)]) * (length / n)
    return np.concatenate([np.array(base)[:, None], np.array(base)[:, None] + np.cumsum(seg, axis=1)], axis=1)


def main():
    intersection = load_reference_intersection()
    rng = np.random.default_rng(2024)
    p1s, p2s, counts = [], [], []
    while len(counts) < 40:
        th = rng.uniform(0, 2 * np.pi)
        a = arm(rng, (0.04 * np.cos(th), 0.04 * np.sin(th)), th + rng.uniform(-0.5, 0.5), curl=rng.choice([4.0, 12.0, 25.0]))
        th2 = th + rng.uniform(0.2, 1.2)
        b = arm(rng, (0.04 * np.cos(th2), 0.04 * np.sin(th2)), th2 + rng.uniform(-1.5, 0.5), curl=rng.choice([4.0, 12.0, 25.0]))
        try:
            xs, ys = intersection(a, b)
        except np.linalg.LinAlgError:      # parallel candidate segments: the reference raises
            continue
        p1s.append(a); p2s.append(b); counts.append(len(xs))
    out = ROOT / "tests" / "golden" / "intersection_vectors.npz"
    np.savez(out, p1=np.stack(p1s), p2=np.stack(p2s), count=np.array(counts, np.int32))
    print("wrote", out, "counts:", np.bincount(counts))


if __name__ == "__main__":
    main()
