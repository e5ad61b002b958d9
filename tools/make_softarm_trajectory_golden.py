"""Golden vectors for SoftArmTracking's moving target (game_mode 2), from the reference's own
`generate_trajectory` (gym_softrobot/envs/soft_arm/soft_arm_tracking.py:46-101).

The module cannot be imported (gymnasium and elastica are absent), but the function is
module-level NumPy: this script parses the reference file, takes the one function definition
out of its syntax tree and executes it with NumPy — the reference's source is read where it
lies and nothing of it is stored.  The committed vectors are the trajectory at every 50th
sample (the env.step boundaries) plus a few samples in between, for three seeds of
`Generator(PCG64(SeedSequence(seed)))` — Gymnasium's `np_random`.

    python tools/make_softarm_trajectory_golden.py   -> tests/golden/softarm_trajectory.npz
"""
import ast
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
REF = Path("/root/reference/gym_softrobot/envs/soft_arm/soft_arm_tracking.py")


def main():
    tree = ast.parse(REF.read_text())
    fn = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "generate_trajectory")
    ns = {"np": np}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), str(REF), "exec"), ns)
    out = {"seeds": np.array([0, 1, 42])}
    for seed in out["seeds"]:
        rng = np.random.Generator(np.random.PCG64(np.random.SeedSequence(int(seed))))
        w = ns["generate_trajectory"](5, 2.0e-4, 0.1, rng)
        out[f"every50_{seed}"] = w[::50]
        out[f"probe_{seed}"] = w[[1, 7, 12345, 27499]]
        out[f"next_draw_{seed}"] = np.array([rng.random()])      # the stream position after the call
    out["shape"] = np.array(w.shape)
    np.savez(ROOT / "tests" / "golden" / "softarm_trajectory.npz", **out)
    print("wrote", ROOT / "tests" / "golden" / "softarm_trajectory.npz", w.shape)


if __name__ == "__main__":
    main()
