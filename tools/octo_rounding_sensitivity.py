"""How sensitive is OctoFlat-v0 to rounding?  Runs the CPU oracle against a second build of
the SAME source compiled with FMA contraction (-ffp-contract=fast -mfma), i.e. two correct
evaluations of the reference algorithm that differ only in rounding, and prints how far
their observations drift apart (a) over whole env.steps of 2857 substeps and (b) over
200-substep windows with the states re-synchronised before every window.

The result is what the OctoFlat parity tests are designed around: rtol 1e-5
is meaningful over a few hundred substeps, not over whole rollouts.

    python tools/octo_rounding_sensitivity.py
"""
import subprocess
import sys
import tempfile
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

from gym_softrobot_amd import _capi  # noqa: E402
import oracle.oracle_c as oc  # noqa: E402


def flat(ob):
    return np.concatenate([ob["individual"].ravel(), ob["shared"]])


def build_fma(tmp: Path) -> None:
    subprocess.check_call(
        ["cc", "-O2", "-fPIC", "-std=gnu11", "-ffp-contract=fast", "-mfma", "-fno-fast-math", "-w",
         "-shared", "-o", str(tmp / "libsoftrod_oracle.so"), str(ROOT / "oracle" / "softrod_oracle.c"), "-lm"])


def pair(cfg, n, tmp):
    oc._libs.clear(); oc._DIR = ROOT / "oracle"
    a = [oc.OracleOcto(cfg) for _ in range(n)]
    oc._libs.clear(); oc._DIR = tmp
    b = [oc.OracleOcto(cfg) for _ in range(n)]
    for x, y in zip(a, b):
        x.reset([1.0, 1.3]); y.reset([1.0, 1.3])
    return a, b


def main():
    with tempfile.TemporaryDirectory() as d:
        tmp = Path(d)
        build_fma(tmp)
        n = 4
        for amp in (22.0, 3.0):
            cfg = _capi.octo_flat_config(1)
            A, B = pair(cfg, n, tmp)
            rng = np.random.default_rng(9)
            for t in range(4):
                worst = 0.0
                for x, y in zip(A, B):
                    a = rng.uniform(-amp, amp, 24).astype(np.float32)
                    worst = max(worst, np.abs(flat(x.env_step(a)[0]) - flat(y.env_step(a)[0])).max())
                print(f"|a|<={amp:4.1f} whole env.step {t}: max |obs - obs_fma| = {worst:.2e}")
            cfg.n_substeps = 200
            A, B = pair(cfg, n, tmp)
            ratios = []
            for w in range(60):
                for x, y in zip(A, B):
                    if w % 3 == 0:
                        x._a = rng.uniform(-amp, amp, 24).astype(np.float32)
                    y.copy_state_from(x)
                    fa, fb = flat(x.env_step(x._a)[0]), flat(y.env_step(x._a)[0])
                    ratios.append((np.abs(fa - fb) / (1e-5 * np.abs(fa) + 5e-7)).max())
            r = np.array(ratios)
            print(f"|a|<={amp:4.1f} 200-substep windows, resynced: {len(r)} windows, "
                  f"{int((r > 1).sum())} beyond rtol 1e-5 (worst ratio {r.max():.3g})")


if __name__ == "__main__":
    main()
