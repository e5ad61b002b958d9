"""Episode-length parity: the HIP path against the oracle over WHOLE
episodes, at north_star's rtol 1e-5, with the horizons the measurement supports
(profiles/parity_episode.json, tools/episode_parity.py — which also records the control: the
same oracle source built with FMA contraction):

  SoftPendulum-v0     zero action / random +-22 N: all 126 steps (truncation fires on #126)
                      stabilising script (keeps the inverted pendulum near its UNSTABLE
                      equilibrium, where rounding differences grow ~e^{3.8 t}): the control itself
                      leaves 1e-5 after 107 steps, the kernel after 89 -> asserted for 60 steps,
                      and over all 126 with the state re-synchronised every 5 steps
  SoftPendulum3D-v0   all 125 steps
  OctoArmSingle-v0    all 201 steps (to truncation)

Every scenario runs on BOTH kernels (VERDICT r2 "next" #2): the default fast-math kernel and the
libm kernel (SOFTROD_MATH_LIBM, the substep as PyElastica writes it).  The measured horizons of
both, and what each fast-math reformulation costs, are in profiles/r3_fastmath_cost.json
(tools/fastmath_cost.sh) and DESIGN.md §3.
"""
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]
TOL = 1e-5
MODES = pytest.mark.parametrize("math_mode", [1, 0], ids=["fast", "libm"])     # _capi.MATH_FAST / MATH_LIBM


@pytest.fixture(scope="module")
def ep(hip_lib, oracle_built):
    import torch

    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    sys.path.insert(0, str(ROOT / "tools"))
    import episode_parity

    return episode_parity


def _worst(res, upto=None):
    c = res["curves"]
    return max(max(c["gpu_obs"][:upto]), max(c["gpu_reward"][:upto]))


@MODES
@pytest.mark.parametrize("script", ["zero", "random"])
def test_softpendulum_whole_episode(ep, script, math_mode):
    n = 6
    acts = np.random.default_rng(3).uniform(-22, 22, (126, n, 1)).astype(np.float32)
    if script == "zero":
        acts[:] = 0.0
    res = ep.run("SoftPendulum-v0", n, 126, lambda t, o: acts[t], with_control=False, math_mode=math_mode)
    assert res["flags_equal_all_steps"]               # incl. truncated on step 126 and only there
    assert _worst(res) <= TOL, res["steps_within_1e-5"]


@MODES
def test_softpendulum_near_the_unstable_equilibrium(ep, math_mode):
    n = 6
    prev = {"th": None}

    def pd(t, obs):
        x, v, th = (obs[:, k].astype(np.float64) for k in (0, 1, 3))
        dth = np.zeros_like(th) if t == 0 else (th - prev["th"]) / 0.04
        prev["th"] = th.copy()
        return np.clip(100.0 * th + 20.0 * dth + 10.0 * x + 8.0 * v, -22, 22).astype(np.float32)[:, None]

    res = ep.run("SoftPendulum-v0", n, 126, pd, with_control=False, math_mode=math_mode)
    assert res["flags_equal_all_steps"]
    assert _worst(res, 60) <= TOL, res["steps_within_1e-5"]
    res = ep.run("SoftPendulum-v0", n, 126, pd, window=5, with_control=False, math_mode=math_mode)
    assert res["flags_equal_all_steps"] and _worst(res) <= TOL, res["steps_within_1e-5"]


@MODES
def test_softpendulum3d_whole_episode(ep, math_mode):
    n = 4
    acts = np.random.default_rng(4).uniform(-1, 1, (125, n, 2)).astype(np.float32)
    res = ep.run("SoftPendulum3D-v0", n, 125, lambda t, o: acts[t], with_control=False, math_mode=math_mode)
    assert res["flags_equal_all_steps"] and _worst(res) <= TOL, res["steps_within_1e-5"]


@MODES
def test_armsingle_to_truncation(ep, math_mode):
    n = 3
    acts = np.random.default_rng(5).uniform(-6, 6, (201, n, 7)).astype(np.float32)
    res = ep.run("OctoArmSingle-v0", n, 201, lambda t, o: acts[t], with_control=False, math_mode=math_mode)
    assert res["flags_equal_all_steps"] and _worst(res) <= TOL, res["steps_within_1e-5"]
