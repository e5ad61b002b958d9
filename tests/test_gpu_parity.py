"""Parity tests proper: the HIP path, called through the C-ABI (ctypes -> libsoftrod_hip.so),
against the fp64 CPU oracle on the same seeds and actions.  Tolerance: rtol 1e-5 on
observations/rewards (BASELINE.json north_star), exact on flags.  Both math modes of the
kernel are held to the same bar."""
import json
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = Path(__file__).parent / "golden"
RTOL = 1e-5


def _theta(seed):
    from gym_softrobot_amd.seeding import initial_angle, np_random

    rng, _ = np_random(seed)
    return initial_angle(rng)


@pytest.fixture(scope="module")
def torch_gpu():
    import torch

    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    return torch


def _make(n, math_mode, **kw):
    import gym_softrobot_amd as gsa

    return gsa.make_vec("SoftPendulum-v0", n, device=0, math_mode=math_mode, **kw)


@pytest.mark.parametrize("math_mode", [0, 1], ids=["libm", "fast"])
def test_reset_observation_golden(torch_gpu, hip_lib, math_mode):
    vectors = json.loads((GOLD / "softpendulum_reset.json").read_text())
    env = _make(len(vectors), math_mode)
    obs, info = env.reset(seed=[v["seed"] for v in vectors])
    obs = obs.cpu().numpy()
    assert obs.dtype == np.float32 and info == {}
    for o, v in zip(obs, vectors):
        np.testing.assert_array_equal(o[:3], 0.0)
        assert abs(float(o[3]) - v["obs"][3]) <= 2 * np.spacing(np.float32(abs(v["obs"][3])))
    env.close()


@pytest.mark.parametrize("math_mode", [0, 1], ids=["libm", "fast"])
def test_rollout_matches_oracle(torch_gpu, hip_lib, oracle_built, math_mode):
    n, T = 16, 6
    env = _make(n, math_mode)
    env.reset(seed=0)
    acts = np.random.default_rng(1).uniform(-22, 22, (T, n)).astype(np.float32)
    rods = []
    for i in range(n):
        r = oracle_built.OracleRod(env.cfg)
        r.reset_pendulum(_theta(i))
        rods.append(r)
    worst = 0.0
    for t in range(T):
        obs, rew, term, trunc, info = env.step(acts[t])
        obs, rew = obs.cpu().numpy(), rew.cpu().numpy()
        term, trunc = term.cpu().numpy(), trunc.cpu().numpy()
        for i, r in enumerate(rods):
            o, rw, te, tr = r.env_step(acts[t, i])
            np.testing.assert_allclose(obs[i], o, rtol=RTOL, atol=1e-7)
            np.testing.assert_allclose(rew[i], rw, rtol=RTOL, atol=1e-9)
            assert bool(term[i]) == te and bool(trunc[i]) == tr
            worst = max(worst, np.max(np.abs(obs[i] - o) / (np.abs(o) + 1e-3)))
    print(f"math_mode={math_mode}: worst |dobs|/(|obs|+1e-3) over {T} steps = {worst:.3e}")
    # full state after T steps
    st = env.backend.state_numpy()
    for i, r in enumerate(rods):
        np.testing.assert_allclose(st["x"][i], r.get("x"), rtol=RTOL, atol=1e-8)
        np.testing.assert_allclose(st["v"][i], r.get("v"), rtol=RTOL, atol=1e-8)
        np.testing.assert_allclose(st["Q"][i], r.get("Q"), rtol=RTOL, atol=1e-8)
        np.testing.assert_allclose(st["w"][i], r.get("w"), rtol=RTOL, atol=1e-8)
        # north_star's tolerance is RELATIVE: every field against its own scale, no absolute floor
        # (the entry-wise checks above carry an atol for the entries that pass through zero)
        for name in ("x", "v", "Q", "w"):
            ref = r.get(name)
            assert np.max(np.abs(st[name][i] - ref)) <= RTOL * np.max(np.abs(ref)), name
        assert st["time"][i] == r.time
    env.close()


@pytest.mark.parametrize("math_mode", [0, 1], ids=["libm", "fast"])
def test_golden_oracle_rollout_fixture(torch_gpu, hip_lib, math_mode):
    # committed fixture (tools/make_golden.py): does not need the oracle at run time
    z = np.load(GOLD / "softpendulum_oracle_rollout.npz")
    seeds = [int(s) for s in z["seeds"]]
    env = _make(len(seeds), math_mode)
    env.reset(seed=seeds)
    for t in range(z["actions"].shape[0]):
        obs, rew, term, trunc, _ = env.step(z["actions"][t])
        np.testing.assert_allclose(obs.cpu().numpy(), z["obs"][t], rtol=RTOL, atol=1e-7)
        np.testing.assert_allclose(rew.cpu().numpy(), z["reward"][t], rtol=RTOL, atol=1e-9)
        assert not term.any() and not trunc.any()
    x = env.backend.state_numpy()["x"]
    np.testing.assert_allclose(x, z["x_final"], rtol=RTOL, atol=1e-8)
    env.close()


def test_other_envs_golden_fixtures(torch_gpu, hip_lib):
    """The committed fixtures of the other envs (tests/golden/other_envs_*): reference-derived
    reset observations and the oracle's regression pins, without the oracle at run time."""
    import gym_softrobot_amd as gsa
    from gym_softrobot_amd import _capi
    from gym_softrobot_amd.backend import HipRodBackend

    vec = json.loads((GOLD / "other_envs_reset.json").read_text())
    v3 = vec["SoftPendulum3D-v0"]
    env = gsa.make_vec("SoftPendulum3D-v0", len(v3), device=0)
    obs, _ = env.reset(seed=[v["seed"] for v in v3])
    np.testing.assert_allclose(obs.cpu().numpy(), np.array([v["obs"] for v in v3], np.float32), rtol=1e-6, atol=1e-9)
    env.close()
    env = gsa.make_vec("OctoArmSingle-v0", 1, device=0)
    obs, _ = env.reset(seed=0)
    np.testing.assert_allclose(obs.cpu().numpy()[0], np.array(vec["OctoArmSingle-v0"][0]["obs"], np.float32),
                               rtol=1e-6, atol=1e-7)
    env.close()
    vo = vec["OctoFlat-v0"]
    env = gsa.make_vec("OctoFlat-v0", len(vo), device=0)
    obs, _ = env.reset(seed=[v["seed"] for v in vo])
    d = env.split_obs(obs.cpu().numpy())
    for i, v in enumerate(vo):
        np.testing.assert_allclose(d["individual"][i], np.array(v["individual"], np.float32), rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(d["shared"][i], np.array(v["shared"], np.float32), rtol=1e-6, atol=1e-7)
    env.close()

    z = np.load(GOLD / "other_envs_oracle_rollout.npz")
    be = HipRodBackend(_capi.softpendulum3d_config(1), 0)
    t0 = np.deg2rad(0.37)
    be.reset_straight([[0, 0, 0]], [[np.sin(t0), 0.0, np.cos(t0)]], [[0, 1.0, 0]])
    for t in range(3):
        o, r, *_ = be.step(z["p3d_actions"][t][None])
        np.testing.assert_allclose(o.cpu().numpy()[0], z["p3d_obs"][t], rtol=RTOL, atol=1e-7)
        np.testing.assert_allclose(float(r[0]), z["p3d_reward"][t], rtol=RTOL, atol=1e-9)
    np.testing.assert_allclose(be.state_numpy()["x"][0], z["p3d_x"], rtol=RTOL, atol=1e-8)
    be.close()
    be = HipRodBackend(_capi.arm_single_config(1), 0)
    be.reset_straight([[0, 0, 0]], [[1.0, 0, 0]], [[0, 0, 1.0]])
    be.observe(None)
    for t in range(3):
        o, r, *_ = be.step(z["arm_actions"][t][None])
        np.testing.assert_allclose(o.cpu().numpy()[0], z["arm_obs"][t], rtol=RTOL, atol=2e-6)
        np.testing.assert_allclose(float(r[0]), z["arm_reward"][t], rtol=RTOL, atol=1e-7)
    np.testing.assert_allclose(be.state_numpy()["x"][0], z["arm_x"], rtol=RTOL, atol=1e-7)
    be.close()
    cfg = _capi.octo_flat_config(1)
    cfg.n_substeps = 200
    be = HipRodBackend(cfg, 0)
    be.reset_octo(z["octo_target"][None])
    for t in range(2):
        o, r, *_ = be.step(z["octo_actions"][t][None])
        ref = np.concatenate([z["octo_individual"][t].ravel(), z["octo_shared"][t]])
        np.testing.assert_allclose(o.cpu().numpy()[0], ref, rtol=RTOL, atol=5e-7)
        np.testing.assert_allclose(float(r[0]), z["octo_reward"][t], rtol=RTOL, atol=1e-8)
    be.close()


def test_fast_and_libm_modes_agree(torch_gpu, hip_lib):
    n, T = 64, 10
    acts = np.random.default_rng(3).uniform(-22, 22, (T, n)).astype(np.float32)
    outs = []
    for mode in (0, 1):
        env = _make(n, mode)
        env.reset(seed=100)
        for t in range(T):
            obs, rew, *_ = env.step(acts[t])
        outs.append((obs.cpu().numpy().copy(), rew.cpu().numpy().copy(), env.backend.state_numpy()))
        env.close()
    np.testing.assert_allclose(outs[0][0], outs[1][0], rtol=RTOL, atol=1e-7)
    np.testing.assert_allclose(outs[0][1], outs[1][1], rtol=RTOL, atol=1e-9)
    np.testing.assert_allclose(outs[0][2]["x"], outs[1][2]["x"], rtol=RTOL, atol=1e-8)
    # The two kernels evaluate the SAME formulas (round 3: the fast paths apply the reference's
    # sin(theta + eps_sin) like the libm kernel does), so after 10 env.steps they differ by roundings
    # only: 9e-15 of the position scale and 3e-13 of the reward measured, against 1.9e-13 and 6.5e-11
    # with the term dropped (variants/libsoftrod_diag_NO_EPS_SIN.so).  A systematic difference
    # between the two paths shows here long before it costs a whole-episode tolerance.
    x0, x1 = outs[0][2]["x"], outs[1][2]["x"]
    assert np.max(np.abs(x0 - x1)) <= 5e-14 * np.max(np.abs(x0))
    assert np.max(np.abs(outs[0][1] - outs[1][1]) / np.abs(outs[0][1])) <= 5e-12


def test_fast_and_libm_3d_kernels_agree_at_rounding_level(torch_gpu, hip_lib):
    """The same guard for the general 3-D fast loop (eps_sin_factor, softrod_fast.hpp) on
    SoftPendulum3D-v0, whose motion is not chaotic: after 10 env.steps the directors agree to
    5e-14 and the reward to 3e-13 of its size (measured), against 1.9e-13 and 2.0e-12 with the
    sin(theta + eps_sin) term dropped."""
    import gym_softrobot_amd as gsa

    n, T = 32, 10
    outs = []
    for mode in (0, 1):
        env = gsa.make_vec("SoftPendulum3D-v0", n, device=0, math_mode=mode)
        env.reset(seed=100)
        acts = np.random.default_rng(3).uniform(-1.0, 1.0, (T, n, env.action_dim)).astype(np.float32)
        for t in range(T):
            obs, rew, *_ = env.step(acts[t])
        outs.append((obs.cpu().numpy().copy(), rew.cpu().numpy().copy(), env.backend.state_numpy()))
        env.close()
    np.testing.assert_array_equal(outs[0][0], outs[1][0])                   # float32 observations: identical
    q0, q1 = outs[0][2]["Q"], outs[1][2]["Q"]
    assert np.max(np.abs(q0 - q1)) <= 1e-13 * np.max(np.abs(q0))
    assert np.max(np.abs(outs[0][1] - outs[1][1]) / np.abs(outs[0][1])) <= 1e-12


def test_bitwise_determinism_and_batch_independence(torch_gpu, hip_lib):
    # K7 (tests/envs/test_determinism.py:46-54 of the reference) and K8
    n, T = 32, 3
    acts = np.random.default_rng(0).uniform(-22, 22, (T, n)).astype(np.float32)

    def run(idx):
        env = _make(len(idx), 1)
        env.reset(seed=[int(i) for i in idx])
        res = []
        for t in range(T):
            o, r, te, tr, _ = env.step(acts[t, idx])
            res.append((o.cpu().numpy().copy(), r.cpu().numpy().copy()))
        st = env.backend.state_numpy()
        env.close()
        return res, st

    full1, st1 = run(np.arange(n))
    full2, st2 = run(np.arange(n))
    for (o1, r1), (o2, r2) in zip(full1, full2):
        np.testing.assert_array_equal(o1, o2)
        np.testing.assert_array_equal(r1, r2)
    for k in ("x", "v", "Q", "w"):
        np.testing.assert_array_equal(st1[k], st2[k])
    sub = np.array([5, 17, 31])
    part, stp = run(sub)
    for (o1, r1), (o2, r2) in zip(full1, part):
        np.testing.assert_array_equal(o1[sub], o2)
        np.testing.assert_array_equal(r1[sub], r2)
    np.testing.assert_array_equal(st1["x"][sub], stp["x"])


def test_single_env_facade_api(torch_gpu, hip_lib, oracle_built):
    # mirrors tests/envs/test_envs.py:27-47 and test_determinism.py of the reference
    import gym_softrobot_amd as gsa

    env = gsa.make("SoftPendulum-v0")
    ob, info = env.reset(seed=0)
    assert isinstance(info, dict) and env.observation_space.contains(ob)
    assert ob.dtype == env.observation_space.dtype
    env.action_space.seed(0)
    a = env.action_space.sample()
    o, r, te, tr, inf = env.step(a)
    assert env.observation_space.contains(o) and o.dtype == np.float32
    assert np.isscalar(r) and isinstance(te, bool) and isinstance(tr, bool)
    assert set(inf) == {"time", "TimeLimit.truncated"}
    rod = oracle_built.OracleRod(env._vec.cfg)
    rod.reset_pendulum(_theta(0))
    oo, rr, _, _ = rod.env_step(a[0])
    np.testing.assert_allclose(o, oo, rtol=RTOL, atol=1e-7)
    assert r == pytest.approx(rr, rel=RTOL)
    assert inf["time"] == rod.time
    # _prev_action survives reset (soft_pendulum.py:97-99): obs[2] of the next reset
    ob2, _ = env.reset(seed=0)
    assert ob2[2] == np.float32(a[0]) and ob2[3] == ob[3]
    env.close()


def test_truncation_fires_on_step_126(torch_gpu, hip_lib):
    env = _make(4, 1)
    env.reset(seed=3)
    zero = np.zeros(4, np.float32)
    fired = None
    for k in range(1, 128):
        _, _, term, trunc, info = env.step(zero)
        if trunc.any().item() and fired is None:
            fired = k
            assert trunc.all().item() and info["TimeLimit.truncated"].all()
    assert fired == 126  # SURVEY.md App. B; soft_pendulum.py:226-229
    env.close()


def test_nan_state_terminates_with_penalty(torch_gpu, hip_lib):
    env = _make(3, 1)
    env.reset(seed=0)
    st = env.backend.state()
    st["velocity"][0, 1, 7] = float("nan")  # poison env 1, node 7
    obs, rew, term, trunc, _ = env.step(np.zeros(3, np.float32))
    term, rew = term.cpu().numpy(), rew.cpu().numpy()
    assert term.tolist() == [False, True, False]
    assert rew[1] == -50.0 and rew[0] >= 0.0   # soft_pendulum.py:205-208,231
    # a dead env is not integrated any more: its state is NaN throughout (what the substeps
    # would smear it to), its clock keeps running, its neighbours are untouched
    sn = env.backend.state_numpy()
    assert np.isnan(sn["x"][1]).all() and np.isfinite(sn["x"][0]).all() and np.isfinite(sn["x"][2]).all()
    assert sn["time"][1] == sn["time"][0]
    obs, rew, term, trunc, _ = env.step(np.zeros(3, np.float32))
    assert term.cpu().numpy().tolist() == [False, True, False] and float(rew[1]) == -50.0
    env.close()


def test_masked_reset_leaves_other_envs_untouched(torch_gpu, hip_lib):
    env = _make(4, 1)
    env.reset(seed=0)
    env.step(np.full(4, 5.0, np.float32))
    before = env.backend.state_numpy()
    mask = np.array([False, True, False, True])
    obs, _ = env.reset(mask=mask)
    after = env.backend.state_numpy()
    np.testing.assert_array_equal(before["x"][~mask], after["x"][~mask])
    assert np.all(after["v"][mask] == 0.0) and np.all(after["time"][mask] == 0.0)
    assert np.all(after["time"][~mask] > 0.0)
    env.close()


def test_full_size_batch_properties(torch_gpu, hip_lib, oracle_built):
    # BASELINE config 2: 4096 envs x 50 elements.  Size-independent properties + spot
    # parity of a few rods against the oracle.
    n = 4096
    env = _make(n, 1)
    env.reset(seed=0)
    acts = np.random.default_rng(1).uniform(-22, 22, (2, n)).astype(np.float32)
    for t in range(2):
        obs, rew, term, trunc, _ = env.step(acts[t])
    obs, rew = obs.cpu().numpy(), rew.cpu().numpy()
    assert np.isfinite(obs).all() and not term.any().item() and not trunc.any().item()
    np.testing.assert_array_equal(obs[:, 2], acts[1])
    np.testing.assert_allclose(rew, 10 * np.abs(obs[:, 0].astype(np.float64)) + obs[:, 3].astype(np.float64) ** 2, rtol=1e-5, atol=1e-6)
    st = env.backend.state_numpy()
    Q = st["Q"]
    QQt = np.einsum("eimk,ejmk->eijk", Q, Q)
    assert np.abs(QQt - np.eye(3)[None, :, :, None]).max() < 1e-11   # K5
    assert np.abs(st["x"][:, 1:, 0]).max() == 0.0                      # BC: y0 = z0 = 0
    assert np.abs(st["x"][:, 2, :]).max() < 1e-12                      # planar motion
    for i in (0, 1, 777, 4095):
        r = oracle_built.OracleRod(env.cfg)
        r.reset_pendulum(_theta(i))
        for t in range(2):
            o, rw, _, _ = r.env_step(acts[t, i])
        np.testing.assert_allclose(obs[i], o, rtol=RTOL, atol=1e-7)
        np.testing.assert_allclose(st["x"][i], r.get("x"), rtol=RTOL, atol=1e-8)
    env.close()


def test_known_answer_cantilever_on_gpu(torch_gpu, hip_lib):
    # K2 on the HIP path itself (FIXED_BC + TIP_FORCE features), discrete-chain formula
    import torch

    from gym_softrobot_amd import _capi
    from gym_softrobot_amd.backend import HipRodBackend

    n_el, F = 40, 0.02
    cfg = _capi.softpendulum_config(2, n_elems=n_el, math_mode=1)
    cfg.dt = 2e-4
    cfg.features = _capi.FEAT_FIXED_BC | _capi.FEAT_TIP_FORCE | _capi.FEAT_ANALYTICAL_DAMPER
    cfg.damping_constant = 0.8
    cfg.tip_force[1] = F
    be = HipRodBackend(cfg, 0)
    be.reset_straight([0, 0, 0], [1, 0, 0], [0, 0, 1])
    be.substeps(None, 50000)
    st = be.state_numpy()
    E, G, r, L = 1e6, 1e6 / 3.0, 0.05, 1.0
    A = np.pi * r * r
    I = A * A / (4 * np.pi)
    expect = F * L**3 / (3 * E * I) * (1 - 1 / n_el) * (1 - 1 / (2 * n_el)) + F * L / (27.0 / 28.0 * G * A)
    assert np.abs(st["v"]).max() < 1e-8
    assert st["x"][0, 1, -1] == pytest.approx(expect, rel=2e-4)
    assert st["x"][1, 1, -1] == st["x"][0, 1, -1]
    be.close()


# ---- range-reduction paths of the fast kernel (softrod_fast.hpp) -------------------------
def _backend(cfg):
    from gym_softrobot_amd.backend import HipRodBackend

    return HipRodBackend(cfg, 0)


def _inject(be, name, arr):
    """arr: (comps, k) for every rod -> resident SoA rows."""
    import torch

    st = be.state()
    t = torch.from_numpy(np.ascontiguousarray(arr)).to(st[name].device)
    st[name][:, :, : arr.shape[1]] = t[:, None, :]


@pytest.mark.parametrize("E,F,lo,hi", [(1e6, 20.0, 0.1, 0.4), (3e5, 20.0, 0.4, 0.79), (1e5, 40.0, 0.79, 2.0)],
                         ids=["12-term", "20-term", "halving"])
def test_fast_kernel_large_bending_angles(torch_gpu, hip_lib, oracle_built, E, F, lo, hi):
    # neighbouring elements > 0.1 / 0.4 / 0.79 rad apart -> the longer series of theta_over_sin
    # and, beyond them, its half-angle recursion
    from gym_softrobot_amd import _capi

    n_el = 10
    cfg = _capi.softpendulum_config(2, n_elems=n_el, math_mode=1)
    cfg.features = _capi.FEAT_FIXED_BC | _capi.FEAT_TIP_FORCE | _capi.FEAT_ANALYTICAL_DAMPER
    cfg.damping_constant = 0.8
    cfg.dt = 2e-4
    cfg.tip_force[1] = F
    cfg.youngs_modulus = E
    cfg.shear_modulus = E / 3.0
    be = _backend(cfg)
    be.reset_straight([0, 0, 0], [1, 0, 0], [0, 0, 1])
    rod = oracle_built.OracleRod(cfg)
    rod.reset_straight([0, 0, 0], [1, 0, 0], [0, 0, 1])
    be.substeps(None, 20000)
    rod.substeps(0.0, 20000)
    st = be.state_numpy()
    rod.refresh_strains()
    ang = np.abs(rod.get("kappa")).max() * (1.0 / n_el)
    assert lo < ang < hi, ang                   # the intended branch really ran
    np.testing.assert_allclose(st["x"][0], rod.get("x"), rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(st["Q"][1], rod.get("Q"), rtol=1e-7, atol=1e-9)
    be.close()


def test_planar_and_general_paths_of_softpendulum(torch_gpu, hip_lib, oracle_built):
    """SoftPendulum-v0 states are planar and take the 2-D substep (softrod_planar.hpp); a
    state that is not — here: an out-of-plane velocity written through the state view —
    must take the general 3-D loop.  Both against the oracle, and against each other when
    the perturbation is far below the tolerance."""
    import gym_softrobot_amd as gsa

    n = 3
    env = gsa.make_vec("SoftPendulum-v0", n, device=0)
    env.reset(seed=0)
    rods = []
    for i in range(n):
        r = oracle_built.OracleRod(env.cfg)
        r.reset_pendulum(_theta(i))
        rods.append(r)
    st = env.backend.state()
    # env 0: planar.  env 1: v_z = 0.3 m/s on nodes 10..29.  env 2: v_z = 1e-200 (3-D loop,
    # numerically the planar trajectory)
    for i, amp in ((1, 0.3), (2, 1e-200)):
        v = rods[i].get("v")
        v[2, 10:30] = amp
        rods[i].set("v", v)
        st["velocity"][2, i, 10:30] = amp
    acts = np.random.default_rng(4).uniform(-22, 22, (3, n)).astype(np.float32)
    for t in range(3):
        obs, rew, term, trunc, _ = env.step(acts[t])
        obs, rew = obs.cpu().numpy(), rew.cpu().numpy()
        for i, r in enumerate(rods):
            o, rw, te, tr = r.env_step(acts[t, i])
            np.testing.assert_allclose(obs[i], o, rtol=RTOL, atol=1e-7)
            np.testing.assert_allclose(rew[i], rw, rtol=RTOL, atol=1e-9)
    sn = env.backend.state_numpy()
    for i, r in enumerate(rods):
        np.testing.assert_allclose(sn["x"][i], r.get("x"), rtol=RTOL, atol=1e-8)
        np.testing.assert_allclose(sn["Q"][i], r.get("Q"), rtol=RTOL, atol=1e-8)
        np.testing.assert_allclose(sn["w"][i], r.get("w"), rtol=RTOL, atol=1e-5)
    assert np.all(sn["x"][0][2] == 0.0) and np.all(sn["v"][0][2] == 0.0)      # stayed exactly planar
    assert np.abs(sn["x"][1][2]).max() > 1e-3                                    # really left the plane
    # env 2 ran the 3-D loop on (numerically) the planar problem of a clone of itself
    env2 = gsa.make_vec("SoftPendulum-v0", n, device=0)
    env2.reset(seed=0)
    for t in range(3):
        env2.step(acts[t])
    sp = env2.backend.state_numpy()
    np.testing.assert_allclose(sn["x"][2][:2], sp["x"][2][:2], rtol=1e-9, atol=1e-11)
    np.testing.assert_array_equal(sn["x"][0], sp["x"][0])
    env.close()
    env2.close()


def test_fast_kernel_large_rotation_rates(torch_gpu, hip_lib, oracle_built):
    # |omega| dt > 0.03 rad -> angle halving + double-angle rebuild in sinc_cosc
    from gym_softrobot_amd import _capi

    n_el = 12
    cfg = _capi.softpendulum_config(1, n_elems=n_el, math_mode=1)
    cfg.features = 0
    cfg.damping_constant = 0.0
    be = _backend(cfg)
    be.reset_straight([0, 0, 0], [1, 0, 0], [0, 0, 1])
    rod = oracle_built.OracleRod(cfg)
    rod.reset_straight([0, 0, 0], [1, 0, 0], [0, 0, 1])
    w = np.zeros((3, n_el))
    w[2] = 900.0            # fast spin about the rod axis: 0.09 rad per substep
    w[0] = 40.0 * np.sin(np.linspace(0, np.pi, n_el))
    rod.set("w", w)
    _inject(be, "omega", w)
    be.substeps(None, 40)
    rod.substeps(0.0, 40)
    st = be.state_numpy()
    np.testing.assert_allclose(st["Q"][0], rod.get("Q"), rtol=0, atol=1e-9)
    np.testing.assert_allclose(st["w"][0], rod.get("w"), rtol=1e-8, atol=1e-7)
    np.testing.assert_allclose(st["x"][0], rod.get("x"), rtol=0, atol=1e-11)
    be.close()


def test_fast_kernel_strong_damping_and_free_features(torch_gpu, hip_lib, oracle_built):
    # |e * log c_r| > 1e-3 -> halving + squaring in exp_pair; also a feature set without BC
    from gym_softrobot_amd import _capi

    n_el = 16
    cfg = _capi.softpendulum_config(1, n_elems=n_el, math_mode=1)
    cfg.features = _capi.FEAT_GRAVITY | _capi.FEAT_ANALYTICAL_DAMPER
    cfg.damping_constant = 3.0
    be = _backend(cfg)
    d, nrm = [np.cos(0.4), np.sin(0.4), 0.0], [np.sin(0.4), -np.cos(0.4), 0.0]
    be.reset_straight([0.1, 0.2, 0.3], d, nrm)
    rod = oracle_built.OracleRod(cfg)
    rod.reset_straight([0.1, 0.2, 0.3], d, nrm)
    w = np.zeros((3, n_el))
    w[1] = 3.0 * np.cos(np.linspace(0, 3, n_el))
    w[0] = 1.0
    rod.set("w", w)
    _inject(be, "omega", w)
    be.substeps(None, 500)
    rod.substeps(0.0, 500)
    st = be.state_numpy()
    np.testing.assert_allclose(st["x"][0], rod.get("x"), rtol=1e-9, atol=1e-11)
    np.testing.assert_allclose(st["v"][0], rod.get("v"), rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(st["w"][0], rod.get("w"), rtol=1e-7, atol=1e-10)
    np.testing.assert_allclose(st["Q"][0], rod.get("Q"), rtol=0, atol=1e-10)
    be.close()


def test_odd_rod_sizes_and_substep_counts(torch_gpu, hip_lib, oracle_built):
    # edge sizes: the smallest rod, the largest one wavefront holds (63 elements), and
    # substep counts 0 / 1 (first and last kinematic half-steps coincide)
    from gym_softrobot_amd import _capi

    for n_el, nsub in ((2, 7), (63, 30), (50, 1), (50, 0)):
        for mode in (0, 1):
            cfg = _capi.softpendulum_config(2, n_elems=n_el, math_mode=mode)
            be = _backend(cfg)
            th = np.deg2rad([91.0, 88.5])
            be.reset(th)
            be.substeps(np.array([4.0, -9.0], np.float32), nsub)
            st = be.state_numpy()
            for i in range(2):
                rod = oracle_built.OracleRod(cfg)
                rod.reset_pendulum(th[i])
                rod.substeps([4.0, -9.0][i], nsub)
                np.testing.assert_allclose(st["x"][i], rod.get("x"), rtol=1e-9, atol=1e-12)
                # velocities far ahead of the elastic wave front are ~1e-10 of the scale
                np.testing.assert_allclose(st["v"][i], rod.get("v"), rtol=1e-7, atol=1e-10)
                np.testing.assert_allclose(st["Q"][i], rod.get("Q"), rtol=0, atol=1e-11)
                assert st["time"][i] == rod.time
            be.close()


# ---- SoftPendulum3D-v0 (moving base, Laplace filter, strong damper) --------------------------
def _tilt(seed):
    from gym_softrobot_amd.envs.soft_pendulum_3d import initial_tilt
    from gym_softrobot_amd.seeding import np_random

    return initial_tilt(np_random(seed)[0])


@pytest.mark.parametrize("math_mode", [0, 1], ids=["libm", "fast"])
def test_softpendulum3d_rollout_matches_oracle(torch_gpu, hip_lib, oracle_built, math_mode):
    import gym_softrobot_amd as gsa

    n, T = 8, 5
    env = gsa.make_vec("SoftPendulum3D-v0", n, device=0, math_mode=math_mode)
    obs0, _ = env.reset(seed=0)
    obs0 = obs0.cpu().numpy().copy()
    acts = np.random.default_rng(2).uniform(-1, 1, (T, n, 2)).astype(np.float32)
    rods = []
    for i in range(n):
        r = oracle_built.OracleRod(env.cfg)
        r.reset_pendulum3d(_tilt(i))
        np.testing.assert_allclose(obs0[i], r.observe3d(), rtol=1e-6, atol=1e-9)
        rods.append(r)
    for t in range(T):
        obs, rew, term, trunc, info = env.step(acts[t])
        obs, rew = obs.cpu().numpy(), rew.cpu().numpy()
        tilt = info["tilt"].cpu().numpy()
        for i, r in enumerate(rods):
            o, rw, te, tr, tl = r.env_step3d(acts[t, i])
            np.testing.assert_allclose(obs[i], o, rtol=RTOL, atol=1e-7)
            np.testing.assert_allclose(rew[i], rw, rtol=RTOL, atol=1e-9)
            np.testing.assert_allclose(tilt[i], tl, rtol=RTOL, atol=1e-9)
            assert bool(term[i]) == te and bool(trunc[i]) == tr
    st = env.backend.state_numpy()
    for i, r in enumerate(rods):
        np.testing.assert_allclose(st["x"][i], r.get("x"), rtol=RTOL, atol=1e-8)
        np.testing.assert_allclose(st["v"][i], r.get("v"), rtol=RTOL, atol=1e-6)
        np.testing.assert_allclose(st["Q"][i], r.get("Q"), rtol=RTOL, atol=1e-8)
        np.testing.assert_allclose(st["w"][i], r.get("w"), rtol=RTOL, atol=1e-5)
        np.testing.assert_allclose(st["control"][i], r.get("control"), rtol=1e-12, atol=0)
        assert st["time"][i] == r.time
    env.close()


@pytest.mark.parametrize("n_elems", [6, 8, 63, 100], ids=lambda n: f"n{n}")
def test_softpendulum3d_filter_paths_by_rod_size(torch_gpu, hip_lib, oracle_built, n_elems):
    """The order-7 Laplace filter runs as DPP passes for rods shorter than 8 elements, as the
    LDS stencil otherwise (one or two slots per lane); all against the oracle's seven passes."""
    import gym_softrobot_amd as gsa

    n, T = 4, 3
    env = gsa.make_vec("SoftPendulum3D-v0", n, device=0, n_elems=n_elems)
    env.reset(seed=0)
    acts = np.random.default_rng(6).uniform(-1, 1, (T, n, 2)).astype(np.float32)
    rods = []
    for i in range(n):
        r = oracle_built.OracleRod(env.cfg)
        r.reset_pendulum3d(_tilt(i))
        rods.append(r)
    for t in range(T):
        obs, rew, term, trunc, info = env.step(acts[t])
        obs, rew = obs.cpu().numpy(), rew.cpu().numpy()
        for i, r in enumerate(rods):
            o, rw, te, tr, tl = r.env_step3d(acts[t, i])
            np.testing.assert_allclose(obs[i], o, rtol=RTOL, atol=1e-7)
            np.testing.assert_allclose(rew[i], rw, rtol=RTOL, atol=1e-9)
    st = env.backend.state_numpy()
    for i, r in enumerate(rods):
        np.testing.assert_allclose(st["v"][i], r.get("v"), rtol=RTOL, atol=1e-7)
        np.testing.assert_allclose(st["w"][i], r.get("w"), rtol=RTOL, atol=1e-6)
    env.close()


@pytest.mark.parametrize("math_mode", [0, 1], ids=["libm", "fast"])
@pytest.mark.parametrize("which", ["damp_before_constrain", "contact_before_forcing", "single_time_add"])
def test_operator_order_switches_match_oracle(torch_gpu, hip_lib, oracle_built, which, math_mode):
    """The softrod_config switches that exist so that a session with pyelastica importable can
    re-pin the operator order must mean the same thing on both sides."""
    from gym_softrobot_amd import _capi

    n = 3
    if which == "contact_before_forcing":
        cfg = _capi.arm_single_config(n, math_mode=math_mode)
        cfg.contact_before_forcing = 1
        cfg.n_substeps = 200
    else:
        cfg = _capi.softpendulum3d_config(n, math_mode=math_mode)
        cfg.n_substeps = 120
        if which == "damp_before_constrain":
            cfg.damp_before_constrain = 1
        else:
            cfg.time_two_half_adds = 0
    be = _backend(cfg)
    rods = [oracle_built.OracleRod(cfg) for _ in range(n)]
    if which == "contact_before_forcing":
        be.reset_straight(np.zeros((n, 3)), np.tile([1.0, 0, 0], (n, 1)), np.tile([0, 0, 1.0], (n, 1)))
        for r in rods:
            r.reset_arm()
        acts = np.random.default_rng(3).uniform(-6, 6, (2, n, 7)).astype(np.float32)
    else:
        tilts = [_tilt(i) for i in range(n)]
        d = np.array([[np.sin(t), 0.0, np.cos(t)] for t in tilts])
        be.reset_straight(np.zeros((n, 3)), d, np.tile([0, 1.0, 0], (n, 1)))
        for r, t in zip(rods, tilts):
            r.reset_pendulum3d(t)
        acts = np.random.default_rng(3).uniform(-1, 1, (2, n, 2)).astype(np.float32)
    be.observe(None)
    for t in range(2):
        obs, rew, term, trunc = (x.cpu().numpy() for x in be.step(acts[t]))
        for i, r in enumerate(rods):
            if which == "contact_before_forcing":
                o, rw, te, tr = r.env_step_arm(acts[t, i])
            else:
                o, rw, te, tr, _ = r.env_step3d(acts[t, i])
            np.testing.assert_allclose(obs[i], o, rtol=RTOL, atol=2e-6)
            np.testing.assert_allclose(rew[i], rw, rtol=RTOL, atol=1e-7)
    st = be.state_numpy()
    for i, r in enumerate(rods):
        np.testing.assert_allclose(st["x"][i], r.get("x"), rtol=RTOL, atol=1e-8)
        assert st["time"][i] == r.time
    be.close()


def test_softpendulum3d_single_env_and_truncation(torch_gpu, hip_lib):
    import gym_softrobot_amd as gsa

    env = gsa.make("SoftPendulum3D-v0")
    ob, info = env.reset(seed=0)
    assert env.observation_space.contains(ob) and ob.dtype == np.float32 and info == {}
    with pytest.raises(ValueError):
        env.step(np.array([2.0, 0.0], np.float32))
    fired = None
    a = np.array([0.2, -0.1], np.float32)
    for k in range(1, 128):
        ob, r, te, tr, inf = env.step(a)
        assert isinstance(r, float) and isinstance(te, bool) and isinstance(tr, bool)
        assert not te
        if tr and fired is None:
            fired = k
    assert fired == 126  # time >= 5.0 on the float64 accumulated by half-steps (t_125 < 5)
    # base walked to the limit and stayed: 0.2e-3 * 125 = 0.025 < 0.5 -> not clipped yet
    assert ob[0] == pytest.approx(127 * float(np.float32(1e-3) * np.float32(0.2)), rel=1e-5)
    env.close()


def test_softpendulum3d_determinism_and_masked_reset(torch_gpu, hip_lib):
    import gym_softrobot_amd as gsa

    acts = np.random.default_rng(4).uniform(-1, 1, (3, 6, 2)).astype(np.float32)
    outs = []
    for _ in range(2):
        env = gsa.make_vec("SoftPendulum3D-v0", 6, device=0)
        env.reset(seed=9)
        for t in range(3):
            o, r, *_ = env.step(acts[t])
        outs.append((o.cpu().numpy().copy(), r.cpu().numpy().copy(), env.backend.state_numpy()))
        if len(outs) == 2:
            mask = np.array([True, False, False, True, False, False])
            o2, _ = env.reset(mask=mask)
            st = env.backend.state_numpy()
            assert np.all(st["control"][mask] == 0.0) and np.all(st["control"][~mask, :2] != 0.0)
            assert np.all(o2.cpu().numpy()[mask, 6:8] == 0.0)   # _prev_action cleared on reset
            np.testing.assert_array_equal(o2.cpu().numpy()[~mask, 6:8], acts[2][~mask])
        env.close()
    np.testing.assert_array_equal(outs[0][0], outs[1][0])
    np.testing.assert_array_equal(outs[0][1], outs[1][1])
    np.testing.assert_array_equal(outs[0][2]["x"], outs[1][2]["x"])


# ---- OctoArmSingle-v0 (plane contact + anisotropic friction + rest-curvature actuation) -----
@pytest.mark.parametrize("math_mode", [0, 1], ids=["libm", "fast"])
def test_arm_single_rollout_matches_oracle(torch_gpu, hip_lib, oracle_built, math_mode):
    import gym_softrobot_amd as gsa

    n, T = 6, 4
    env = gsa.make_vec("OctoArmSingle-v0", n, device=0, math_mode=math_mode)
    obs0, _ = env.reset()
    obs0 = obs0.cpu().numpy().copy()
    # gentle (|a| <= 6) and aggressive (|a| <= 22) curvature commands
    rng = np.random.default_rng(11)
    acts = rng.uniform(-1, 1, (T, n, 7)).astype(np.float32)
    acts[:, : n // 2] *= 6.0
    acts[:, n // 2 :] *= 22.0
    rods = []
    for i in range(n):
        r = oracle_built.OracleRod(env.cfg)
        o = r.reset_arm()
        np.testing.assert_allclose(obs0[i], o, rtol=1e-6, atol=1e-7)
        rods.append(r)
    for t in range(T):
        obs, rew, term, trunc, info = env.step(acts[t])
        obs, rew = obs.cpu().numpy(), rew.cpu().numpy()
        for i, r in enumerate(rods):
            o, rw, te, tr = r.env_step_arm(acts[t, i])
            np.testing.assert_allclose(obs[i], o, rtol=RTOL, atol=2e-6)
            np.testing.assert_allclose(rew[i], rw, rtol=RTOL, atol=1e-7)
            assert bool(term[i]) == te and bool(trunc[i]) == tr
    st = env.backend.state_numpy()
    for i, r in enumerate(rods):
        np.testing.assert_allclose(st["x"][i], r.get("x"), rtol=RTOL, atol=1e-7)
        np.testing.assert_allclose(st["Q"][i], r.get("Q"), rtol=RTOL, atol=1e-6)
        np.testing.assert_allclose(st["rest_kappa"][i][0], r.get("rest_kappa")[0], rtol=1e-12, atol=1e-12)
        assert st["time"][i] == r.time
    env.close()


def test_arm_single_rest_and_api(torch_gpu, hip_lib):
    import gym_softrobot_amd as gsa

    env = gsa.make("OctoArmSingle-v0")
    ob, info = env.reset(seed=0)
    assert ob.shape == (25,) and ob.dtype == np.float32 and info == {}
    # zero action: the arm rests on the plane, nothing moves (contact response cancels weight)
    for _ in range(2):
        o, r, te, tr, inf = env.step(np.zeros(7, np.float32))
    assert isinstance(r, float) and not te and not tr
    assert r == pytest.approx(np.exp(-0.825 / 0.35) - 0.096, rel=1e-9)
    st = env._vec.backend.state_numpy()
    assert np.abs(st["v"]).max() < 1e-10 and np.abs(st["x"][0, 2]).max() < 1e-12
    np.testing.assert_allclose(o[14:16], 0.0, atol=1e-12)
    assert inf["time"] == pytest.approx(2 * 714 * 7e-5, rel=1e-9)
    env.close()


def test_autoreset_on_gpu(torch_gpu, hip_lib):
    # NEXT_STEP auto-reset through the HIP path: truncation after 3 short steps, reset on the 4th
    import gym_softrobot_amd as gsa

    kw = dict(final_time=5e-4, time_step=1e-4, recording_fps=5000, n_elems=20)
    env = gsa.make_vec("SoftPendulum-v0", 5, device=0, autoreset=True, **kw)
    obs0, _ = env.reset(seed=0)
    obs0 = obs0.cpu().numpy().copy()
    a = np.linspace(-5, 5, 5).astype(np.float32)
    for k in range(1, 5):
        obs, rew, term, trunc, _ = env.step(a)
        if k == 3:
            assert trunc.all().item()
        if k == 4:
            assert not trunc.any().item() and (rew == 0).all().item()
            o = obs.cpu().numpy()
            np.testing.assert_array_equal(o[:, :2], 0.0)
            assert not np.allclose(o[:, 3], obs0[:, 3])   # a new initial angle was drawn
            st = env.backend.state_numpy()
            assert np.all(st["time"] == 0.0) and np.all(st["v"] == 0.0)
    env.close()


@pytest.mark.parametrize("env_id,kw,amax", [
    ("SoftPendulum-v0", dict(final_time=5e-4, time_step=1e-4, recording_fps=5000, n_elems=20), 5.0),
    ("SoftPendulum-v0", dict(final_time=7e-4, time_step=1e-4, recording_fps=5000, n_elems=100), 5.0),
    ("SoftPendulum3D-v0", dict(final_time=6e-4, time_step=1e-4, recording_fps=5000, n_elems=20), 1.0),
    ("OctoArmSingle-v0", dict(final_time=3e-4, time_step=7e-5, recording_fps=7000, n_elems=20), 5.0),
    ("OctoArmSingle-v0", dict(final_time=3e-4, time_step=7e-5, recording_fps=7000, n_elems=100), 5.0),
    ("OctoFlat-v0", dict(final_time=3e-4, time_step=7e-5, recording_fps=7000), 20.0),
], ids=["pendulum", "pendulum-2-per-lane", "pendulum3d", "arm", "arm-two-windows", "octo"])
def test_device_autoreset_equals_host_autoreset(torch_gpu, hip_lib, env_id, kw, amax):
    """autoreset="device" (staged reset records, no host read) must reproduce autoreset=True
    (host reads the flags and resets) bit for bit: same NEXT_STEP semantics, and the staged
    records are the draws the host path would have taken from each env's stream."""
    import gym_softrobot_amd as gsa

    n, T = 6, 23           # > queue_depth steps: the queue is topped up on the way
    from gym_softrobot_amd.distributed import unpack_outputs

    host = gsa.make_vec(env_id, n, device=0, autoreset=True, **kw)
    dev = gsa.make_vec(env_id, n, device=0, autoreset="device", **kw)
    devp = gsa.make_vec(env_id, n, device=0, autoreset="device", **kw)   # packed rows (multi-GPU path)
    oh, _ = host.reset(seed=3)
    od, _ = dev.reset(seed=3)
    devp.reset(seed=3)
    assert torch_gpu.equal(oh, od)
    acts = np.random.default_rng(0).uniform(-amax, amax, (T, n, host.action_dim)).astype(np.float32)
    resets = 0
    for t in range(T):
        a = acts[t].copy()
        if env_id == "SoftPendulum3D-v0":
            a = np.clip(a, -1, 1)
        o1, r1, te1, tr1, i1 = host.step(a)
        o2, r2, te2, tr2, i2 = dev.step(a)
        o3, r3, te3, tr3 = unpack_outputs(devp.step_packed(a)[0], dev.obs_dim)
        assert torch_gpu.equal(o2, o3) and torch_gpu.equal(r2, r3) and torch_gpu.equal(tr2, tr3), t
        assert torch_gpu.equal(o1, o2), t
        assert torch_gpu.equal(r1, r2) and torch_gpu.equal(te1, te2) and torch_gpu.equal(tr1, tr2), t
        np.testing.assert_array_equal(i1["time"], i2["time"].cpu().numpy())
        resets += int((host._steps == 0).sum())
    assert resets >= 2 * n, "episodes were meant to end and restart several times"
    cons, underflow = dev.backend.queue_status()
    assert underflow == 0 and cons.min() >= 2
    # a masked manual reset of a finished env clears its pending auto-reset and re-bases its queue
    m = np.zeros(n, bool)
    m[1] = True
    oh, _ = host.reset(mask=m)
    od, _ = dev.reset(mask=m)
    assert torch_gpu.equal(oh, od)
    for t in range(4):
        o1, r1, te1, tr1, _ = host.step(acts[t])
        o2, r2, te2, tr2, _ = dev.step(acts[t])
        assert torch_gpu.equal(o1, o2) and torch_gpu.equal(r1, r2) and torch_gpu.equal(tr1, tr2)
    host.close()
    dev.close()
    devp.close()


def test_queue_status_without_stalling_the_stream(torch_gpu, hip_lib):
    """softrod_queue_status_begin / _poll (C-ABI v11): the same counters as the blocking read, as of
    the begin; poll without a begin is refused; a compact push lands where the whole-ring upload did."""
    import gym_softrobot_amd as gsa
    from gym_softrobot_amd._capi import SoftrodError

    kw = dict(final_time=2e-4, time_step=1e-4, recording_fps=10000, n_elems=10)   # an episode is 3 steps + the resetting one
    env = gsa.make_vec("SoftPendulum-v0", 5, device=0, autoreset="device", **kw)
    env.top_up_paused = True                      # no top-up on the way: the counters are ours to read
    env.reset(seed=3)
    b = env.backend
    with pytest.raises(SoftrodError, match="no softrod_queue_status_begin"):
        b.queue_status_poll()
    a = np.zeros(5, np.float32)
    for _ in range(9):
        env.step(a)
    b.queue_status_begin()
    for _ in range(4):                             # work enqueued after the begin does not change what it reports
        env.step(a)
    st = b.queue_status_poll(wait=True)
    assert st is not None
    cons_at_begin, uf = st
    assert uf == 0 and (cons_at_begin == cons_at_begin[0]).all() and 1 <= cons_at_begin[0] <= 3
    cons_now, _ = b.queue_status()
    assert (cons_now >= cons_at_begin).all() and cons_now.sum() > cons_at_begin.sum()
    # a push of two records for env 1 only (compact path): consumed later, in order, by env 1's resets
    env._top_up()
    twin = gsa.make_vec("SoftPendulum-v0", 5, device=0, autoreset=True, **kw)
    twin.reset(seed=3)
    for _ in range(13):
        twin.step(a)
    for _ in range(12):
        o1 = env.step(a)[0].cpu().numpy()
        o2 = twin.step(a)[0].cpu().numpy()
        np.testing.assert_array_equal(o1, o2)
    env.close(); twin.close()


def test_device_autoreset_underflow_is_reported(torch_gpu, hip_lib):
    import gym_softrobot_amd as gsa
    from gym_softrobot_amd._capi import SoftrodError

    kw = dict(final_time=1e-4, time_step=1e-4, recording_fps=10000, n_elems=10)   # every step truncates
    env = gsa.make_vec("SoftPendulum-v0", 3, device=0, autoreset="device", **kw)
    env.top_up_paused = True         # never top up on the way
    env.reset(seed=0)
    a = np.zeros(3, np.float32)
    with pytest.raises(SoftrodError, match="no staged record"):
        for _ in range(4 * env.queue_depth):     # an episode here is 2 steps + the resetting one
            env.step(a)
        env._top_up()
    env.close()


def test_rod_recorder_on_gpu(torch_gpu, hip_lib, oracle_built):
    """Diagnostic taps (RodCallBack fields) from the resident state of chosen envs."""
    import gym_softrobot_amd as gsa

    n = 4
    env = gsa.make_vec("SoftPendulum-v0", n, device=0, config_generate_video=True)
    env.record_envs = (0, 2)
    env.reset(seed=0)
    rods = {}
    for i in env.record_envs:
        rods[i] = oracle_built.OracleRod(env.cfg)
        rods[i].reset_pendulum(_theta(i))
    acts = np.random.default_rng(2).uniform(-22, 22, (2, n)).astype(np.float32)
    for t in range(2):
        env.step(acts[t])
        for i, r in rods.items():
            r.env_step(acts[t, i])
    st = env.backend.state_numpy()
    for k, i in enumerate(env.record_envs):
        p = env.recorder.params[k]
        assert len(p["time"]) == 2 and p["time"][-1] == pytest.approx(0.08, rel=1e-9)
        np.testing.assert_array_equal(p["position"][-1], st["x"][i])
        np.testing.assert_array_equal(p["omega"][-1], st["w"][i])
        # the reference's RodCallBack copies the CACHED strain arrays (callback_func.py:31-41): those of the
        # last force evaluation, i.e. the mid-substep configuration — the oracle's caches as they are, no refresh
        np.testing.assert_allclose(p["kappa"][-1], rods[i].get("kappa"), rtol=RTOL, atol=1e-7)
        np.testing.assert_allclose(p["sigma"][-1], rods[i].get("sigma"), rtol=RTOL, atol=1e-8)
        np.testing.assert_allclose(p["radius"][-1], rods[i].get("radius"), rtol=1e-9)
        np.testing.assert_allclose(p["dilatation"][-1], rods[i].get("dilatation"), rtol=1e-9)
        # the rebuilt mid-substep configuration IS the one the kernel evaluated its forces at: its tangents
        # are the row the kernel cached there (softrod_state_view.tangents), to rounding
        np.testing.assert_allclose(env.recorder.last_mid_tangents[k], st["tangents"][i], rtol=0, atol=1e-13)
        # and the end-of-step strains are measurably different (the defect of round 5's taps)
        end = rods[i].get("sigma").copy()
        rods[i].refresh_strains()
        assert np.abs(rods[i].get("sigma") - end).max() > 1e-9
    env.close()


# ---- rods longer than one node per lane (two slots per lane, EPL = 2) -------------------------
def test_long_rod_softpendulum_matches_oracle(torch_gpu, hip_lib, oracle_built):
    import gym_softrobot_amd as gsa

    for n_el in (64, 100, 126):
        env = gsa.make_vec("SoftPendulum-v0", 3, device=0, n_elems=n_el)
        env.reset(seed=4)
        acts = np.random.default_rng(n_el).uniform(-22, 22, (3, 3)).astype(np.float32)
        rods = []
        for i in range(3):
            r = oracle_built.OracleRod(env.cfg)
            r.reset_pendulum(_theta(4 + i))
            rods.append(r)
        for t in range(3):
            obs, rew, term, trunc, _ = env.step(acts[t])
            obs, rew = obs.cpu().numpy(), rew.cpu().numpy()
            for i, r in enumerate(rods):
                o, rw, te, tr = r.env_step(acts[t, i])
                np.testing.assert_allclose(obs[i], o, rtol=RTOL, atol=1e-7)
                np.testing.assert_allclose(rew[i], rw, rtol=RTOL, atol=1e-9)
                assert bool(term[i]) == te and bool(trunc[i]) == tr
        st = env.backend.state_numpy()
        for i, r in enumerate(rods):
            np.testing.assert_allclose(st["x"][i], r.get("x"), rtol=RTOL, atol=1e-8)
            np.testing.assert_allclose(st["Q"][i], r.get("Q"), rtol=RTOL, atol=1e-8)
            np.testing.assert_allclose(st["w"][i], r.get("w"), rtol=RTOL, atol=1e-5)
        env.close()


def test_arm_single_100_elements_matches_oracle(torch_gpu, hip_lib, oracle_built):
    # BASELINE config 3: OctoArmSingle-style, 100 elements (7 np.array_split bins of the 99
    # curvatures instead of the reference's 7x7 reshape)
    import gym_softrobot_amd as gsa

    n, T = 4, 3
    env = gsa.make_vec("OctoArmSingle-v0", n, device=0, n_elems=100)
    obs0, _ = env.reset()
    obs0 = obs0.cpu().numpy().copy()
    rng = np.random.default_rng(5)
    acts = (rng.uniform(-1, 1, (T, n, 7)) * np.array([4.0, 8.0, 14.0, 22.0])[None, :, None]).astype(np.float32)
    rods = []
    for i in range(n):
        r = oracle_built.OracleRod(env.cfg)
        o = r.reset_arm()
        np.testing.assert_allclose(obs0[i], o, rtol=1e-6, atol=1e-7)
        rods.append(r)
    for t in range(T):
        obs, rew, term, trunc, _ = env.step(acts[t])
        obs, rew = obs.cpu().numpy(), rew.cpu().numpy()
        for i, r in enumerate(rods):
            o, rw, te, tr = r.env_step_arm(acts[t, i])
            np.testing.assert_allclose(obs[i], o, rtol=RTOL, atol=2e-6)
            np.testing.assert_allclose(rew[i], rw, rtol=RTOL, atol=1e-7)
            assert bool(term[i]) == te and bool(trunc[i]) == tr
    st = env.backend.state_numpy()
    for i, r in enumerate(rods):
        np.testing.assert_allclose(st["x"][i], r.get("x"), rtol=RTOL, atol=1e-7)
        np.testing.assert_allclose(st["rest_kappa"][i][0], r.get("rest_kappa")[0], rtol=1e-12, atol=1e-12)
    env.close()


@pytest.mark.parametrize("n_elems", [64, 80, 100, 102], ids=lambda n: f"n{n}")
def test_two_window_kernel_for_long_arms(torch_gpu, hip_lib, oracle_built, n_elems, monkeypatch):
    """OctoArmSingle rods of 64..102 elements run on two overlapping one-node-per-lane windows
    (softrod_window.hpp).  Against the oracle, and against the two-slots-per-lane kernel the same
    rods use when the window form is switched off: the owned halves must not see the halo.  The
    shipped form runs FOUR rods per workgroup with a rod's two windows on one SIMD and an LDS counter
    as the rendezvous; with SOFTROD_WINDOW_PAIRED=0 it is one rod per workgroup with s_barrier —
    the same arithmetic, so the two must agree bit for bit (3 rods: one slot of the workgroup idles)."""
    from gym_softrobot_amd import _capi
    from gym_softrobot_amd.backend import HipRodBackend

    n = 3
    acts = np.random.default_rng(n_elems).uniform(-6, 6, (2, n, 7)).astype(np.float32)
    cfg = _capi.arm_single_config(n, n_elems=n_elems)
    cfg.n_substeps = 300
    rods = [oracle_built.OracleRod(cfg) for _ in range(n)]
    for r in rods:
        r.reset_arm()
    outs = {}
    for name, env in (("window", {}), ("barrier", {"SOFTROD_WINDOW_PAIRED": "0"}), ("two-slot", {"SOFTROD_NO_WINDOW": "1"})):
        monkeypatch.setenv("SOFTROD_DEBUG_SWITCHES", "1")      # the A/B switches count only with this one
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        be = HipRodBackend(cfg, 0)
        assert ("window" in be.kernel_tier()) == (name != "two-slot"), be.kernel_tier()
        be.reset_straight(np.zeros((n, 3)), np.tile([1.0, 0, 0], (n, 1)), np.tile([0, 0, 1.0], (n, 1)))
        be.observe(None)
        res = []
        for t in range(2):
            o, r, te, tr = (x.cpu().numpy().copy() for x in be.step(acts[t]))
            res.append((o, r, te, tr))
        outs[name] = (res, be.state_numpy())
        be.close()
        for k in env:
            monkeypatch.delenv(k)
        monkeypatch.delenv("SOFTROD_DEBUG_SWITCHES")
    for t in range(2):
        for i, r in enumerate(rods):
            o, rw, te, tr = r.env_step_arm(acts[t, i])
            np.testing.assert_allclose(outs["window"][0][t][0][i], o, rtol=RTOL, atol=2e-6)
            np.testing.assert_allclose(outs["window"][0][t][1][i], rw, rtol=RTOL, atol=1e-7)
            assert bool(outs["window"][0][t][2][i]) == te and bool(outs["window"][0][t][3][i]) == tr
    sw, ss = outs["window"][1], outs["two-slot"][1]
    for k in ("x", "v", "w", "Q", "time", "kappa"):
        np.testing.assert_array_equal(sw[k], outs["barrier"][1][k], err_msg=f"paired vs barrier window form: {k}")
    for a, b in zip(outs["window"][0], outs["barrier"][0]):
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)
    for i, r in enumerate(rods):
        np.testing.assert_allclose(sw["x"][i], r.get("x"), rtol=RTOL, atol=1e-7)
        np.testing.assert_allclose(sw["Q"][i], r.get("Q"), rtol=RTOL, atol=1e-6)
    np.testing.assert_allclose(sw["x"], ss["x"], rtol=0, atol=1e-12)
    np.testing.assert_allclose(sw["w"], ss["w"], rtol=0, atol=1e-7)
    np.testing.assert_array_equal(sw["time"], ss["time"])
    for a, b in zip(outs["window"][0], outs["two-slot"][0]):
        np.testing.assert_array_equal(a[0], b[0])          # float32 observations: identical


@pytest.mark.parametrize("env_id,amax", [("SoftPendulum-v0", 22.0), ("SoftPendulum3D-v0", 1.0), ("OctoArmSingle-v0", 8.0)])
def test_packed_step_equals_separate_outputs(torch_gpu, hip_lib, env_id, amax):
    # the multi-GPU path lets the kernel write one packed row per env; unpacking is views only
    import gym_softrobot_amd as gsa
    from gym_softrobot_amd.distributed import packed_width, unpack_outputs

    n = 5
    a_env, b_env = gsa.make_vec(env_id, n, device=0), gsa.make_vec(env_id, n, device=0)
    a_env.reset(seed=3)
    b_env.reset(seed=3)
    acts = np.random.default_rng(8).uniform(-amax, amax, (2, n, a_env.action_dim)).astype(np.float32)
    for t in range(2):
        o, r, te, tr, _ = a_env.step(acts[t])
        packed, _ = b_env.step_packed(acts[t])
        assert packed.shape == (n, packed_width(a_env.obs_dim))
        po, pr, pte, ptr = unpack_outputs(packed, a_env.obs_dim)
        assert torch_gpu.equal(po, o) and torch_gpu.equal(pr, r)
        assert torch_gpu.equal(pte, te) and torch_gpu.equal(ptr, tr)
        assert pr.dtype == torch_gpu.float64 and pte.dtype == torch_gpu.bool
    a_env.close()
    b_env.close()


@pytest.mark.parametrize("env_id,kw,amax", [("SoftPendulum-v0", {}, 22.0), ("OctoArmSingle-v0", {}, 6.0),
                                            ("SoftArmTracking-v0", {"game_mode": 2}, 0.5),
                                            ("OctoFlat-v0", {"recording_fps": 71}, 10.0)])
def test_checkpoint_resume_is_exact(torch_gpu, hip_lib, env_id, kw, amax):
    """state_dict() / load_state_dict(): a batch put back from its snapshot continues bit for bit
    — also across a host auto-reset (the RNG streams are part of the snapshot) and into a FRESH
    env object."""
    import gym_softrobot_amd as gsa

    n = 3
    env = gsa.make_vec(env_id, n, numpy_output=True, autoreset=True, **kw)
    env.reset(seed=7)
    rng = np.random.default_rng(0)
    acts = rng.uniform(-amax, amax, (8, n, env.action_dim)).astype(np.float32)
    for t in range(3):
        env.step(acts[t])
    sd = env.state_dict()
    first = [env.step(acts[t])[:4] for t in range(3, 8)]
    other = gsa.make_vec(env_id, n, numpy_output=True, autoreset=True, **kw)
    other.reset(seed=99)                       # a different history, then overwritten by the snapshot
    other.step(acts[0])
    for e in (env, other):
        e.load_state_dict(sd)
        again = [e.step(acts[t])[:4] for t in range(3, 8)]
        for a, b in zip(first, again):
            for x, y in zip(a, b):
                np.testing.assert_array_equal(x, y)
    env.close()
    other.close()


def test_restore_refuses_a_snapshot_of_other_physics(torch_gpu, hip_lib):
    """A snapshot carries the fingerprint of the softrod_config (and ABI) it was taken under:
    same shapes but another dt / substep count / feature set must not load silently; and a
    handle with device-side auto-reset refuses snapshot/restore (its pending-reset flags and
    staged queue are outside the state view)."""
    import gym_softrobot_amd as gsa
    from gym_softrobot_amd import _capi

    a = gsa.make_vec("SoftPendulum-v0", 3)
    a.reset(seed=1)
    snap = a.backend.snapshot()
    a.backend.restore(snap)                                   # same config: fine
    b = gsa.make_vec("SoftPendulum-v0", 3, time_step=5e-5)    # same shapes, other physics
    b.reset(seed=1)
    with pytest.raises(ValueError, match="different softrod_config"):
        b.backend.restore(snap)
    c = gsa.make_vec("SoftPendulum-v0", 3, autoreset="device")
    c.reset(seed=1)
    with pytest.raises(_capi.SoftrodError, match="auto-reset"):
        c.backend.restore(snap)
    with pytest.raises(_capi.SoftrodError, match="auto-reset"):
        c.backend.snapshot()
    for e in (a, b, c):
        e.close()


def test_capi_calls_leave_the_current_device_alone(torch_gpu, hip_lib):
    """Every extern "C" entry point switches to the handle's device under a guard and restores
    the caller's: torch's current device is what it was after a reset / step / observe / destroy."""
    import gym_softrobot_amd as gsa

    before = torch_gpu.cuda.current_device()
    env = gsa.make_vec("SoftPendulum-v0", 2, device=0)
    env.reset(seed=0)
    env.step(np.zeros((2, 1), np.float32))
    env.backend.observe(None)
    env.close()
    assert torch_gpu.cuda.current_device() == before


@pytest.mark.parametrize("math_mode", [0, 1], ids=["libm", "fast"])
@pytest.mark.parametrize("mode", ["twist", "stretch", "twist+bend"])
def test_torsional_and_axial_modes_match_oracle(torch_gpu, hip_lib, oracle_built, mode, math_mode):
    """The channels no planar env excites: twist (G I3, J3, rotation about d3), stretch and their
    coupling with bending through the transport term — the rods of tests/test_oracle_physics.py
    k14 / k15 (exact discrete-chain frequencies), stepped by both kernels and compared with the
    oracle; `twist+bend` adds a transverse rate so that (J w) x w and kappa x B kappa are non-zero."""
    from gym_softrobot_amd import _capi

    n = 50
    cfg = _capi.softpendulum_config(2, n_elems=n, math_mode=math_mode)
    cfg.env_kind = _capi.ENV_NONE
    cfg.features = _capi.FEAT_FIXED_BC
    cfg.damping_constant = 0.0
    be = _backend(cfg)
    be.reset_straight([0, 0, 0], [1, 0, 0], [0, 0, 1])
    c1 = cfg.copy()
    c1.n_envs = 1
    rod = oracle_built.OracleRod(c1)
    rod.reset_straight([0, 0, 0], [1, 0, 0], [0, 0, 1])
    s_el = (np.arange(n) + 0.5) / n
    s_nd = np.arange(n + 1) / n
    w, v = np.zeros((3, n)), np.zeros((3, n + 1))
    if mode in ("twist", "twist+bend"):
        w[2] = 5.0 * np.sin(np.pi * s_el / 2)
    if mode == "stretch":
        v[0] = 0.05 * np.sin(np.pi * s_nd / 2)
    if mode == "twist+bend":
        w[0] = 2.0 * np.sin(np.pi * s_el)
        v[1] = 0.3 * s_nd ** 2
    w[:, 0] = 0.0
    v[:, 0] = 0.0
    rod.set("w", w)
    rod.set("v", v)
    _inject(be, "omega", w)
    _inject(be, "velocity", v)
    be.substeps(None, 600)
    rod.substeps(0.0, 600)
    torch_gpu.cuda.synchronize()
    st = be.state_numpy()
    for name in ("x", "v", "w", "Q"):
        np.testing.assert_allclose(st[name][0], rod.get(name), rtol=RTOL, atol=1e-9, err_msg=name)
        np.testing.assert_array_equal(st[name][0], st[name][1])
    assert np.abs(st["w"][0][2]).max() > 1e-2 or mode == "stretch"
    be.close()


def test_clock_table_and_its_fallback_are_bit_identical_to_the_accumulation(torch_gpu, hip_lib, oracle_built):
    """The planar loop carries no clock: a rod whose time IS the k-th entry of the host-accumulated
    table takes entry k + 1, any other clock (written through the state view, or beyond the table)
    is advanced by the 2 n_substeps additions after the loop.  Both must equal the oracle's
    `time += dt/2` chain bit for bit (the truncation flag is a strict comparison on it)."""
    import gym_softrobot_amd as gsa

    n = 6
    env = gsa.make_vec("SoftPendulum-v0", n)
    env.reset(seed=0)
    st = env.backend.state()
    start = np.array([0.0, 0.0, 0.12345678901234567, 40.96 + 1e-13, 3.9999999999, 1e6])
    acts = np.zeros((3, n, 1), np.float32)
    env.step(acts[0])                                     # everybody at table entry 1
    t1 = st["time"].cpu().numpy().copy()
    st["time"][2:] = torch_gpu.from_numpy(start[2:]).to(st["time"].device)      # off the table / beyond it
    rods = []
    for i in range(n):
        r = oracle_built.OracleRod(env.cfg)
        r.reset_pendulum(_theta(i))
        r.env_step(0.0)
        if i >= 2:
            r.set("time", [start[i]])
        rods.append(r)
    assert t1[0] == rods[0].time
    for t in (1, 2):
        env.step(acts[t])
        for r in rods:
            r.env_step(0.0)
        np.testing.assert_array_equal(st["time"].cpu().numpy(), np.array([r.time for r in rods]))
    env.close()


def test_clock_outside_the_general_loops_equals_the_accumulation(torch_gpu, hip_lib, oracle_built):
    """clock_after (general 3-D loop: OctoArmSingle; also with a substep count the table was not built
    for): table entry and after-the-loop additions both equal the oracle's clock bit for bit."""
    from gym_softrobot_amd import _capi
    from gym_softrobot_amd.backend import HipRodBackend

    n = 4
    cfg = _capi.arm_single_config(n)
    cfg.n_substeps = 40
    be = HipRodBackend(cfg, 0)
    be.reset_straight(np.zeros((n, 3)), np.tile([1.0, 0, 0], (n, 1)), np.tile([0, 0, 1.0], (n, 1)))
    be.observe(None)
    rods = [oracle_built.OracleRod(cfg) for _ in range(n)]
    for r in rods:
        r.reset_arm()
    st = be.state()
    a = np.zeros((n, 7), np.float32)
    be.step(a)
    for r in rods:
        r.env_step_arm(a[0])
    off = np.array([0.0, 0.0, 0.0123456789, 7.0e-5 * 40 * 2000 + 1e-9])          # on the table, off it, beyond it
    st["time"][2:] = torch_gpu.from_numpy(off[2:]).to(st["time"].device)
    for i in (2, 3):
        rods[i].set("time", [off[i]])
    for _ in range(3):
        be.step(a)
        for r in rods:
            r.env_step_arm(a[0])
        np.testing.assert_array_equal(st["time"].cpu().numpy(), np.array([r.time for r in rods]))
    be.substeps(None, 7)                 # not the substep count of the table: the additions run
    for r in rods:
        r.substeps(0.0, 7)
    np.testing.assert_array_equal(st["time"].cpu().numpy(), np.array([r.time for r in rods]))
    be.close()


def test_contact_kernels_ignore_non_finite_padding(torch_gpu, hip_lib):
    """the fast-math contact law keeps its masks as factors (zero normal force -> zero
    friction), and the slot past the last element feeds the tip node's average, so NaN / Inf
    written through the state view into the UNUSED slots of the rows must not reach the rod:
    OctoArmSingle (one rod per wave) and OctoFlat (ghost slots between the arms) step exactly
    as they do with clean padding."""
    import gym_softrobot_amd as gsa

    for env_id, n, amax in (("OctoArmSingle-v0", 3, 6.0), ("OctoFlat-v0", 2, 22.0)):
        clean = gsa.make_vec(env_id, n, device=0)
        dirty = gsa.make_vec(env_id, n, device=0)
        clean.reset(seed=3)
        dirty.reset(seed=3)
        st = dirty.backend.state()
        ne = int(dirty.cfg.n_elem)
        if env_id == "OctoFlat-v0":
            seg, na = int(st["arm_stride"]), int(dirty.cfg.n_arm)
            node_pad = [a * seg + k for a in range(na) for k in range(ne + 1, seg)]
            elem_pad = [a * seg + k for a in range(na) for k in range(ne, seg)]
        else:
            node_pad, elem_pad = list(range(ne + 1, 64)), list(range(ne, 64))
        bad = torch_gpu.tensor([float("nan"), float("inf"), -float("inf")], dtype=torch_gpu.float64, device=st["position"].device)
        for name, pad in (("position", node_pad), ("velocity", node_pad), ("omega", elem_pad), ("director", elem_pad)):
            idx = torch_gpu.tensor(pad, device=st[name].device)
            st[name][:, :, idx] = bad[torch_gpu.arange(len(pad), device=bad.device) % 3]
        acts = np.random.default_rng(0).uniform(-amax, amax, (2, n, clean.action_dim)).astype(np.float32)
        for t in range(2):
            a = clean.step(acts[t])
            b = dirty.step(acts[t])
            torch_gpu.cuda.synchronize()
            for x, y in zip(a[:4], b[:4]):
                assert torch_gpu.equal(x, y), (env_id, t)
            assert bool(torch_gpu.isfinite(a[0]).all())
        clean.close()
        dirty.close()


@pytest.mark.parametrize("math_mode", [0, 1], ids=["libm", "fast"])
@pytest.mark.parametrize("phi", [0.0, 0.7], ids=["planar", "turned"])
@pytest.mark.parametrize("alpha", [1.0, 10.0])
def test_known_answer_discrete_elastica_is_a_fixed_point_on_the_gpu(torch_gpu, hip_lib, math_mode, alpha, phi):
    """K16b on the HIP kernels themselves (no oracle involved): a rod placed in the directly solved
    large-deflection equilibrium of the discrete Cosserat rod (tests/elastica_chain.py; tip angles of
    26 and 82 degrees) stays there — after one substep without a damper its rates are 1e-9 of what
    the tip force alone would cause.  Checks sigma, S, Q^T n / e, kappa = -log(Q+ Q^T) / D at finite
    joint angles, the eps^3 on the couple and the shear couple's lever arm against an INDEPENDENT
    solution; and a settled run ends on it."""
    from gym_softrobot_amd import _capi
    from tests.elastica_chain import state

    n, E, r = 12, 1e7, 0.02
    A = np.pi * r * r
    I = A * A / (4 * np.pi)
    G, F = E / 3.0, alpha * E * I
    cfg = _capi.softpendulum_config(3, n_elems=n, math_mode=math_mode)
    cfg.env_kind = _capi.ENV_NONE
    cfg.features = _capi.FEAT_FIXED_BC | _capi.FEAT_TIP_FORCE
    cfg.base_radius, cfg.youngs_modulus, cfg.shear_modulus, cfg.damping_constant = r, E, G, 0.0
    cfg.tip_force[1], cfg.tip_force[2] = F * np.cos(phi), F * np.sin(phi)      # phi != 0: curvature on d1 AND d2
    be = _backend(cfg)
    be.reset_straight([0, 0, 0], [1, 0, 0], [0, 0, 1])
    _, x, Q = state(n, F, E * I, 27.0 / 28.0 * G * A, E * A, phi=phi)
    _inject(be, "position", x)
    _inject(be, "director", Q.reshape(9, n))
    be.substeps(None, 1)
    torch_gpu.cuda.synchronize()
    st = be.state_numpy()
    m_node = cfg.density * A * (1.0 / n)
    J1 = I * cfg.density * (1.0 / n)
    assert np.abs(st["v"]).max() < 1e-9 * cfg.dt * F / m_node
    assert np.abs(st["w"]).max() < 1e-9 * cfg.dt * F * (1.0 / n) / J1
    np.testing.assert_allclose(st["x"][0, :, 1:], x[:, 1:], rtol=1e-12, atol=1e-15)
    be.close()


@pytest.mark.parametrize("math_mode", [0, 1], ids=["libm", "fast"])
def test_known_answer_angular_momentum_of_a_tumbling_rod_on_the_gpu(torch_gpu, hip_lib, math_mode):
    """K17 on the HIP kernels themselves (no oracle involved): a free rod that tumbles, bends in two
    planes, twists, stretches and shears conserves its linear momentum to rounding and its total angular
    momentum sum m x cross v + sum Q^T (J omega / e) to the integrator's 2e-9 over 2000 substeps."""
    from gym_softrobot_amd import _capi
    from tests.test_oracle_physics import momenta, tumbling_rod_state

    n, r, rho = 16, 0.03, 1000.0
    cfg = _capi.softpendulum_config(2, n_elems=n, math_mode=math_mode)
    cfg.env_kind, cfg.features, cfg.dt, cfg.damping_constant = _capi.ENV_NONE, 0, 2e-5, 0.0
    cfg.base_radius, cfg.youngs_modulus, cfg.shear_modulus = r, 1e6, 1e6 / 3
    be = _backend(cfg)
    be.reset_straight([0, 0, 0], [1, 0, 0], [0, 0, 1])
    st = be.state_numpy()
    v, w = tumbling_rod_state(st["x"][0], st["Q"][0])
    _inject(be, "velocity", v)
    _inject(be, "omega", w)
    A = np.pi * r * r
    I1 = A * A / (4 * np.pi)
    l0 = 1.0 / n
    mass = np.full(n + 1, rho * A * l0)
    mass[[0, -1]] *= 0.5
    J = np.array([I1, I1, 2 * I1])[:, None] * rho * l0 * np.ones((1, n))

    def now():
        s = be.state_numpy()
        x = s["x"][0]
        dil = np.linalg.norm(x[:, 1:] - x[:, :-1], axis=0) / l0
        return momenta(x, s["v"][0], s["Q"][0], s["w"][0], mass, J, dil)

    be.substeps(None, 1)
    torch_gpu.cuda.synchronize()
    P0, L0 = now()
    be.substeps(None, 2000)
    torch_gpu.cuda.synchronize()
    P1, L1 = now()
    assert np.abs(P1 - P0).max() <= 1e-12 * np.abs(P0).max()
    assert np.abs(L1 - L0).max() / np.abs(L0).max() < 5e-9
    assert np.abs(be.state_numpy()["kappa"][0]).max() > 0.05
    be.close()


@pytest.mark.parametrize("math_mode", [0, 1], ids=["libm", "fast"])
def test_known_answer_discrete_bending_mode_on_the_gpu(torch_gpu, hip_lib, math_mode):
    """K18 on the HIP kernels themselves (no oracle involved): released from its first small-amplitude
    bending mode — the eigenvector of a mass / stiffness pair assembled independently from the discrete
    energies — the clamped rod oscillates in that mode alone at that eigenfrequency (1e-7)."""
    from gym_softrobot_amd import _capi
    from tests.elastica_chain import frequency_from_samples
    from tests.test_oracle_physics import _mode_case, mode_state

    c = _mode_case(16)
    om, x, Q, project = mode_state(c, 0)
    dt = 1e-4
    cfg = _capi.softpendulum_config(2, n_elems=c["n"], math_mode=math_mode)
    cfg.env_kind, cfg.features, cfg.dt, cfg.damping_constant = _capi.ENV_NONE, _capi.FEAT_FIXED_BC, dt, 0.0
    cfg.base_radius, cfg.youngs_modulus, cfg.shear_modulus = c["r"], c["E"], c["G"]
    be = _backend(cfg)
    be.reset_straight([0, 0, 0], [1, 0, 0], [0, 0, 1])
    _inject(be, "position", x)
    _inject(be, "director", Q.reshape(9, c["n"]))
    every, q = 25, [project(x, Q)]
    for _ in range(int(0.6 * 2 * np.pi / om / dt / every)):
        be.substeps(None, every)
        st = be.state_numpy()
        q.append(project(st["x"][1], st["Q"][1]))
    w, purity = frequency_from_samples(q, every * dt, dt)
    assert purity < 1e-8 and w == pytest.approx(om, rel=1e-7)
    be.close()
