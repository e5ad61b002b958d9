"""OctoFlat-v0 (8 arms + rigid head, BASELINE.json configs[4]) parity: the HIP path through the
C-ABI against oracle/octoflat_oracle.inc.c on the same targets and actions.  rtol 1e-5 on
observations/rewards, exact flags and crossing counts."""
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL = 1e-5


@pytest.fixture(scope="module")
def torch_gpu():
    import torch

    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    return torch


def _targets(n, seed=0):
    from gym_softrobot_amd.seeding import np_random

    out = np.empty((n, 2))
    for i in range(n):
        rng, _ = np_random(seed + i)
        out[i] = (2 - 0.5) * rng.random(2) + 0.5      # flat_env.py:221
    return out


def _flat(ob):
    return np.concatenate([ob["individual"].ravel(), ob["shared"]])


def _compare_state(be, oracles, atol_x=1e-8):
    st = be.octo_state_numpy()
    for i, o in enumerate(oracles):
        hd = o.head()
        np.testing.assert_allclose(st["head_x"][i], hd["x"], rtol=RTOL, atol=1e-9)
        np.testing.assert_allclose(st["head_v"][i], hd["v"], rtol=RTOL, atol=1e-7)
        np.testing.assert_allclose(st["head_Q"][i], hd["Q"], rtol=RTOL, atol=1e-9)
        np.testing.assert_allclose(st["head_w"][i], hd["w"], rtol=RTOL, atol=1e-6)
        for a in range(o.n_arm):
            arm = o.arm(a)
            np.testing.assert_allclose(st["x"][i, a], arm.get("x"), rtol=RTOL, atol=atol_x)
            np.testing.assert_allclose(st["v"][i, a], arm.get("v"), rtol=RTOL, atol=1e-6)
            np.testing.assert_allclose(st["Q"][i, a], arm.get("Q"), rtol=RTOL, atol=1e-7)
            np.testing.assert_allclose(st["w"][i, a], arm.get("w"), rtol=RTOL, atol=1e-4)


def test_octo_reset_observation_and_state(torch_gpu, hip_lib, oracle_built):
    import gym_softrobot_amd as gsa

    n = 3
    env = gsa.make_vec("OctoFlat-v0", n, device=0)
    assert env.obs_dim == 461 and env.action_dim == 24
    obs, info = env.reset(seed=0)
    obs = obs.cpu().numpy()
    tg = _targets(n)
    np.testing.assert_array_equal(env.targets, tg)
    oracles = []
    for i in range(n):
        o = oracle_built.OracleOcto(env.cfg)
        ob = o.reset(tg[i])
        np.testing.assert_allclose(obs[i], _flat(ob), rtol=1e-6, atol=1e-7)
        oracles.append(o)
    _compare_state(env.backend, oracles, atol_x=1e-15)
    d = env.split_obs(obs)
    assert d["individual"].shape == (n, 8, 56) and d["shared"].shape == (n, 13)
    env.close()


@pytest.mark.parametrize("n_sub", [1, 7, 200])
def test_octo_short_steps_match_oracle(torch_gpu, hip_lib, oracle_built, n_sub):
    """Few substeps per env.step: isolates the substep operator order (joints, gravity,
    contact; head BC; dampers) from long-horizon drift."""
    from gym_softrobot_amd import _capi
    from gym_softrobot_amd.backend import HipRodBackend

    n, T = 4, 3
    cfg = _capi.octo_flat_config(n)
    cfg.n_substeps = n_sub
    be = HipRodBackend(cfg, device=0)
    tg = _targets(n, 11)
    be.reset_octo(tg)
    oracles = []
    for i in range(n):
        o = oracle_built.OracleOcto(cfg)
        o.reset(tg[i])
        oracles.append(o)
    acts = np.random.default_rng(3).uniform(-22, 22, (T, n, 24)).astype(np.float32)
    for t in range(T):
        obs, rew, term, trunc = (x.cpu().numpy() for x in be.step(acts[t]))
        for i, o in enumerate(oracles):
            ob, rw, te, tr = o.env_step(acts[t, i])
            np.testing.assert_allclose(obs[i], _flat(ob), rtol=RTOL, atol=2e-7)
            # forward reward = (dist - dist_before)/(n_sub dt): a difference of two O(1)
            # numbers divided by a small time -> absolute tolerance scales with 1/(n_sub dt)
            np.testing.assert_allclose(rew[i], rw, rtol=RTOL, atol=1e-13 / (n_sub * cfg.dt) + 1e-9)
            assert bool(term[i]) == te and bool(trunc[i]) == tr
    _compare_state(be, oracles)
    np.testing.assert_allclose(be.octo_state_numpy()["time"], [o.time for o in oracles], rtol=1e-12)
    be.close()


def test_octo_one_wave_variant_matches_oracle_and_the_two_wave_kernel(torch_gpu, hip_lib, oracle_built, monkeypatch):
    """softrod_octo1w.hpp (SOFTROD_OCTO_ONE_WAVE=1 at softrod_create: one wave per env, two slots per
    lane, no LDS rendezvous) is an A/B variant kept for measurement (profiles/README.md r3c: it loses,
    11.99 against 9.53 ms).  It must stay CORRECT: 200-substep steps against the oracle at rtol 1e-5,
    and within rounding of the shipped two-wave kernel (they differ in the order the eight joint
    loads are summed)."""
    from gym_softrobot_amd import _capi
    from gym_softrobot_amd.backend import HipRodBackend

    n, T, n_sub = 4, 3, 200
    cfg = _capi.octo_flat_config(n)
    cfg.n_substeps = n_sub
    two = HipRodBackend(cfg, device=0)
    monkeypatch.setenv("SOFTROD_DEBUG_SWITCHES", "1")      # the A/B switches count only with this one
    monkeypatch.setenv("SOFTROD_OCTO_ONE_WAVE", "1")       # read once, in softrod_create
    one = HipRodBackend(cfg, device=0)
    monkeypatch.delenv("SOFTROD_OCTO_ONE_WAVE")
    monkeypatch.delenv("SOFTROD_DEBUG_SWITCHES")
    assert "octo1w" in one.kernel_tier() and "2 waves,4 envs/wg" in two.kernel_tier()
    tg = _targets(n, 11)
    oracles = []
    for be in (one, two):
        be.reset_octo(tg)
    for i in range(n):
        o = oracle_built.OracleOcto(cfg)
        o.reset(tg[i])
        oracles.append(o)
    acts = np.random.default_rng(3).uniform(-22, 22, (T, n, 24)).astype(np.float32)
    for t in range(T):
        obs, rew, term, trunc = (x.cpu().numpy().copy() for x in one.step(acts[t]))
        obs2, rew2, term2, trunc2 = (x.cpu().numpy().copy() for x in two.step(acts[t]))
        np.testing.assert_allclose(obs, obs2, rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(rew, rew2, rtol=1e-6, atol=1e-7)
        assert (term == term2).all() and (trunc == trunc2).all()
        for i, o in enumerate(oracles):
            ob, rw, te, tr = o.env_step(acts[t, i])
            np.testing.assert_allclose(obs[i], _flat(ob), rtol=RTOL, atol=2e-7)
            np.testing.assert_allclose(rew[i], rw, rtol=RTOL, atol=1e-13 / (n_sub * cfg.dt) + 1e-9)
            assert bool(term[i]) == te and bool(trunc[i]) == tr
    _compare_state(one, oracles)
    x1, x2 = one.octo_state_numpy()["x"], two.octo_state_numpy()["x"]
    np.testing.assert_allclose(x1, x2, rtol=1e-7, atol=1e-9)
    one.close()
    two.close()


def test_octo_first_full_step_matches_oracle(torch_gpu, hip_lib, oracle_built):
    """Reference configuration: 2857 substeps per env.step from rest, random knots.

    Only the FIRST env.step is held to rtol 1e-5 at this horizon.  The reference's physics is
    chaotic at the level of rounding once the arms slide on the plane (sign() and stick/slip
    switches of the anisotropic friction): two builds of the SAME oracle source, with and
    without FMA contraction, already differ by 1e-3 in the observation after two such steps
    (tools/octo_rounding_sensitivity.py).  Later steps are covered by the
    windowed test below, which re-synchronises the states."""
    import gym_softrobot_amd as gsa

    n = 6
    env = gsa.make_vec("OctoFlat-v0", n, device=0)
    env.reset(seed=5)
    tg = _targets(n, 5)
    acts = np.random.default_rng(9).uniform(-22, 22, (n, 24)).astype(np.float32)
    obs, rew, term, trunc, info = env.step(acts)
    obs, rew, term, trunc = (x.cpu().numpy() for x in (obs, rew, term, trunc))
    crossings = 0
    for i in range(n):
        o = oracle_built.OracleOcto(env.cfg)
        o.reset(tg[i])
        ob, rw, te, tr = o.env_step(acts[i])
        crossings += o.crossings()
        np.testing.assert_allclose(obs[i], _flat(ob), rtol=RTOL, atol=2e-6)
        np.testing.assert_allclose(rew[i], rw, rtol=RTOL, atol=1e-7)
        assert bool(term[i]) == te and bool(trunc[i]) == tr
        assert info["time"][i] == pytest.approx(o.time, rel=1e-12)
    assert crossings > 0, "the step was meant to exercise the arm-crossing count"
    env.close()


def _inject_octo(be, oracles):
    """Overwrite the resident state of every env with its oracle's (arms, head, time)."""
    import torch

    st = be.state()
    seg = st["arm_stride"]
    dev = st["position"].device
    for i, o in enumerate(oracles):
        for a in range(o.n_arm):
            arm = o.arm(a)
            x, v, w = arm.get("x"), arm.get("v"), arm.get("w")
            q = arm.get("Q").reshape(9, -1)
            lo = a * seg
            st["position"][:, i, lo : lo + x.shape[1]] = torch.from_numpy(x).to(dev)
            st["velocity"][:, i, lo : lo + v.shape[1]] = torch.from_numpy(v).to(dev)
            st["omega"][:, i, lo : lo + w.shape[1]] = torch.from_numpy(w).to(dev)
            st["director"][:, i, lo : lo + q.shape[1]] = torch.from_numpy(np.ascontiguousarray(q)).to(dev)
        h = o.head()
        st["head"][0:18, i] = torch.from_numpy(np.concatenate([h["x"], h["v"], h["Q"].ravel(), h["w"]])).to(dev)
        st["time"][i] = o.time


@pytest.mark.parametrize("amp,min_strict", [(22.0, 1.0), (3.0, 0.9)], ids=["hard", "gentle"])
def test_octo_windowed_parity_along_oracle_trajectory(torch_gpu, hip_lib, oracle_built, amp, min_strict):
    """Strict parity in every dynamical regime the octopus reaches: the oracle runs a long
    trajectory (48 windows of 200 substeps, new random knots every third window); before each
    window the GPU state is overwritten with the oracle's, both advance one window, and the
    observations/rewards/flags are compared at rtol 1e-5.  Errors therefore never accumulate
    beyond 200 substeps while the arms fold, cross and slide.

    With gentle actions the arms hover around stick/slip, where a sign() switch of the friction
    law can amplify a rounding difference beyond the tolerance inside one window (the oracle
    against its own FMA-contracted build does so in 1 of 240 windows); such windows are allowed
    in at most 10 % of the cases and must still agree to 1e-2."""
    from gym_softrobot_amd import _capi
    from gym_softrobot_amd.backend import HipRodBackend

    n, windows = 4, 48
    cfg = _capi.octo_flat_config(n)
    cfg.n_substeps = 200
    be = HipRodBackend(cfg, device=0)
    tg = _targets(n, 31)
    be.reset_octo(tg)
    oracles = []
    for i in range(n):
        o = oracle_built.OracleOcto(cfg)
        o.reset(tg[i])
        oracles.append(o)
    rng = np.random.default_rng(17)
    acts = None
    strict = total = 0
    crossings = 0
    for w in range(windows):
        if w % 3 == 0:
            acts = rng.uniform(-amp, amp, (n, 24)).astype(np.float32)
        _inject_octo(be, oracles)
        obs, rew, term, trunc = (x.cpu().numpy() for x in be.step(acts))
        for i, o in enumerate(oracles):
            ob, rw, te, tr = o.env_step(acts[i])
            crossings += o.crossings()
            ref = _flat(ob)
            ok = np.allclose(obs[i], ref, rtol=RTOL, atol=5e-7) and np.isclose(
                rew[i], rw, rtol=RTOL, atol=1e-11 / (200 * cfg.dt) + 1e-9)
            strict += bool(ok)
            total += 1
            np.testing.assert_allclose(obs[i], ref, rtol=1e-2, atol=1e-2)
            assert bool(term[i]) == te and bool(trunc[i]) == tr
    assert strict >= min_strict * total, f"{strict}/{total} windows within rtol 1e-5"
    if amp > 10:
        assert crossings > 0
    be.close()


def test_octo_crossing_count_and_symmetric_curl(torch_gpu, hip_lib, oracle_built):
    """Alternating +-22 knots fold neighbouring arms over each other (20 crossings after one
    step in the oracle); survive_reward = -0.02 * crossings must agree exactly."""
    from gym_softrobot_amd import _capi
    from gym_softrobot_amd.backend import HipRodBackend

    cfg = _capi.octo_flat_config(1)
    be = HipRodBackend(cfg, device=0)
    tg = np.array([[1.0, 1.5]])
    be.reset_octo(tg)
    o = oracle_built.OracleOcto(cfg)
    o.reset(tg[0])
    a = np.zeros((8, 3), np.float32)
    for i in range(8):
        a[i] = 22.0 * (1 if i % 2 == 0 else -1)
    for _ in range(3):
        obs, rew, term, trunc = (x.cpu().numpy() for x in be.step(a.reshape(1, 24)))
        ob, rw, te, tr = o.env_step(a.reshape(-1))
        assert o.crossings() > 0
        # the symmetric load leaves the head at rest up to rounding: forward reward ~ 1e-15/0.2
        assert round(float(rew[0]) / -0.02) == o.crossings()
        np.testing.assert_allclose(rew[0], rw, rtol=RTOL, atol=1e-9)
        np.testing.assert_allclose(obs[0], _flat(ob), rtol=RTOL, atol=5e-6)
    be.close()


def test_octo_reaching_the_target_terminates(torch_gpu, hip_lib, oracle_built):
    from gym_softrobot_amd import _capi
    from gym_softrobot_amd.backend import HipRodBackend

    cfg = _capi.octo_flat_config(2)
    cfg.n_substeps = 20
    be = HipRodBackend(cfg, device=0)
    tg = np.array([[0.05, 0.02], [0.7, 0.9]])       # env 0: already within 0.1 of the head
    be.reset_octo(tg)
    a = np.zeros((2, 24), np.float32)
    obs, rew, term, trunc = (x.cpu().numpy() for x in be.step(a))
    assert term.tolist() == [1, 0] and trunc.tolist() == [0, 0]
    for i in range(2):
        o = oracle_built.OracleOcto(cfg)
        o.reset(tg[i])
        ob, rw, te, tr = o.env_step(a[i])
        assert te == bool(term[i])
        np.testing.assert_allclose(rew[i], rw, rtol=RTOL, atol=1e-9)
    assert rew[0] == pytest.approx(100.0 - (np.hypot(0.05, 0.02) - 0.1), rel=1e-9)
    be.close()


def test_octo_lite_variant_one_arm_eight_knots(torch_gpu, hip_lib, oracle_built):
    """OctoFlatLite-v0 (gym_softrobot/__init__.py:11-15): n_arm = 1, n_action = 8."""
    import gym_softrobot_amd as gsa

    env = gsa.make("OctoFlatLite-v0", device=0)
    assert env.action_space.shape == (8,)
    assert env.observation_space["individual"].shape == (1, 9 + 44 + 8)
    ob0, info = env.reset(seed=2)
    o = oracle_built.OracleOcto(env._vec.cfg)
    oo = o.reset(env._target)
    np.testing.assert_allclose(ob0["individual"], oo["individual"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(ob0["shared"], oo["shared"], rtol=1e-6, atol=1e-7)
    rng = np.random.default_rng(4)
    for _ in range(2):
        a = rng.uniform(-22, 22, 8).astype(np.float32)
        ob, r, te, tr, info = env.step(a)
        ob2, rw, te2, tr2 = o.env_step(a)
        assert env.observation_space.contains(ob)
        np.testing.assert_allclose(ob["individual"], ob2["individual"], rtol=RTOL, atol=5e-6)
        np.testing.assert_allclose(ob["shared"], ob2["shared"], rtol=RTOL, atol=5e-6)
        assert r == pytest.approx(rw, rel=RTOL, abs=1e-6) and te == te2 and tr == tr2
        assert set(info) == {"time", "TimeLimit.truncated"}
    env.close()


def test_octo_other_shapes(torch_gpu, hip_lib, oracle_built):
    """Arm counts and lengths other than the reference's: 5 arms of 12 elements (ghost slots
    past the last arm), 3 arms of 16 elements (32 slots per arm), 8 arms of 16 elements (four
    wavefronts per env: the wide instantiation of the step kernel), 2 arms of 40 elements (64
    slots per arm).

    The explicit joint spring limits the arm resolution: node 0 of an arm carries half an
    element mass m0 on a spring of stiffness joint_k = 1e6, and PositionVerlet needs
    dt * sqrt(joint_k / m0) < 2; at the reference's dt = 7e-5 that is 1.35 for 10 elements, 1.7
    for 16 and 1.9 for 20, where the oracle itself amplifies a 1e-10 perturbation to overflow
    within 90 substeps.  Longer arms are therefore tested with a softer joint."""
    from gym_softrobot_amd import _capi
    from gym_softrobot_amd.backend import HipRodBackend

    for n_arm, n_elems, nk, joint_k in ((5, 12, 3, 1e6), (3, 16, 4, 1e6), (8, 16, 3, 1e6), (2, 40, 3, 1e5)):
        cfg = _capi.octo_flat_config(2, n_elems=n_elems, n_arm=n_arm, n_action=nk)
        cfg.n_substeps = 150
        cfg.joint_k = joint_k
        be = HipRodBackend(cfg, device=0)
        tg = _targets(2, 21)
        be.reset_octo(tg)
        oracles = []
        for i in range(2):
            o = oracle_built.OracleOcto(cfg)
            o.reset(tg[i])
            oracles.append(o)
        acts = np.random.default_rng(n_arm).uniform(-15, 15, (3, 2, n_arm * nk)).astype(np.float32)
        for t in range(3):
            obs, rew, term, trunc = (x.cpu().numpy() for x in be.step(acts[t]))
            for i, o in enumerate(oracles):
                ob, rw, te, tr = o.env_step(acts[t, i])
                np.testing.assert_allclose(obs[i], _flat(ob), rtol=RTOL, atol=1e-6)
                np.testing.assert_allclose(rew[i], rw, rtol=RTOL, atol=1e-8)
                assert bool(term[i]) == te and bool(trunc[i]) == tr
        _compare_state(be, oracles)
        be.close()


def test_octo_determinism_masked_reset_and_packed(torch_gpu, hip_lib):
    import gym_softrobot_amd as gsa
    from gym_softrobot_amd.distributed import packed_width, unpack_outputs

    n = 5
    acts = np.random.default_rng(2).uniform(-22, 22, (2, n, 24)).astype(np.float32)
    runs = []
    for _ in range(2):
        env = gsa.make_vec("OctoFlat-v0", n, device=0)
        env.reset(seed=3)
        out = []
        for t in range(2):
            o, r, te, tr, _ = env.step(acts[t])
            out.append((o.cpu().numpy().copy(), r.cpu().numpy().copy(), te.cpu().numpy().copy()))
        runs.append((env, out))
    for (o1, r1, t1), (o2, r2, t2) in zip(runs[0][1], runs[1][1]):
        np.testing.assert_array_equal(o1, o2)
        np.testing.assert_array_equal(r1, r2)
        np.testing.assert_array_equal(t1, t2)
    env_a, env_b = runs[0][0], runs[1][0]
    # packed step == separate outputs, bit for bit (obs_dim 461 is odd -> padded row)
    p = env_a.step_packed(acts[0])[0]
    assert p.shape == (n, packed_width(461))
    o2, r2, te2, tr2, _ = env_b.step(acts[0])
    o1, r1, te1, tr1 = unpack_outputs(p, 461)
    assert torch_gpu.equal(o1, o2) and torch_gpu.equal(r1, r2)
    assert torch_gpu.equal(te1, te2) and torch_gpu.equal(tr1, tr2)
    # masked reset: only env 1 restarts (new target drawn from its own stream)
    before = env_a.backend.octo_state_numpy()
    mask = np.zeros(n, bool)
    mask[1] = True
    obs, _ = env_a.reset(mask=mask)
    after = env_a.backend.octo_state_numpy()
    for k in ("x", "v", "head_x", "head_v", "time"):
        np.testing.assert_array_equal(np.delete(after[k], 1, axis=0), np.delete(before[k], 1, axis=0))
    assert after["time"][1] == 0.0 and np.all(after["head_x"][1] == 0.0)
    assert not np.array_equal(after["target"][1], before["target"][1])
    env_a.close()
    env_b.close()


def test_octo_crossing_count_against_the_reference_function(torch_gpu, hip_lib):
    """The kernel's crossing count against outputs of the reference's own intersection()
    (tests/golden/intersection_vectors.npz): arm shapes are written into the resident state of
    a two-arm env, one substep of negligible length runs, and the count is read off the reward
    (survive_reward = -0.02 * crossings, flat_env.py:372)."""
    import torch

    from gym_softrobot_amd import _capi
    from gym_softrobot_amd.backend import HipRodBackend

    z = np.load(Path(__file__).parent / "golden" / "intersection_vectors.npz")
    n = len(z["count"])
    cfg = _capi.octo_flat_config(n, n_arm=2)
    cfg.n_substeps = 1
    cfg.dt = 1e-13            # the arms do not move: forward reward ~ 0
    be = HipRodBackend(cfg, device=0)
    be.reset_octo(np.tile([[1.0, 1.0]], (n, 1)))
    st = be.state()
    seg = st["arm_stride"]
    for a, key in ((1, "p1"), (0, "p2")):
        st["position"][0:2, :, a * seg : a * seg + 11] = torch.from_numpy(
            np.ascontiguousarray(z[key].transpose(1, 0, 2))).to(st["position"].device)
    obs, rew, term, trunc = be.step(np.zeros((n, 6), np.float32))
    rew = rew.cpu().numpy()
    got = np.rint(-rew / 0.02).astype(int)
    assert np.all(np.abs(rew + 0.02 * got) < 1e-3)
    np.testing.assert_array_equal(got, z["count"])
    be.close()


def test_octo_diagnostic_taps(hip_lib):
    """FlatEnv(config_generate_video=True, config_save_head_data=True): one RodCallBack dict per
    arm and the head's dict, one sample per env.step (octopus/flat_env.py:188-206,
    utils/custom_elastica/callback_func.py:4-41)."""
    import gym_softrobot_amd as gsa

    env = gsa.make("OctoFlat-v0", config_generate_video=True, config_save_head_data=True, recording_fps=71)
    env.reset(seed=0)
    assert len(env.rod_parameters_dict_list) == 8 and len(env.head_dict["time"]) == 0
    rng = np.random.default_rng(0)
    for _ in range(3):
        obs, *_ = env.step(rng.uniform(-5, 5, 24).astype(np.float32))
    n = 10
    for p in env.rod_parameters_dict_list:
        assert set(p) == {"time", "radius", "dilatation", "voronoi_dilatation", "position", "director",
                          "velocity", "omega", "sigma", "kappa"}
        assert len(p["time"]) == 3 and p["position"][-1].shape == (3, n + 1)
        assert p["director"][-1].shape == (3, 3, n) and p["kappa"][-1].shape == (3, n - 1)
        assert np.all(np.isfinite(p["kappa"][-1])) and np.allclose(p["dilatation"][-1], 1.0, atol=0.05)
    hd = env.head_dict
    assert set(hd) == {"time", "step", "position", "velocity"} and hd["step"] == [201, 402, 603]
    assert hd["position"][-1].shape == (3, 1) and hd["time"][-1] == pytest.approx(3 * 201 * 7.0e-5, rel=1e-9)
    # every arm's base node stays on the head's rim (joint springs): |x_arm0 - x_head| ~ head radius
    for p in env.rod_parameters_dict_list:
        d = p["position"][-1][:2, 0] - hd["position"][-1][:2, 0]
        assert np.hypot(*d) == pytest.approx(0.04, abs=2e-3)
    env.reset(seed=0)
    assert len(env.head_dict["time"]) == 0           # fresh dicts per reset
    env.close()


def test_octo_decentralized_policy_mode(hip_lib):
    """policy_mode="decentralized" (flat_env.py:111-130, 248-260): the same simulation, one arm's
    spaces declared, the one-hot arm index appended to every arm's observation row."""
    import gym_softrobot_amd as gsa

    a = np.random.default_rng(2).uniform(-5, 5, 24).astype(np.float32)
    outs = {}
    for mode in ("centralized", "decentralized"):
        env = gsa.make("OctoFlat-v0", policy_mode=mode, recording_fps=71)
        ob0, _ = env.reset(seed=4)
        ob1, rew, term, trunc, _ = env.step(a)
        outs[mode] = (ob0, ob1, rew, env.action_space.shape, env.observation_space["individual"].shape)
        env.close()
    c, d = outs["centralized"], outs["decentralized"]
    assert c[3] == (24,) and d[3] == (3,) and c[4] == (8, 56) and d[4] == (64,)
    for k in (0, 1):
        np.testing.assert_array_equal(d[k]["individual"][:, :56], c[k]["individual"])
        np.testing.assert_array_equal(d[k]["individual"][:, 56:], np.eye(8, dtype=np.float32))
        np.testing.assert_array_equal(d[k]["shared"], c[k]["shared"])
    assert c[2] == d[2]


def test_octo_render_rgb_array(hip_lib):
    import gym_softrobot_amd as gsa

    env = gsa.make("OctoFlat-v0", render_mode="rgb_array", recording_fps=71)
    env.reset(seed=0)
    f0 = env.render()
    assert f0.shape == (600, 800, 3) and f0.dtype == np.uint8
    env.step(np.full(24, 10.0, np.float32))
    assert (env.render() != f0).any()
    env.close()
