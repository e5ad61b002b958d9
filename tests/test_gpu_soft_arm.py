"""SoftArmTracking-v0 on the GPU against the oracle (rtol 1e-5, flags exact), through the C-ABI.
The oracle's actuation is pinned against the reference's own MuscleTorquesWithVaryingBetaSplines
in tests/test_oracle_golden.py; here the HIP kernel is held to the oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

RTOL, ATOL = 1e-5, 1e-7
GOLD = __import__("pathlib").Path(__file__).resolve().parent / "golden"


@pytest.fixture(scope="module")
def torch_gpu():
    import torch

    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    return torch


def _oracle_rollout(oracle_c, cfg, actions):
    o = oracle_c.OracleRod(cfg)
    out = [(o.reset_soft_arm(), 0.0, False, False)]
    for a in actions:
        out.append(o.env_step_soft_arm(a))
    return o, out


def test_reset_observation_and_spaces(torch_gpu, hip_lib):
    import gym_softrobot_amd as gsa

    env = gsa.make("SoftArmTracking-v0")
    assert env.action_space.shape == (8,) and env.observation_space.shape == (14,)
    obs, info = env.reset(seed=3)
    z = np.load(GOLD / "softarm_vectors.npz")
    assert obs.dtype == np.float64
    np.testing.assert_allclose(obs, z["reset_obs"], rtol=0, atol=1e-7)
    env.close()


@pytest.mark.parametrize("n_envs,steps,n_elems", [(3, 12, 40), (1, 40, 40), (2, 6, 20), (2, 6, 80)])
def test_rollout_matches_oracle(torch_gpu, hip_lib, oracle_built, n_envs, steps, n_elems):
    """40 elements as the reference; 20 and 80 (two slots per lane) exercise the slot-generic
    prefix sum and segment means."""
    import gym_softrobot_amd as gsa
    from gym_softrobot_amd import _capi

    rng = np.random.default_rng(11 + n_envs)
    acts = rng.uniform(-1, 1, (steps, n_envs, 8)).astype(np.float32)
    if n_elems == 80:
        acts *= 0.2                       # dt = 2e-4 is near the stability limit of 12.5 mm elements
    acts[3] = acts[2]                     # unchanged control points: the cached profile is kept
    env = gsa.make_vec("SoftArmTracking-v0", n_envs, numpy_output=True, n_elems=n_elems)
    obs0, _ = env.reset(seed=0)
    got = [env.step(acts[t]) for t in range(steps)]
    cfg1 = _capi.soft_arm_config(1, n_elems=n_elems)
    for i in range(n_envs):
        o, ref = _oracle_rollout(oracle_built, cfg1, acts[:, i])
        np.testing.assert_allclose(obs0[i], ref[0][0], rtol=RTOL, atol=ATOL)
        for t in range(steps):
            ob, rew, term, trunc, _ = got[t]
            r_obs, r_rew, r_term, r_trunc = ref[t + 1]
            np.testing.assert_allclose(ob[i], r_obs, rtol=RTOL, atol=ATOL, err_msg=f"env {i} step {t}")
            np.testing.assert_allclose(rew[i], r_rew, rtol=RTOL, atol=ATOL)
            assert bool(term[i]) == r_term and bool(trunc[i]) == r_trunc
        # full state after the rollout, and the forcing objects' state
        snap = env.backend.rod_snapshot([i])
        np.testing.assert_allclose(snap["x"][0], o.get("x"), rtol=RTOL, atol=1e-4)    # millimetres
        np.testing.assert_allclose(snap["v"][0], o.get("v"), rtol=1e-4, atol=1e-2)
        np.testing.assert_allclose(snap["Q"][0], o.get("Q"), rtol=RTOL, atol=1e-7)
    env.close()


def test_truncation_after_five_seconds_and_determinism(torch_gpu, hip_lib):
    import gym_softrobot_amd as gsa

    env = gsa.make_vec("SoftArmTracking-v0", 2, numpy_output=True)
    env.reset(seed=0)
    a = np.zeros((2, 8), np.float32)
    a[1, :4] = 0.3
    first = None
    for t in range(500):                  # tick * sim_dt >= 5.0 first holds after 500 steps (:253)
        obs, rew, term, trunc, info = env.step(a)
        if trunc.any() and first is None:
            first = t + 1
    assert first == 500 and trunc.all() and not term.any()
    assert info["TimeLimit.truncated"].all()
    final = obs.copy()
    env.reset(seed=0)
    for t in range(500):
        obs, *_ = env.step(a)
    np.testing.assert_array_equal(obs, final)        # bitwise determinism
    assert np.isfinite(final).all()
    # the bent arm has moved its tip off the axis, the relaxed one has not
    assert abs(final[0, 8]) < 1e-6 and abs(final[1, 8]) + abs(final[1, 10]) > 1e-3
    env.close()


def test_autoreset_host_and_device_agree(torch_gpu, hip_lib):
    """NEXT_STEP auto-reset over the episode boundary (step 500 truncates, step 501 restarts):
    the host-driven and the device-side form must give the same stream, bit for bit."""
    import gym_softrobot_amd as gsa

    rng = np.random.default_rng(5)
    acts = (0.2 * rng.uniform(-1, 1, (8, 2, 8))).astype(np.float32)   # gentle: no blow-up before the time limit
    outs = []
    for mode in (True, "device"):
        env = gsa.make_vec("SoftArmTracking-v0", 2, numpy_output=True, autoreset=mode)
        env.reset(seed=0)
        rows = []
        for t in range(506):
            obs, rew, term, trunc, _ = env.step(acts[t % 8])
            if t >= 496:
                rows.append((obs.copy(), rew.copy(), term.copy(), trunc.copy()))
        outs.append(rows)
        env.close()
    for a, b in zip(*outs):
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)
    # step 500 (index 3 of the kept rows) truncates; the step after it returns the reset observation
    assert outs[0][3][3].all() and not outs[0][4][3].any()
    z = np.load(GOLD / "softarm_vectors.npz")
    np.testing.assert_allclose(outs[0][4][0][0], z["reset_obs"], rtol=0, atol=1e-7)


def test_game_mode_2_moving_target(torch_gpu, hip_lib, oracle_built):
    """game_mode 2: the target follows the trajectory drawn from the env's RNG at reset
    (pinned against the reference's generate_trajectory in tests/test_oracle_golden.py); it
    enters the observation and the reward only.  Oracle: same rod, target set per step."""
    import gym_softrobot_amd as gsa
    from gym_softrobot_amd import _capi

    z = np.load(GOLD / "softarm_trajectory.npz")
    env = gsa.make_vec("SoftArmTracking-v0", 2, game_mode=2, numpy_output=True)
    obs, _ = env.reset(seed=[0, 42])
    np.testing.assert_allclose(obs[0, 11:14], z["every50_0"][0] / 1000, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(obs[1, 11:14], z["every50_42"][0] / 1000, rtol=1e-6, atol=1e-7)
    rng = np.random.default_rng(3)
    o = oracle_built.OracleRod(_capi.soft_arm_config(1))
    o.reset_soft_arm()
    for t in range(10):
        a = (0.5 * rng.uniform(-1, 1, (2, 8))).astype(np.float32)
        obs, rew, term, trunc, _ = env.step(a)
        np.testing.assert_allclose(obs[0, 11:14], z["every50_0"][t + 1] / 1000, rtol=1e-6, atol=1e-7)
        np.testing.assert_allclose(obs[1, 11:14], z["every50_42"][t + 1] / 1000, rtol=1e-6, atol=1e-7)
        tgt = np.ascontiguousarray(z["every50_0"][t + 1], np.float64)     # (kept alive across the call)
        o._lib.oracle_set_arm_target(o._h, tgt.ctypes.data)
        r_obs, r_rew, r_term, r_trunc = o.env_step_soft_arm(a[0])
        np.testing.assert_allclose(obs[0], r_obs, rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(rew[0], r_rew, rtol=RTOL, atol=ATOL)
    env.close()
    # the single-env class takes the keyword like the reference's
    e1 = gsa.make("SoftArmTracking-v0", game_mode=2)
    ob, _ = e1.reset(seed=42)
    np.testing.assert_allclose(ob[11:14], z["every50_42"][0] / 1000, rtol=1e-6, atol=1e-7)
    e1.close()


def test_golden_oracle_rollout_fixture(torch_gpu, hip_lib):
    """The committed oracle rollout (tests/golden/softarm_oracle_rollout.npz): parity without
    the oracle at run time."""
    import gym_softrobot_amd as gsa

    z = np.load(GOLD / "softarm_oracle_rollout.npz")
    env = gsa.make_vec("SoftArmTracking-v0", 1, numpy_output=True)
    env.reset(seed=0)
    for t in range(len(z["actions"])):
        obs, rew, term, trunc, _ = env.step(z["actions"][t][None])
        np.testing.assert_allclose(obs[0], z["obs"][t], rtol=RTOL, atol=ATOL)
        np.testing.assert_allclose(rew[0], z["reward"][t], rtol=RTOL, atol=ATOL)
    np.testing.assert_allclose(env.backend.rod_snapshot([0])["x"][0], z["x"], rtol=RTOL, atol=1e-4)
    env.close()
