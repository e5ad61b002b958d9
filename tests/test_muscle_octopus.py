"""The muscle octopus envs (OctoCrawl-v0, OctoArmTwo-v0, OctoReach-v0) on the CPU: the configuration against the builds
recorded while EXECUTING the reference's files (tools/make_muscle_octopus_golden.py ->
tests/golden/ref_muscle_octopus_build_records.json), the oracle's env code (tests/oracle_mocto.py) against the executed
reference's set_action / get_state / step (ref_muscle_octopus.npz), the host classes over the CPU test double.  The
muscle law underneath is the restated COOMM model: PARITY UNPINNED (tests/test_muscles.py says what is and is not
pinned); the GPU side of the same fixtures is tests/test_gpu_muscle_octopus.py."""
import ctypes as C
import json
from pathlib import Path

import numpy as np
import pytest

from gym_softrobot_amd import _capi

GOLD = Path(__file__).resolve().parent / "golden"
ENVS = {"OctoCrawl": ("crawl_", _capi.ENV_CRAWL, "OctoCrawl-v0"), "OctoArmTwo": ("armtwo_", _capi.ENV_ARM_TWO, "OctoArmTwo-v0"),
        "OctoReach": ("reach_", _capi.ENV_REACH, "OctoReach-v0")}


@pytest.mark.parametrize("name", list(ENVS))
def test_config_matches_the_executed_builds(name):
    """build_octopus_muscles / build_two_arms and the envs' reset(), executed: every argument they hand to PyElastica."""
    r = json.loads((GOLD / "ref_muscle_octopus_build_records.json").read_text())[name]
    _, kind, _ = ENVS[name]
    cfg = _capi.muscle_octopus_config(kind, 1)
    init = r["init"]
    assert (init["step_skip"], init["time_step"], init["final_time"]) == (cfg.n_substeps, cfg.dt, cfg.final_time)
    assert (init["n_arm"], init["n_elems"], init["n_action"]) == (cfg.n_arm, cfg.n_elem, cfg.n_knots)
    assert init["obs_shape"] == [_capi.config_obs_dim(cfg)] and init["action_shape"] == [_capi.config_action_dim(cfg)]
    assert (init["action_low"], init["action_high"], init["reward_range"]) == (0.0, 1.0, 100.0)
    pos, dirs, angles = _capi.muscle_octopus_arm_frames(kind, cfg.head_radius)
    assert len(r["arms"]) == cfg.n_arm
    for a, arm in enumerate(r["arms"]):
        np.testing.assert_array_equal(arm["start"], pos[a])
        np.testing.assert_array_equal(arm["direction"], dirs[a])
        assert arm["normal"] == [0.0, 0.0, 1.0] and arm["n_elements"] == cfg.n_elem
        np.testing.assert_array_equal(arm["base_radius"], _capi.muscle_octopus_radii(cfg.n_elem))
        assert (arm["base_length"], arm["density"], arm["youngs_modulus"], arm["shear_modulus"]) == \
            (cfg.base_length, cfg.density, cfg.youngs_modulus, cfg.shear_modulus)
    cyl = r["cylinder"]
    assert (cyl["base_length"], cyl["base_radius"], cyl["density"]) == (cfg.head_length, cfg.head_radius, cfg.head_density)
    assert cyl["direction"] == [0.0, 0.0, 1.0] and cyl["normal"] == [0.0, 1.0, 0.0]
    np.testing.assert_allclose(np.asarray(cyl["start"]) + np.asarray(cyl["direction"]) * cyl["base_length"] / 2,
                               [cfg.head_center[i] for i in range(3)], rtol=0, atol=1e-18)
    ops = r["ops"]
    dampers = [o["kwargs"] for o in ops if o["cls"] == "AnalyticalLinearDamper"]
    assert len(dampers) == cfg.n_arm and all(d == {"damping_constant": cfg.damping_constant, "time_step": cfg.damper_time_step} for d in dampers)
    assert cfg.damper_time_step == 7e-5 != cfg.dt                      # the dampers' own time_step (build_muscle_octopus.py:105)
    joints = [o["kwargs"] for o in ops if o["cls"] == "FixedJoint2Rigid"]
    assert [j["angle"] for j in joints] == angles == [cfg.joint_angle0 + cfg.joint_angle_step * a for a in range(cfg.n_arm)]
    assert all((j["k"], j["nu"], j["kt"], j["radius"]) == (cfg.joint_k, cfg.joint_nu, cfg.joint_kt, cfg.head_radius) for j in joints)
    assert r["connect_indices"] == [-1, 0]
    # registration order: per arm append + dampen, the head, its BodyBoundaryCondition, the joints, the muscles; THEN the
    # env's own constraints — the dampers are registered before every arm constraint (damp_before_constrain)
    order = r["order"]
    na = cfg.n_arm
    assert order[: 2 * na] == [s for a in range(na) for s in (f"append:FakeRod[{a}]", f"damping:AnalyticalLinearDamper[{a}]")]
    assert order[2 * na: 2 * na + 2] == [f"append:Cylinder[{na}]", f"constrain:BodyBoundaryCondition[{na}]"]
    assert order[2 * na + 2: 3 * na + 2] == [f"connect:FixedJoint2Rigid[{na},{a}]" for a in range(na)]
    assert order[3 * na + 2: 4 * na + 2] == [f"forcing:ApplyMuscles[{a}]" for a in range(na)]
    tail = order[4 * na + 2:]
    if kind == _capi.ENV_CRAWL:
        assert tail == [f"constrain:ControllableFixConstraint[{a}]" for a in range(na)]
        assert r["suckers"] == [[{"index": 0, "flag": True, "reduction_ratio": 1.0}]] * na
        assert (cfg.n_suckers, cfg.sucker_index[0], cfg.sucker_reduction_ratio, cfg.head_fixed) == (1, 0, 1.0, 0)
    elif kind == _capi.ENV_ARM_TWO:
        assert tail == [f"constrain:ControllableFixConstraint[{a}]" for a in range(na) for _ in range(3)]
        assert init["sucker_location"] == [cfg.sucker_index[j] for j in range(3)] == [3, 9, 15]
        assert init["control_location"] == [0, 3, 9, 15, 19]
        assert r["suckers"] == [[{"index": i, "flag": True, "reduction_ratio": 1.0} for i in (3, 9, 15)]] * 2
        assert (cfg.n_suckers, cfg.head_fixed) == (3, 0)
    else:
        assert tail == [f"constrain:OneEndFixedBC[{na}]"]               # on the HEAD, after its BodyBoundaryCondition
        assert (cfg.n_suckers, cfg.head_fixed) == (0, 1)
    assert cfg.damp_before_constrain == 1
    # the layers: create_es_muscle_layers(rod.radius, base_radius) — what _capi.es_muscle_layers turns into tables
    layers = r["muscle_layers_arm0"]
    assert [m["kind"] for m in layers] == ["LongitudinalMuscle", "LongitudinalMuscle", "TransverseMuscle"]
    radii = _capi.muscle_octopus_radii(cfg.n_elem)
    np.testing.assert_allclose(layers[0]["rest_muscle_area"], (radii / 0.013) ** 2, rtol=1e-15)
    assert [o["kwargs"] for o in ops if o["cls"] == "ApplyMuscles"] == [{"step_skip": 10000}] * na


def test_every_muscle_config_builder_matches_its_c_twin(hip_lib):
    for kind in _capi.MUSCLE_OCTOPUS_ENVS:
        c = _capi.SoftrodConfig()
        assert hip_lib.softrod_config_muscle_octopus(C.byref(c), 5, kind) == 0
        py = _capi.muscle_octopus_config(kind, 5)
        assert bytes(c) == bytes(py), kind
        assert hip_lib.softrod_config_action_dim(C.byref(c)) == _capi.config_action_dim(c) == _capi.action_dim(kind)
        assert hip_lib.softrod_config_obs_dim(C.byref(c)) == _capi.config_obs_dim(c) == _capi.obs_dim(kind)
    for cname, py in (("softrod_config_arm_pull_weight", _capi.arm_pull_weight_config(5)),):
        c = _capi.SoftrodConfig()
        assert getattr(hip_lib, cname)(C.byref(c), 5) == 0 and bytes(c) == bytes(py)
        assert hip_lib.softrod_config_obs_dim(C.byref(c)) == _capi.config_obs_dim(c) == 84
        assert hip_lib.softrod_config_action_dim(C.byref(c)) == _capi.config_action_dim(c) == 2
    for mode, name in ((0, "discrete"), (1, "continuous")):
        c = _capi.SoftrodConfig()
        assert hip_lib.softrod_config_arm_push(C.byref(c), 5, mode) == 0 and bytes(c) == bytes(_capi.arm_push_config(5, mode=name))
    assert hip_lib.softrod_config_muscle_octopus(C.byref(_capi.SoftrodConfig()), 5, _capi.ENV_OCTO_FLAT) != 0


def _replay_env(kind, z, p, k):
    """The oracle's env code on fixture row k: pre-step bookkeeping installed, the recorded post-loop state as 'stepper'."""
    from tests.oracle_mocto import MuscleOctopusOracleEnv

    cfg = _capi.muscle_octopus_config(kind, 1)
    e = MuscleOctopusOracleEnv(cfg)
    e.reset(z[p + "reset_target"] if kind == _capi.ENV_REACH else None)
    e._prev_action = z[p + "step_prev_action_before"][k].astype(np.float32).copy()
    e._prev_kappa[...] = z[p + "step_prev_kappa_before"][k]
    hx = z[p + "step_pre_hx"][k]
    e.body.set_head(hx, np.zeros(3), np.eye(3), np.zeros(3))
    e.body.set_time(float(z[p + "step_pre_time"][k]))

    def stepper():
        for a in range(e.n_arm):
            e.arm(a).set("x", z[p + "step_x"][k][a])
            e.arm(a).set("v", z[p + "step_v"][k][a])
            e.arm(a).set("kappa", z[p + "step_kappa"][k][a])
        e.body.set_head(z[p + "step_hx"][k], z[p + "step_hv"][k], z[p + "step_hQ"][k], np.zeros(3))
        e.body.set_time(float(z[p + "step_time"][k]))
    return e, stepper


@pytest.mark.parametrize("name", list(ENVS))
def test_oracle_env_code_replays_the_executed_reference(oracle_built, name):
    """set_action, get_state and step()'s reward / termination code of the reference's files, executed around a scripted
    stepper, against tests/oracle_mocto.py on the same states: observations bit for bit, rewards to rounding, flags,
    sucker indices / ratios and every layer's activation array as the reference's objects received them."""
    p, kind, _ = ENVS[name]
    z = np.load(GOLD / "ref_muscle_octopus.npz")
    e = _replay_env(kind, z, p, 0)[0]
    np.testing.assert_allclose(e.reset(z[p + "reset_target"] if kind == _capi.ENV_REACH else None), z[p + "reset_obs"], rtol=0, atol=1e-12)
    labels = [str(s) for s in z[p + "step_label"]]
    assert {"nan_x", "nan_v", "inf_v", "time_just_past", "time_eq_final"} <= set(labels)
    for k, label in enumerate(labels):
        e, stepper = _replay_env(kind, z, p, k)
        obs, rew, term, trunc = e.step(z[p + "step_action"][k], stepper=stepper)
        np.testing.assert_array_equal(obs, z[p + "step_obs"][k], err_msg=label)
        assert obs.dtype == np.float32
        want = float(z[p + "step_reward"][k])
        assert (np.isnan(want) and np.isnan(rew)) or rew == pytest.approx(want, rel=1e-13, abs=1e-15), (label, rew, want)
        assert (bool(term), bool(trunc)) == (bool(z[p + "step_terminated"][k]), bool(z[p + "step_truncated"][k])), label
        for a in range(e.n_arm):
            acts = z[p + "step_activations"][k][a]
            got = e.arm(a).get("muscle_activation")
            for m in range(3):
                if np.isfinite(acts[m]).all():
                    np.testing.assert_allclose(got[m], acts[m], rtol=1e-15, atol=0, err_msg=f"{label} arm {a} layer {m}")
                else:                                               # the layer received nothing in this step: as reset left it
                    np.testing.assert_array_equal(got[m], 0.0)
            ns = int(e.cfg.n_suckers)
            np.testing.assert_array_equal(e.arm(a).get("sucker_index")[:ns], z[p + "step_sucker_index"][k][a][:ns])
            np.testing.assert_array_equal(e.arm(a).get("sucker_ratio")[:ns], z[p + "step_sucker_ratio"][k][a][:ns])
        if kind == _capi.ENV_ARM_TWO:
            np.testing.assert_array_equal(e._prev_kappa, z[p + "step_prev_kappa_after"][k])
    flags = {lab: (bool(z[p + "step_terminated"][k]), bool(z[p + "step_truncated"][k])) for k, lab in enumerate(labels)}
    assert flags["nan_x"] == (True, False) and flags["time_eq_final"] == (False, False) and flags["time_just_past"] == (False, True)
    assert flags["nan_kappa"] == (False, False) and flags["inf_v"] == (False, False)       # only NaN positions / velocities end it
    if kind == _capi.ENV_REACH:
        assert flags["tip_at_target"] == (True, False) and flags["tip_at_target_late"] == (True, True)      # reach_env.py:247-252
    else:
        assert flags["head_at_target"] == (True, False) and flags["head_at_target_late"] == (True, False)
        assert flags["moved_and_late"] == (False, True)


def test_arm_two_activation_basis_is_the_reference_interpolation():
    """The device computes ArmTwoEnv's `interp1d(control_location, [0, *act, 0], "cubic")(range(n))` as basis @ act."""
    from scipy.interpolate import interp1d

    basis, loc = _capi.arm_two_activation_basis(20, 3)
    assert loc == [3, 9, 15] and basis.shape == (20, 3)
    rng = np.random.default_rng(0)
    for _ in range(20):
        act = rng.uniform(0, 1, 3)
        want = interp1d([0] + loc + [19], [0] + list(act) + [0], kind="cubic")(range(20))
        np.testing.assert_allclose(basis @ act, want, rtol=1e-13, atol=1e-15)


@pytest.mark.parametrize("name", list(ENVS))
def test_host_classes_over_the_cpu_double(oracle_built, name):
    """make / make_vec, spaces, the reference's API assertions (tests/envs/test_envs.py:27-47), ReachEnv's target draw, the
    parity label, host auto-reset; ArmTwoEnv's _prev_kappa survives reset() like the reference's attribute."""
    import gym_softrobot_amd as gsa
    from tests.oracle_backend import OracleBackend

    p, kind, env_id = ENVS[name]
    z = np.load(GOLD / "ref_muscle_octopus.npz")
    label = gsa.parity_label(env_id)
    assert label is not None and "parity-unpinned" in label and "COOMM" in label
    env = gsa.make(env_id, backend=OracleBackend(_capi.muscle_octopus_config(kind, 1)))
    ob_space, act_space = env.observation_space, env.action_space
    ob, info = env.reset(seed=0)
    assert isinstance(info, dict) and ob_space.contains(ob) and ob.dtype == ob_space.dtype
    np.testing.assert_allclose(ob, z[p + "reset_obs"], rtol=0, atol=1e-12)          # the executed reference's reset(seed=0)
    if kind == _capi.ENV_REACH:                                                     # np_random.random(3) * sum(rest_lengths)
        np.testing.assert_array_equal(env._target, z[p + "reset_target"])
        np.testing.assert_array_equal(_capi.muscle_octopus_rest_length_sum(20), sum(env._vec.backend.rods[0].arm(0).get("rest_lengths")))
    else:
        assert env._target.dtype == np.float32 and list(env._target) == [5.0, 0.0]
    a = act_space.sample()
    observation, reward, terminated, truncated, _info = env.step(a)
    assert ob_space.contains(observation) and np.isscalar(reward)
    assert isinstance(terminated, bool) and isinstance(truncated, bool) and isinstance(_info, dict)
    assert env.get_env_info() == json.loads((GOLD / "ref_muscle_octopus_build_records.json").read_text())[name]["init"]["env_info"]
    if kind == _capi.ENV_ARM_TWO:
        pk = env._vec.backend.rods[0]._prev_kappa.copy()
        ob2, _ = env.reset(seed=0)
        n = 19
        np.testing.assert_array_equal(ob2.reshape(2, -1)[:, n:2 * n], pk)           # the previous episode's last kappa
    env.close()
    # batched, host auto-reset past a shortened final_time
    vec = gsa.make_vec(env_id, 2, final_time=0.07, autoreset=True, numpy_output=True,
                       backend=OracleBackend(_capi.muscle_octopus_config(kind, 2, final_time=0.07)))
    vec.reset(seed=4)
    rng = np.random.default_rng(1)
    truncs = []
    for t in range(4):
        o, r, te, tr, inf = vec.step(rng.uniform(0, 0.5, (2, vec.action_dim)).astype(np.float32))
        truncs.append(bool(tr.any()))
        assert o.shape == (2, vec.obs_dim) and np.isfinite(o).all()
    assert truncs[1] and not truncs[0]
    vec.close()


def test_crawl_random_final_time_draws_like_the_reference(oracle_built):
    """CrawlEnv(config_random_final_time=True): `self.final_time = self.np_random.uniform(3.0, 10.0)` is the first draw of
    every reset (crawl_env.py:135-136) — seeds 0, 1, 42 and an unseeded reset continuing seed 42's stream, against the
    executed reference; the batched env draws per env from its own stream and truncation follows the env's OWN final_time."""
    import gym_softrobot_amd as gsa
    from tests.oracle_backend import OracleBackend

    want = np.load(GOLD / "ref_muscle_octopus.npz")["crawl_random_final_time"]
    env = gsa.make("OctoCrawl-v0", config_random_final_time=True, backend=OracleBackend(_capi.muscle_octopus_config(_capi.ENV_CRAWL, 1)))
    got = []
    for seed in (0, 1, 42):
        env.reset(seed=seed)
        got.append(env.final_time)
    env.reset()
    got.append(env.final_time)
    np.testing.assert_array_equal(got, want)
    assert all(3.0 <= f <= 10.0 for f in got)
    env.close()
    vec = gsa.make_vec("OctoCrawl-v0", 3, config_random_final_time=True, numpy_output=True,
                       backend=OracleBackend(_capi.muscle_octopus_config(_capi.ENV_CRAWL, 3)))
    vec.reset(seed=0)
    np.testing.assert_array_equal(vec.final_times[:2], want[:2])           # env i is seeded seed + i
    # put env 0 one step short of ITS final_time, env 1 two steps short of its own: only env 0 truncates
    be = vec.backend
    for i, back in ((0, 0.02), (1, 0.06), (2, 5.0)):
        be.rods[i].body.set_time(float(vec.final_times[i]) - back)
    o, r, te, tr, info = vec.step(np.zeros((3, 24), np.float32))
    assert list(tr) == [True, False, False] and not te.any()
    vec.close()
    plain = gsa.make_vec("OctoCrawl-v0", 2, backend=OracleBackend(_capi.muscle_octopus_config(_capi.ENV_CRAWL, 2)), numpy_output=True)
    plain.reset(seed=0)
    np.testing.assert_array_equal(plain.final_times, [10.0, 10.0])
    plain.close()


def test_symmetric_actuation_leaves_the_head_where_it_is(oracle_built):
    """Known answer for the whole body (oracle; tests/test_gpu_muscle_octopus.py holds the same on the GPU): eight identical
    arms at 45-degree spacing under the same transverse activation with their suckers released — the joint loads on the
    head cancel, it stays put, every arm extends alike."""
    from tests.oracle_mocto import MuscleOctopusOracleEnv

    e = MuscleOctopusOracleEnv(_capi.muscle_octopus_config(_capi.ENV_CRAWL, 1))
    e.reset()
    a = np.tile(np.array([0.0, 0.5, 0.0], np.float32), 8)
    for _ in range(3):
        e.step(a)
    h = e.head()
    assert np.abs(h["x"][:2]).max() < 1e-9 and np.abs(h["v"][:2]).max() < 1e-7
    reach = np.array([np.linalg.norm(e.arm(k).get("x")[:, 20] - e.arm(k).get("x")[:, 0]) for k in range(8)])
    assert reach.min() > 0.2505 and reach.max() - reach.min() < 1e-9
