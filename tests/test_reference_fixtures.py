"""Replay of the fixtures recorded by EXECUTING the reference's own env code
(tools/make_env_golden.py over tools/refshim.py: the reference's __init__ / reset / set_action /
step / get_state and the operator classes its build functions define, run on non-trivial rod
states) through this repo's oracle and host logic.  CPU only; the HIP path replays the same
files in tests/test_gpu_reference_fixtures.py.

What each comparison pins (reference file:line in the section comments):
  * the arguments the reference's build_* functions hand to PyElastica and the ORDER in which
    they register operators  ==  softrod_config defaults and the order switches
  * reset(seed): RNG draw -> rod frame -> reset observation
  * step(): set_action side effects, NaN / blow-up checks, reward, flags, info, observation, on
    the very state the reference's code saw
  * constrain_values / constrain_rates / apply_forces of the env-defined operator classes
"""
import json
from pathlib import Path

import numpy as np
import pytest

GOLD = Path(__file__).parent / "golden"
OBS_TOL = dict(rtol=2e-7, atol=1e-9)      # float32 observations: one ulp (NumPy's pairwise mean vs a loop)
F64_TOL = dict(rtol=1e-12, atol=1e-13)


@pytest.fixture(scope="module")
def records():
    return json.loads((GOLD / "ref_build_records.json").read_text())


def _ops(rec, cls):
    return [o for o in rec["ops"] if o["cls"] == cls]


# ---------------------------------------------------------------------------------------------
# build records: constants and operator order
# ---------------------------------------------------------------------------------------------
def test_softpendulum_build_record_matches_config(records):
    """soft_pendulum.py:59-106 and soft_pendulum/build.py:18-115 as executed."""
    from gym_softrobot_amd import _capi

    rec, cfg = records["SoftPendulum-v0"], _capi.softpendulum_config(1)
    init = rec["init"]
    assert init["step_skip"] == cfg.n_substeps == 400 and init["n_elems"] == cfg.n_elem
    assert init["final_time"] == cfg.final_time and init["time_step"] == cfg.dt
    assert init["action_low"] == [-22.0] and init["action_high"] == [22.0] and init["obs_shape"] == [4]
    rod = _ops(rec, "FakeRod")[0]["recorded"]
    assert rod["n_elements"] == 50 and rod["base_length"] == cfg.base_length and rod["base_radius"] == cfg.base_radius
    assert rod["density"] == cfg.density and rod["youngs_modulus"] == cfg.youngs_modulus
    assert "shear_modulus" not in rod          # -> PyElastica's default, cfg.shear_modulus = E / 3
    assert _ops(rec, "GravityForces")[0]["kwargs"]["acc_gravity"] == list(cfg.gravity)
    d = _ops(rec, "AnalyticalLinearDamper")[0]["kwargs"]
    assert d["damping_constant"] == cfg.damping_constant and d["time_step"] == cfg.dt
    bc = _ops(rec, "PendulumBoundaryConditions")[0]["kwargs"]
    assert bc["constrained_position_idx"] == [0] and bc["constrained_director_idx"] == [0]
    # registration order: constrain before dampen (damp_before_constrain = 0); gravity before the
    # point force, which therefore ASSIGNS over gravity's x component of node 0
    assert rec["order"] == ["append:FakeRod[0]", "constrain:PendulumBoundaryConditions[0]", "forcing:GravityForces[0]",
                            "forcing:PendulumPointForces[0]", "damping:AnalyticalLinearDamper[0]"]
    assert cfg.damp_before_constrain == 0


def test_softpendulum3d_build_record_matches_config(records):
    """soft_pendulum_3d.py:28-58 and soft_pendulum_3d/build.py:43-86 as executed."""
    from gym_softrobot_amd import _capi

    rec, cfg = records["SoftPendulum3D-v0"], _capi.softpendulum3d_config(1)
    init = rec["init"]
    assert init["step_skip"] == cfg.n_substeps and init["base_step"] == cfg.base_step and init["base_limit"] == cfg.base_limit
    assert init["action_low"] == [-1.0, -1.0] and init["action_high"] == [1.0, 1.0]
    rod = _ops(rec, "FakeRod")[0]["recorded"]
    assert (rod["base_length"], rod["base_radius"], rod["density"], rod["youngs_modulus"]) == \
        (cfg.base_length, cfg.base_radius, cfg.density, cfg.youngs_modulus)
    assert "shear_modulus" not in rod
    assert _ops(rec, "GravityForces")[0]["kwargs"]["acc_gravity"] == list(cfg.gravity)
    assert _ops(rec, "AnalyticalLinearDamper")[0]["kwargs"] == {"damping_constant": cfg.damping_constant, "time_step": cfg.dt}
    assert _ops(rec, "LaplaceDissipationFilter")[0]["kwargs"] == {"filter_order": cfg.filter_order}
    # constraint, then the analytical damper, then the Laplace filter
    assert rec["order"] == ["append:FakeRod[0]", "constrain:MovingBaseConstraint[0]", "forcing:GravityForces[0]",
                            "damping:AnalyticalLinearDamper[0]", "damping:LaplaceDissipationFilter[0]"]
    assert rec["bad_action_raises_ValueError"] == [True, True]


def test_armsingle_build_record_matches_config(records):
    """arm_single_env.py:55-113 and octopus/build.py:30-49,220-292 as executed."""
    from gym_softrobot_amd import _capi

    rec, cfg = records["OctoArmSingle-v0"], _capi.arm_single_config(1)
    init = rec["init"]
    assert init["step_skip"] == cfg.n_substeps == 714 and init["final_time"] == cfg.final_time
    assert init["control_penalty_coeff"] == cfg.control_penalty_coeff
    assert init["kappa_range"] == list(cfg.kappa_range) and init["kappa_rate_range"] == list(cfg.kappa_rate_range)
    assert init["target"] == list(cfg.target)
    rod = _ops(rec, "FakeRod")[0]["recorded"]
    assert (rod["base_length"], rod["base_radius"], rod["density"], rod["youngs_modulus"]) == \
        (cfg.base_length, cfg.base_radius, cfg.density, cfg.youngs_modulus)
    assert rod["start"] == [0.0, 0.0, 0.0] and rod["direction"] == [1.0, 0.0, 0.0] and rod["normal"] == [0.0, 0.0, 1.0]
    assert _ops(rec, "GravityForces")[0]["kwargs"]["acc_gravity"] == list(cfg.gravity)
    plane = _ops(rec, "Plane")[0]["recorded"]
    assert plane["plane_origin"] == list(cfg.plane_origin) and plane["plane_normal"] == list(cfg.plane_normal)
    c = _ops(rec, "RodPlaneContactWithAnisotropicFriction")[0]["kwargs"]
    assert (c["k"], c["nu"], c["slip_velocity_tol"]) == (cfg.contact_k, cfg.contact_nu, cfg.slip_velocity_tol)
    assert c["kinetic_mu_array"] == list(cfg.kinetic_mu) and c["static_mu_array"] == list(cfg.static_mu)
    assert _ops(rec, "AnalyticalLinearDamper")[0]["kwargs"] == {"damping_constant": cfg.damping_constant, "time_step": cfg.dt}
    # gravity is registered before the contact: contact_before_forcing = 0
    assert rec["order"] == ["append:FakeRod[0]", "forcing:GravityForces[0]", "append:Plane[1]",
                            "contact:RodPlaneContactWithAnisotropicFriction[0,1]", "damping:AnalyticalLinearDamper[0]"]
    assert cfg.contact_before_forcing == 0


def test_octoflat_build_record_matches_config(records):
    """flat_env.py:55-110 and octopus/build.py:52-217 as executed."""
    from gym_softrobot_amd import _capi

    rec, cfg = records["OctoFlat-v0"], _capi.octo_flat_config(1)
    init = rec["init"]
    assert init["default_step_skip"] == cfg.n_substeps == 2857
    assert (init["n_arm"], init["n_elems"], init["n_action"]) == (cfg.n_arm, cfg.n_elem, cfg.n_knots)
    arms = _ops(rec, "FakeRod")
    assert len(arms) == 8
    pos, dirs = _capi.octo_arm_frames(8, float(cfg.head_radius))
    for a, op in enumerate(arms):
        r = op["recorded"]
        assert (r["base_length"], r["base_radius"], r["density"], r["youngs_modulus"]) == \
            (cfg.base_length, cfg.base_radius, cfg.density, cfg.youngs_modulus)
        np.testing.assert_array_equal(r["start"], pos[a])           # Rotation.from_euler(...).apply, bit for bit
        np.testing.assert_array_equal(r["direction"], dirs[a])
        assert r["normal"] == [0.0, 0.0, 1.0]
    head = _ops(rec, "Cylinder")[0]["recorded"]
    assert head["base_radius"] == cfg.head_radius and head["density"] == cfg.head_density
    assert head["base_length"] == 2 * cfg.base_radius and head["start"] == [0.0, 0.0, -cfg.base_radius]
    assert head["direction"] == [0.0, 0.0, 1.0] and head["normal"] == [0.0, 1.0, 0.0]
    joints = _ops(rec, "FixedJoint2Rigid")
    assert [j["kwargs"]["angle"] for j in joints] == [45.0 * a for a in range(8)]
    assert all((j["kwargs"]["k"], j["kwargs"]["nu"], j["kwargs"]["kt"], j["kwargs"]["radius"]) ==
               (cfg.joint_k, cfg.joint_nu, cfg.joint_kt, cfg.head_radius) for j in joints)
    assert all(j["targets"] == [8, a] for a, j in enumerate(joints))
    c = _ops(rec, "RodPlaneContactWithAnisotropicFriction")[0]["kwargs"]
    assert c["kinetic_mu_array"] == list(cfg.kinetic_mu) and c["static_mu_array"] == list(cfg.static_mu)
    kinds = [o.split(":")[0] for o in rec["order"]]
    # head constraint; the eight joints; gravity per arm; dampers; then the contacts
    first = {k: kinds.index(k) for k in ("constrain", "connect", "forcing", "damping", "contact")}
    assert first["constrain"] < first["connect"] < first["forcing"] < first["damping"] < first["contact"]


# ---------------------------------------------------------------------------------------------
# SoftPendulum-v0: soft_pendulum.py:108-251, soft_pendulum/build.py:46-105
# ---------------------------------------------------------------------------------------------
def test_softpendulum_reset_against_the_reference(oracle_built):
    from gym_softrobot_amd import _capi
    from gym_softrobot_amd.seeding import initial_angle, np_random

    z = np.load(GOLD / "ref_softpendulum.npz")
    cfg = _capi.softpendulum_config(1)
    for i, seed in enumerate(z["reset_seed"]):
        rng, _ = np_random(int(seed))
        th = initial_angle(rng)
        np.testing.assert_array_equal(z["reset_direction"][i], [1.0 * np.cos(th), 1.0 * np.sin(th), 0.0])
        np.testing.assert_array_equal(z["reset_normal"][i], [1.0 * np.sin(th), -1.0 * np.cos(th), 0.0])
        np.testing.assert_array_equal(z["reset_start"][i], 0.0)
        rod = oracle_built.OracleRod(cfg)
        rod.reset_pendulum(th)
        np.testing.assert_allclose(rod.observe(), z["reset_obs"][i], **OBS_TOL)
        # what finalize() hands the constraint: position / directors of node / element 0
        np.testing.assert_array_equal(rod.get("fixed_pos"), z["reset_bc_fixed_position"][i])
        np.testing.assert_array_equal(rod.get("fixed_dir"), z["reset_bc_fixed_directors"][i])


def _load_rod(rod, z, i, pre="step_"):
    for name, key in (("x", "x"), ("v", "v"), ("Q", "Q"), ("w", "w"), ("tangents", "tangents")):
        rod.set(name, z[pre + key][i])
    rod.set("time", [z[pre + "time"][i]])


def test_softpendulum_step_epilogue_against_the_reference(oracle_built):
    from gym_softrobot_amd import _capi

    z = np.load(GOLD / "ref_softpendulum.npz")
    cfg = _capi.softpendulum_config(1)
    assert {"nan_x", "nan_v", "time_just_past", "theta_q3"} <= set(z["step_label"])
    for i, label in enumerate(z["step_label"]):
        rod = oracle_built.OracleRod(cfg)
        rod.reset_pendulum(1.5)
        rod.set_run_substeps(0)          # set_action + the epilogue alone, on the state the reference saw
        _load_rod(rod, z, i)
        rod.set("prev_action", [z["step_prev_action_before"][i][0]])
        obs, rew, term, trunc = rod.env_step(z["step_action"][i])
        np.testing.assert_allclose(obs, z["step_obs"][i], err_msg=str(label), **OBS_TOL)
        np.testing.assert_allclose(rew, z["step_reward"][i], err_msg=str(label), **F64_TOL)
        assert (term, trunc) == (bool(z["step_terminated"][i]), bool(z["step_truncated"][i])), label
        assert rod.time == z["step_info_time"][i] and trunc == bool(z["step_info_trunc"][i])
        # set_action: the float32 action lands in the float64 mailbox and in _prev_action
        assert z["step_point_force"][i] == np.float64(z["step_action"][i])
        assert z["step_prev_action_after"][i][0] == z["step_action"][i]


def test_pendulum_operators_against_the_reference_classes(oracle_built):
    """PendulumBoundaryConditions.constrain_values / constrain_rates (build.py:71-79) and
    PendulumPointForces.apply_forces (:100-101) evaluated by the reference's own classes."""
    from gym_softrobot_amd import _capi

    z = np.load(GOLD / "ref_softpendulum.npz")
    cfg = _capi.softpendulum_config(1)
    cfg.features = _capi.FEAT_PENDULUM_BC | _capi.FEAT_POINT_FORCE_NODE0_X      # no gravity: f_in already holds it
    for i in range(len(z["op_force"])):
        rod = oracle_built.OracleRod(cfg)
        rod.reset_pendulum(1.5)
        for name, key in (("x", "x_in"), ("v", "v_in"), ("Q", "Q_in"), ("w", "w_in"), ("f_ext", "f_in")):
            rod.set(name, z["op_" + key][i])
        rod.set("fixed_pos", z["op_fixed_position"][i])
        rod.set("fixed_dir", z["op_fixed_directors"][i])
        rod.constrain_probe()
        rod.forcing_probe(float(z["op_force"][i]))
        for name, key in (("x", "x_out"), ("v", "v_out"), ("Q", "Q_out"), ("w", "w_out"), ("f_ext", "f_out")):
            np.testing.assert_array_equal(rod.get(name), z["op_" + key][i], err_msg=name)
        # row 1 of the held director is NOT reset (build.py:73-74), the point force assigns
        assert not np.array_equal(z["op_Q_out"][i][1, :, 0], z["op_fixed_directors"][i][1])
        assert z["op_f_out"][i][0, 0] == z["op_force"][i] != z["op_f_in"][i][0, 0]


# ---------------------------------------------------------------------------------------------
# SoftPendulum3D-v0: soft_pendulum_3d.py:60-174, soft_pendulum_3d/build.py:15-64
# ---------------------------------------------------------------------------------------------
def test_softpendulum3d_reset_against_the_reference(oracle_built):
    from gym_softrobot_amd import _capi
    from gym_softrobot_amd.seeding import np_random

    z = np.load(GOLD / "ref_softpendulum3d.npz")
    cfg = _capi.softpendulum3d_config(1)
    for i, seed in enumerate(z["reset_seed"]):
        rng, _ = np_random(int(seed))
        tilt = np.deg2rad(rng.uniform(-1.0, 1.0))
        np.testing.assert_array_equal(z["reset_direction"][i], [np.sin(tilt), 0.0, np.cos(tilt)])
        np.testing.assert_array_equal(z["reset_normal"][i], [0.0, 1.0, 0.0])
        rod = oracle_built.OracleRod(cfg)
        rod.reset_pendulum3d(tilt)
        np.testing.assert_allclose(rod.observe3d(), z["reset_obs"][i], **OBS_TOL)
        assert np.all(z["reset_obs"][i][6:8] == 0.0)       # _prev_action cleared by reset (:68)


def test_softpendulum3d_step_against_the_reference(oracle_built):
    from gym_softrobot_amd import _capi

    z = np.load(GOLD / "ref_softpendulum3d.npz")
    cfg = _capi.softpendulum3d_config(1)
    assert {"clip_hi", "clip_lo", "nan_v", "time_eq_final", "tilt_down"} <= set(z["step_label"])
    for i, label in enumerate(z["step_label"]):
        rod = oracle_built.OracleRod(cfg)
        rod.reset_pendulum3d(0.0)
        rod.set_run_substeps(0)          # set_action still divides by the real step_skip * time_step
        _load_rod(rod, z, i)
        rod.set("control", z["step_ctrl_before"][i])
        rod.set("prev_action2", z["step_prev_action_before"][i])
        obs, rew, term, trunc, tilt = rod.env_step3d(z["step_action"][i])
        np.testing.assert_allclose(obs, z["step_obs"][i], err_msg=str(label), **OBS_TOL)
        np.testing.assert_allclose(rew, z["step_reward"][i], err_msg=str(label), **F64_TOL)
        np.testing.assert_allclose(tilt, z["step_info_tilt"][i], err_msg=str(label), **F64_TOL)
        assert (term, trunc) == (bool(z["step_terminated"][i]), bool(z["step_truncated"][i])), label
        np.testing.assert_allclose(rod.get("control"), z["step_ctrl_after"][i], err_msg=str(label), rtol=1e-15, atol=0)
        assert z["step_ctrl_pos_z"][i] == 0.0


def test_moving_base_constraint_against_the_reference_class(oracle_built):
    """MovingBaseConstraint.constrain_values / constrain_rates (soft_pendulum_3d/build.py:31-39)."""
    from gym_softrobot_amd import _capi

    z = np.load(GOLD / "ref_softpendulum3d.npz")
    cfg = _capi.softpendulum3d_config(1)
    for i in range(len(z["op_ctrl"])):
        rod = oracle_built.OracleRod(cfg)
        rod.reset_pendulum3d(0.0)
        for name, key in (("x", "x_in"), ("v", "v_in"), ("Q", "Q_in"), ("w", "w_in")):
            rod.set(name, z["op_" + key][i])
        rod.set("control", z["op_ctrl"][i])
        rod.set("fixed_pos", [0.0, 0.0, z["op_fixed_height"][i]])
        rod.set("fixed_dir", z["op_fixed_director"][i])
        rod.constrain_probe()
        for name, key in (("x", "x_out"), ("v", "v_out"), ("Q", "Q_out"), ("w", "w_out")):
            np.testing.assert_array_equal(rod.get(name), z["op_" + key][i], err_msg=name)


# ---------------------------------------------------------------------------------------------
# OctoArmSingle-v0: arm_single_env.py:135-316
# ---------------------------------------------------------------------------------------------
def test_armsingle_reset_and_step_against_the_reference(oracle_built):
    from gym_softrobot_amd import _capi

    z = np.load(GOLD / "ref_armsingle.npz")
    cfg = _capi.arm_single_config(1)
    rod = oracle_built.OracleRod(cfg)
    obs0 = rod.reset_arm()
    np.testing.assert_allclose(obs0, z["reset_obs"], **OBS_TOL)
    np.testing.assert_allclose(rod.get("prev_com"), z["reset_prev_com"], **F64_TOL)
    np.testing.assert_array_equal(rod.get("prev_kappa"), z["reset_prev_kappa"])
    np.testing.assert_allclose(rod.get("mass"), z["mass"], rtol=1e-15)
    assert {"nan_v", "omega_blown", "omega_at_threshold", "at_target", "time_just_past"} <= set(z["step_label"])
    for i, label in enumerate(z["step_label"]):
        rod = oracle_built.OracleRod(cfg)
        rod.reset_arm()
        rod.set_run_substeps(0)
        _load_rod(rod, z, i)
        rod.set("kappa", z["step_kappa"][i])
        rod.set("prev_kappa", z["step_prev_kappa_before"][i])
        rod.set("prev_com", z["step_prev_com_before"][i])
        rod.set("prev_action7", z["step_prev_action_before"][i])
        obs, rew, term, trunc = rod.env_step_arm(z["step_action"][i])
        np.testing.assert_allclose(obs, z["step_obs"][i], err_msg=str(label), rtol=2e-6, atol=2e-7)
        np.testing.assert_allclose(rew, z["step_reward"][i], err_msg=str(label), rtol=1e-9, atol=1e-12)
        assert (term, trunc) == (bool(z["step_terminated"][i]), bool(z["step_truncated"][i])), label
        # set_action: rest_kappa[0, :] = cubic interp1d of the action; rows 1, 2 untouched
        np.testing.assert_allclose(rod.get("rest_kappa"), z["step_rest_kappa"][i], rtol=1e-14, atol=1e-14)
        if not np.isnan(z["step_obs"][i]).any():
            np.testing.assert_allclose(rod.get("prev_com"), z["step_prev_com_after"][i], **F64_TOL)
        np.testing.assert_array_equal(rod.get("prev_kappa"), z["step_prev_kappa_after"][i])
    lab = list(z["step_label"])
    assert bool(z["step_terminated"][lab.index("omega_blown")]) and not bool(z["step_terminated"][lab.index("omega_at_threshold")])
    assert bool(z["step_terminated"][lab.index("at_target")]) and z["step_reward"][lab.index("at_target")] > 4.0


def test_armsingle_action_basis_against_the_reference_set_action():
    """The HIP path applies set_action's interp1d as a constant matrix; the reference's own
    set_action output (arm_single_env.py:226-235) must equal basis @ action."""
    from gym_softrobot_amd import _capi

    z = np.load(GOLD / "ref_armsingle.npz")
    W = _capi.action_basis(50, 7)
    for i in range(len(z["step_label"])):
        a = z["step_action"][i].astype(np.float64)
        np.testing.assert_allclose(W @ a, z["step_rest_kappa"][i][0], rtol=1e-12, atol=1e-12)
        assert np.all(z["step_rest_kappa"][i][1:] == 0.0)


# ---------------------------------------------------------------------------------------------
# OctoFlat-v0: flat_env.py:171-408
# ---------------------------------------------------------------------------------------------
FLAT_FPS = 357


def _load_octo(o, z, i, which):
    p = f"step_{which}_"
    for a in range(o.n_arm):
        arm = o.arm(a)
        for name in ("x", "v", "Q", "w", "kappa"):
            arm.set(name, z[p + name][i][a])
    o.set_head(z[p + "head_x"][i], z[p + "head_v"][i], z[p + "head_Q"][i], z[p + "head_w"][i])


def test_octoflat_reset_against_the_reference(oracle_built):
    from gym_softrobot_amd import _capi
    from gym_softrobot_amd.seeding import np_random

    z = np.load(GOLD / "ref_octoflat.npz")
    cfg = _capi.octo_flat_config(1, recording_fps=FLAT_FPS)
    for i, seed in enumerate(z["reset_seed"]):
        rng, _ = np_random(int(seed))
        target = (2 - 0.5) * rng.random(2) + 0.5
        np.testing.assert_array_equal(target, z["reset_target"][i])
        o = oracle_built.OracleOcto(cfg)
        ob = o.reset(target)
        np.testing.assert_allclose(ob["individual"], z["reset_individual"][i], **OBS_TOL)
        np.testing.assert_allclose(ob["shared"], z["reset_shared"][i], **OBS_TOL)


def test_octoflat_step_epilogue_against_the_reference(oracle_built):
    from gym_softrobot_amd import _capi

    z = np.load(GOLD / "ref_octoflat.npz")
    cfg = _capi.octo_flat_config(1, recording_fps=FLAT_FPS)
    assert {"nan_x", "at_target", "time_just_past", "crossing"} <= set(z["step_label"])
    for i, label in enumerate(z["step_label"]):
        o = oracle_built.OracleOcto(cfg)
        o.reset(z["step_target"][i])
        _load_octo(o, z, i, "post")
        o.set_time(z["step_time"][i])
        ob, rew, term, trunc = o.epilogue_probe(z["step_action"][i], z["step_pre_head_x"][i][:2])
        np.testing.assert_allclose(ob["individual"], z["step_individual"][i], err_msg=str(label), **OBS_TOL)
        np.testing.assert_allclose(ob["shared"], z["step_shared"][i], err_msg=str(label), **OBS_TOL)
        np.testing.assert_allclose(rew, z["step_reward"][i], err_msg=str(label), rtol=1e-9, atol=1e-9)
        assert (term, trunc) == (bool(z["step_terminated"][i]), bool(z["step_truncated"][i])), label
    lab = list(z["step_label"])
    # -0.02 per crossing: arm 1 x arm 0 and arm 7 x arm 0 count, arm 6 x arm 7 does not (the pair
    # (6, 7) is never tested, flat_env.py:347-357); same head motion and target as "time_eq_final"
    diff = z["step_reward"][lab.index("crossing")] - z["step_reward"][lab.index("time_eq_final")]
    assert diff == pytest.approx(-0.04, abs=1e-12)


def test_octoflat_full_step_from_the_pre_state(oracle_built):
    """set_action (zero-padded cubic interp1d per arm, flat_env.py:288-311) + 40 substeps + the
    epilogue from the recorded pre-loop state: the oracle lands on the recorded post-loop state
    and returns what the reference's step() returned for it."""
    from gym_softrobot_amd import _capi

    z = np.load(GOLD / "ref_octoflat.npz")
    cfg = _capi.octo_flat_config(1, recording_fps=FLAT_FPS)
    for i, label in enumerate(z["step_label"]):
        if not str(label).startswith("rollout"):
            continue
        o = oracle_built.OracleOcto(cfg)
        o.reset(z["step_target"][i])
        _load_octo(o, z, i, "pre")
        for a in range(o.n_arm):
            o.arm(a).set("rest_kappa", z["step_pre_rest_kappa"][i][a])
        o.set_time(z["step_time"][i] - 40 * cfg.dt)
        ob, rew, term, trunc = o.env_step(z["step_action"][i])
        for a in range(o.n_arm):
            np.testing.assert_allclose(o.arm(a).get("rest_kappa"), z["step_rest_kappa"][i][a], rtol=1e-13, atol=1e-13)
            np.testing.assert_allclose(o.arm(a).get("x"), z["step_post_x"][i][a], rtol=1e-9, atol=1e-12)
        np.testing.assert_allclose(ob["individual"], z["step_individual"][i], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(rew, z["step_reward"][i], rtol=1e-6, atol=1e-7)
        assert (term, trunc) == (bool(z["step_terminated"][i]), bool(z["step_truncated"][i]))


# ---------------------------------------------------------------------------------------------
# OctoFlatLite-v0: FlatEnv registered with n_arm = 1, n_action = 8 (gym_softrobot/__init__.py:11-15)
# ---------------------------------------------------------------------------------------------
def test_octoflatlite_against_the_reference(oracle_built, records):
    from gym_softrobot_amd import _capi
    from gym_softrobot_amd.seeding import np_random

    z = np.load(GOLD / "ref_octoflatlite.npz")
    rec = records["OctoFlatLite-v0"]
    assert (rec["init"]["n_arm"], rec["init"]["n_action"], rec["init"]["default_step_skip"]) == (1, 8, 2857)
    assert rec["init"]["action_low"] == [-22.0] * 8
    cfg = _capi.octo_flat_config(1, recording_fps=FLAT_FPS, n_arm=1, n_action=8)
    for i, seed in enumerate(z["reset_seed"]):
        rng, _ = np_random(int(seed))
        target = (2 - 0.5) * rng.random(2) + 0.5
        np.testing.assert_array_equal(target, z["reset_target"][i])
        o = oracle_built.OracleOcto(cfg)
        ob = o.reset(target)
        assert ob["individual"].shape == (1, 9 + 44 + 8)
        np.testing.assert_allclose(ob["individual"], z["reset_individual"][i], **OBS_TOL)
        np.testing.assert_allclose(ob["shared"], z["reset_shared"][i], **OBS_TOL)
    for i, label in enumerate(z["step_label"]):
        # the epilogue alone on the recorded post-loop state ...
        o = oracle_built.OracleOcto(cfg)
        o.reset(z["step_target"][i])
        arm = o.arm(0)
        for name in ("x", "v", "Q", "w", "kappa"):
            arm.set(name, z["step_post_" + name][i][0])
        o.set_head(z["step_post_head_x"][i], z["step_post_head_v"][i], z["step_post_head_Q"][i], z["step_post_head_w"][i])
        o.set_time(z["step_time"][i])
        ob, rew, term, trunc = o.epilogue_probe(z["step_action"][i], z["step_pre_head_x"][i][:2])
        np.testing.assert_allclose(ob["individual"], z["step_individual"][i], err_msg=str(label), **OBS_TOL)
        np.testing.assert_allclose(ob["shared"], z["step_shared"][i], err_msg=str(label), **OBS_TOL)
        np.testing.assert_allclose(rew, z["step_reward"][i], rtol=1e-9, atol=1e-9)
        assert (term, trunc) == (bool(z["step_terminated"][i]), bool(z["step_truncated"][i]))
        # ... and set_action: 8 knots zero-padded to 10, cubic interp1d onto the 9 Voronoi vertices
        W = _capi.octo_action_basis(10, 8)
        np.testing.assert_allclose(W @ z["step_action"][i].astype(np.float64), z["step_rest_kappa"][i][0][0],
                                   rtol=1e-12, atol=1e-12)


def test_diagnostic_tap_fields_match_the_reference_callbacks(records):
    """RodCallBack / RigidCylinderCallBack (utils/custom_elastica/callback_func.py:4-41), executed:
    the field sets and the firing rule (`current_step % step_skip == 0`: once per env.step) that
    gym_softrobot_amd.diagnostics reproduces from the resident state."""
    from gym_softrobot_amd.diagnostics import RodRecorder

    cb = records["callbacks"]
    assert tuple(cb["RodCallBack"]["fields"]) == RodRecorder.FIELDS
    assert cb["RodCallBack"]["fires_at_steps_with_step_skip_4"] == [0, 4, 8, 12]
    assert cb["RodCallBack"]["shapes"]["position"] == [3, 7] and cb["RodCallBack"]["shapes"]["director"] == [3, 3, 6]
    assert cb["RodCallBack"]["shapes"]["kappa"] == [3, 5] and cb["RodCallBack"]["shapes"]["sigma"] == [3, 6]
    assert cb["RigidCylinderCallBack"]["fields"] == ["time", "step", "position", "velocity"]
    assert cb["RigidCylinderCallBack"]["shapes"]["position"] == [3, 1]
