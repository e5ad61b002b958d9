"""SURVEY.md §8(f) N3 on the CPU: the oracle's COOMM muscle layers and OctoArmPush-v0 / -v1.

Three kinds of evidence, in decreasing strength:
  1. PINNED against the executed reference: everything gym_softrobot/envs/octopus/arm_push_env.py:52-347 and
     create_es_muscle_layers (octopus/build.py:295-338) do themselves — geometry, material, operator order, the
     layers' constructor arguments, set_action, the NaN check, reward, truncation, get_state — replayed from
     tests/golden/ref_armpush.npz / ref_muscle_build_records.json (tools/make_muscle_env_golden.py ran the
     reference's files under recording stand-ins for COOMM).
  2. KNOWN ANSWERS that do not depend on anyone's recollection of COOMM's source: a constant muscle force at an
     offset is statically equivalent to the end couple F r_m plus the axial end force -F (the discrete equations
     of tests/elastica_chain.py: S sigma / e = -f, B kappa / eps^3 = -c) -> a circular arc; a transverse layer ->
     uniform stretch; the force-length polynomial and its clip as a formula.
  3. UNPINNED: the muscle law itself (COOMM, uv.lock:173-175, is not on disk).  Two transcriptions of the recalled
     algorithm — the C oracle's per-element loops and oracle/softrod_oracle_np.py's whole-array form — agree to
     1e-12 for every setting of the recalled-detail switches; that guards against transcription slips, not
     against a wrong recollection."""
import json
from pathlib import Path

import numpy as np
import pytest

from gym_softrobot_amd import _capi

GOLD = Path(__file__).resolve().parent / "golden"
N_ELEM = 40


def _push_rod(oracle_c, mode="discrete", **cfg_kw):
    cfg = _capi.arm_push_config(1, mode=mode)
    for k, v in cfg_kw.items():
        setattr(cfg, k, v)
    rod = oracle_c.OracleRod(cfg)
    radii = _capi.arm_push_radii(N_ELEM)
    rod.set_radius_profile(radii)
    rod.set_muscle_layers(*_capi.es_muscle_layers(radii, 0.012))
    rod.reset_push()
    return rod, cfg


# ---- 1. pinned against the executed reference -------------------------------------------------------------------
def test_build_arguments_match_the_executed_reference():
    rec = json.loads((GOLD / "ref_muscle_build_records.json").read_text())
    for mode in ("discrete", "continuous"):
        r = rec[f"OctoArmPush ({mode})"]
        cfg = _capi.arm_push_config(1, mode=mode)
        init, rod = r["init"], r["straight_rod"]
        assert (init["step_skip"], init["n_elem"], init["obs_shape"]) == (cfg.n_substeps, cfg.n_elem, [_capi.config_obs_dim(cfg)])
        assert (init["final_time"], init["time_step"], init["mode"]) == (cfg.final_time, cfg.dt, cfg.arm_push_mode)
        assert rod["n_elements"] == cfg.n_elem and rod["base_length"] == cfg.base_length and rod["density"] == cfg.density
        assert rod["youngs_modulus"] == cfg.youngs_modulus and rod["shear_modulus"] == cfg.shear_modulus
        np.testing.assert_array_equal(rod["base_radius"], _capi.arm_push_radii(N_ELEM))
        assert (rod["start"], rod["direction"], rod["normal"]) == ([0.0, 0.0, 0.0], [1.0, 0.0, 0.0], [0.0, 1.0, -0.0])
        # operator registration order: the damper BEFORE the sucker constraint, the muscles as a forcing; nothing else
        assert r["order"] == ["append:FakeRod[0]", "damping:AnalyticalLinearDamper[0]",
                              "constrain:ControllableFixConstraint[0]", "forcing:ApplyMuscles[0]"]
        assert cfg.damp_before_constrain == 1 and cfg.features == _capi.FEATURES_ARM_PUSH
        ops = {o["cls"]: o["kwargs"] for o in r["ops"]}
        assert ops["AnalyticalLinearDamper"] == {"damping_constant": cfg.damping_constant, "time_step": cfg.dt}
        assert ops["ControllableFixConstraint"] == {"index": cfg.sucker_index[0]} and cfg.n_suckers == 1
        assert ops["ApplyMuscles"]["step_skip"] == cfg.n_substeps
        assert r["sucker"] == {"index": 0, "flag": True, "reduction_ratio": cfg.sucker_reduction_ratio}
        # the layers as create_es_muscle_layers hands them to COOMM's constructors
        layers = r["muscle_layers"]
        assert [l["kind"] for l in layers] == ["LongitudinalMuscle", "LongitudinalMuscle", "TransverseMuscle"]
        assert [cfg.muscle_kind[m] for m in range(3)] == [_capi.MUSCLE_LONGITUDINAL, _capi.MUSCLE_LONGITUDINAL, _capi.MUSCLE_TRANSVERSE]
        radii = _capi.arm_push_radii(N_ELEM)
        raw, strength = _capi.es_muscle_layers(radii, 0.012, init_angle_rotates=False, tm_sign=1.0)
        assert [layers[0]["muscle_init_angle"], layers[1]["muscle_init_angle"]] == [np.pi / 2, -np.pi / 2]
        for m in (0, 1):
            np.testing.assert_array_equal(layers[m]["ratio_muscle_position"], raw[m])
        for m in range(3):
            np.testing.assert_array_equal(np.asarray(layers[m]["rest_muscle_area"]) * layers[m]["max_muscle_stress"], strength[m])
        # what the recalled COOMM constructors make of them (the defaults of _capi.es_muscle_layers)
        ratio, strength = _capi.es_muscle_layers(radii, 0.012)
        np.testing.assert_allclose(ratio[0, 0], 2 / 3, rtol=1e-15)
        np.testing.assert_allclose(ratio[1, 0], -2 / 3, rtol=1e-15)
        assert np.abs(ratio[:2, 1:]).max() < 1e-16 and not ratio[2].any() and (strength[2] < 0).all()


@pytest.mark.parametrize("mode", ["discrete", "continuous"])
def test_reset_and_step_replay_the_executed_reference(oracle_built, mode):
    """The reference's reset observation and its step() around a scripted stepper — set_action (sucker index,
    activations), prev_cm_pos, the NaN check, reward, truncation, get_state with nan_to_num — through the
    oracle's env_step_push with run_substeps = 0 (the epilogue on the installed state)."""
    z = np.load(GOLD / "ref_armpush.npz")
    p = "d_" if mode == "discrete" else "c_"
    rod, cfg = _push_rod(oracle_built, mode)
    np.testing.assert_array_equal(rod.observe_push(), z[p + "reset_obs"])
    mass = rod.get("mass")
    rod.set_run_substeps(0)
    labels = [str(s) for s in z[p + "step_label"]]
    assert {"nan_x", "nan_w", "nan_Q", "inf_v0", "time_eq_final", "time_just_past", "shifted0"} <= set(labels)
    for i, label in enumerate(labels):
        if label == "nan_alpha":
            # alpha_collection is not part of the state: in the stepper a NaN angular acceleration makes omega NaN in
            # the same substep (omega += dt alpha), which the omega check catches — the fixture holds the reference's
            # answer for the synthetic case (terminated, -20) for completeness
            assert bool(z[p + "step_terminated"][i]) and z[p + "step_reward"][i] == -20.0
            continue
        pre_x = z[p + "step_pre_x"][i]
        act_before = rod.get("muscle_activation").copy()
        rod.set("x", z[p + "step_x"][i]); rod.set("v", z[p + "step_v"][i])
        rod.set("Q", z[p + "step_Q"][i]); rod.set("w", z[p + "step_w"][i])
        rod.set("time", np.array([z[p + "step_time"][i]]))
        rod.set("prev_com", (pre_x[:2] * mass).sum(axis=1) / mass.sum())
        action = z[p + "step_action"][i]
        obs, rew, term, trunc = rod.env_step_push(action[:1] if mode == "discrete" else action)
        np.testing.assert_array_equal(obs, z[p + "step_obs"][i], err_msg=label)
        np.testing.assert_allclose(rew, z[p + "step_reward"][i], rtol=1e-12, atol=1e-15, err_msg=label)
        assert (term, trunc) == (bool(z[p + "step_terminated"][i]), bool(z[p + "step_truncated"][i])), label
        assert int(rod.get("sucker_index")[0]) == int(z[p + "step_sucker_index"][i]), label
        want = z[p + "step_activations"][i]                  # NaN: this layer's apply_activation was not called
        got = rod.get("muscle_activation")
        for m in range(3):
            np.testing.assert_array_equal(got[m], np.full(N_ELEM, want[m]) if np.isfinite(want[m]) else act_before[m],
                                          err_msg=f"{label} layer {m}")


def test_sucker_constraint_with_python_indexing(oracle_built):
    """ControllableFixConstraint.constrain_rates executed from the reference's file with the indices set_action
    produces: -1 is the LAST NODE of velocity_collection and the LAST ELEMENT of omega_collection."""
    z = np.load(GOLD / "ref_armpush.npz")
    rod, cfg = _push_rod(oracle_built)
    for i in range(len(z["sucker_index"])):
        # the controller the constraint really holds (a controller that is OFF at construction is replaced by a fresh
        # one that is on, `controller or SuckerController(...)` with SuckerController.__bool__ = flag: a quirk of
        # controllable_constraint.py:28-30 the envs never meet — theirs are on when the constraint is built)
        idx, ratio, on = int(z["sucker_effective_index"][i]), float(z["sucker_effective_ratio"][i]), bool(z["sucker_effective_flag"][i])
        rod.set("v", z["sucker_v_in"][i]); rod.set("w", z["sucker_w_in"][i])
        rod.set("sucker_index", np.array([idx, 0, 0, 0], np.float64))
        rod.set_sucker_ratio([ratio if on else 0.0])          # a controller that is off = an effective ratio of 0
        rod.constrain_probe()
        np.testing.assert_array_equal(rod.get("v"), z["sucker_v_out"][i])
        np.testing.assert_array_equal(rod.get("w"), z["sucker_w_out"][i])
    assert [str(o) for o in z["sucker_off"]].count("construction") == 1 and "later" in [str(o) for o in z["sucker_off"]]
    k = [str(o) for o in z["sucker_off"]].index("construction")
    assert (int(z["sucker_effective_index"][k]), float(z["sucker_effective_ratio"][k])) == (2, 1.0)
    idx = [int(k) for k in z["sucker_index"]]
    assert -1 in idx and 0 in idx and 39 in idx


# ---- 2. known answers ------------------------------------------------------------------------------------------
def _uniform_muscle_rod(oracle_c, kind, form=0, n=12):
    L, r0, E, rho = 0.2, 0.01, 2e6, 700.0
    cfg = _capi.softpendulum_config(1, n_elems=n)
    cfg.env_kind = _capi.ENV_NONE
    cfg.features = _capi.FEAT_FIXED_BC | _capi.FEAT_ANALYTICAL_DAMPER | _capi.FEAT_COOMM_MUSCLES
    cfg.base_length, cfg.base_radius, cfg.density, cfg.youngs_modulus, cfg.shear_modulus = L, r0, rho, E, E / 1.5
    cfg.dt, cfg.damping_constant, cfg.damper_protocol = 5e-5, 45.0, 1      # critical damping of the bending mode
    _capi.muscle_defaults(cfg)
    cfg.n_muscles = 1
    cfg.muscle_kind[0] = kind
    cfg.muscle_fl_degree = 0
    cfg.muscle_fl_coef[0] = 1.0                                            # fl = 1: a constant force
    cfg.muscle_equiv_load_form = form
    rod = oracle_c.OracleRod(cfg)
    return rod, cfg, (L, r0, E, np.pi * r0 ** 2, np.pi * r0 ** 4 / 4)


@pytest.mark.parametrize("form", [0, 1])
def test_known_answer_longitudinal_force_is_an_end_couple(oracle_built, form):
    """A constant force F along a muscle at x_m = 0.6 r e_1: each cross-section carries -F t_m and x_m x (-F t_m),
    the internal loads of a rod under the END couple F r_m and the axial end force -F.  Discrete equilibrium
    (tests/elastica_chain.py's equations), form 0: S sigma / e = -F -> e = 1 / (1 + F / EA), B kappa / eps^3 = F r_m
    -> every joint turns by kappa l^ with kappa = F r_m e^3 / EI, r_m = 0.6 r0 / sqrt(e) (the radius follows the
    stretch); form 1 (PyElastica's internal-load form on both sides): e = 1 - F / EA, kappa = F r_m / EI."""
    n = 12
    rod, cfg, (L, r0, E, A, I) = _uniform_muscle_rod(oracle_built, _capi.MUSCLE_LONGITUDINAL, form, n)
    ratio = np.zeros((1, 3, n)); ratio[0, 0] = 0.6
    F = 3.0
    rod.set_muscle_layers(ratio, np.full((1, n), F))
    rod.reset_straight(np.zeros(3), np.array([1.0, 0.0, 0.0]), np.array([0.0, 1.0, 0.0]))
    rod.apply_activation(0, 1.0)
    rod.substeps(0.0, 30000)
    assert np.abs(rod.get("v")).max() < 1e-11
    x, Q = rod.get("x"), rod.get("Q")
    e = 1.0 / (1.0 + F / (E * A)) if form == 0 else 1.0 - F / (E * A)
    kappa = F * (0.6 * r0 / np.sqrt(e)) * (e ** 3 if form == 0 else 1.0) / (E * I)
    lens = np.sqrt(((x[:, 1:] - x[:, :-1]) ** 2).sum(axis=0))
    np.testing.assert_allclose(lens / (L / n), e, rtol=1e-12)
    heading = np.arctan2(Q[2, 1], Q[2, 0])
    np.testing.assert_allclose(np.diff(heading), kappa * L / n, rtol=1e-9)
    np.testing.assert_allclose(rod.get("kappa")[1], kappa, rtol=1e-9)      # about d2 = d3 x d1
    # tip of the discrete arc: element k at heading k kappa l^, length e l^ (element 0 is clamped along x)
    th = kappa * (L / n) * np.arange(n)
    np.testing.assert_allclose(x[:2, -1], [e * L / n * np.cos(th).sum(), e * L / n * np.sin(th).sum()], rtol=1e-9)
    assert np.abs(x[2]).max() < 1e-14


def test_known_answer_transverse_layer_stretches_uniformly(oracle_built):
    """A transverse layer (on the axis, negative strength): S sigma / e = +|F| -> e = 1 / (1 - |F| / EA), straight."""
    n = 12
    rod, cfg, (L, r0, E, A, I) = _uniform_muscle_rod(oracle_built, _capi.MUSCLE_TRANSVERSE, 0, n)
    rod.set_muscle_layers(np.zeros((1, 3, n)), np.full((1, n), -25.0))
    rod.reset_straight(np.zeros(3), np.array([1.0, 0.0, 0.0]), np.array([0.0, 1.0, 0.0]))
    rod.apply_activation(0, 0.8)
    rod.substeps(0.0, 30000)
    e = 1.0 / (1.0 - 0.8 * 25.0 / (E * A))
    np.testing.assert_allclose(rod.get("x")[0], np.linspace(0.0, L * e, n + 1), rtol=1e-11, atol=1e-15)
    assert np.abs(rod.get("x")[1:]).max() == 0.0 and e > 1.03


def test_force_length_law_is_the_published_cubic_with_its_clip(oracle_built):
    """F_m = u sigma_max A max{3.06 l^3 - 13.64 l^2 + 18.01 l - 6.44, 0} (Chang et al. 2023): the transverse layer of
    the ArmPush arm at prescribed stretches, l = 1 / sqrt(e); the law is ~1 at rest length and clips to zero."""
    rod, cfg = _push_rod(oracle_built)
    assert [cfg.muscle_fl_coef[k] for k in range(4)] == [-6.44, 18.01, -13.64, 3.06] and cfg.muscle_fl_degree == 3
    radii = _capi.arm_push_radii(N_ELEM)
    area = (radii / 0.012) ** 2
    rod.apply_activation(2, 0.7)
    for stretch in (1.0, 1.1, 0.8, 3.5, 0.3):
        sig = np.zeros((3, N_ELEM)); sig[2] = stretch - 1.0
        rod.set("sigma", sig)
        rod.muscle_probe()
        l = 1.0 / np.sqrt(stretch)
        w = max(3.06 * l ** 3 - 13.64 * l ** 2 + 18.01 * l - 6.44, 0.0)
        np.testing.assert_allclose(rod.get("muscle_length")[2], l, rtol=1e-15)
        np.testing.assert_allclose(rod.get("muscle_force")[2], 0.7 * (-1.0 * area) * w, rtol=1e-13, atol=1e-18)
    assert abs(3.06 - 13.64 + 18.01 - 6.44 - 0.99) < 1e-12       # ~1 at rest length
    assert max(3.06 * 0.5345 ** 3 - 13.64 * 0.5345 ** 2 + 18.01 * 0.5345 - 6.44, 0.0) == 0.0     # stretch 3.5: clipped


# ---- 3. two transcriptions of the recalled algorithm ---------------------------------------------------------------
@pytest.mark.parametrize("form,current_radius,tm_law", [(0, 1, 0), (1, 1, 0), (0, 0, 0), (0, 1, 1), (1, 0, 1)])
def test_c_oracle_equals_the_numpy_twin(oracle_built, form, current_radius, tm_law):
    from oracle import softrod_oracle_np as onp

    rod, cfg = _push_rod(oracle_built, muscle_equiv_load_form=form, muscle_position_current_radius=current_radius,
                         muscle_tm_length_law=tm_law)
    radii = _capi.arm_push_radii(N_ELEM)
    ratio, strength = _capi.es_muscle_layers(radii, 0.012)
    th = 0.4                                                   # off the d1 axis: couples on d1 and d2
    ratio[0, 0], ratio[0, 1] = np.cos(th) * ratio[0, 0], np.sin(th) * ratio[0, 0]
    rod.set_muscle_layers(ratio, strength)
    rng = np.random.default_rng(form * 4 + current_radius * 2 + tm_law)
    # a bent, twisted, stretched, sheared arm: let the oracle evaluate its own caches on it
    x = rod.get("x") + rng.normal(0, 2e-4, (3, N_ELEM + 1))
    x[0] *= 1.05
    Q = rod.get("Q")
    for k in range(N_ELEM):                                    # rotate the frames progressively about a skew axis
        a = 0.02 * k * np.array([0.3, 0.8, 0.5])
        ang = np.linalg.norm(a)
        K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]]) / max(ang, 1e-30)
        R = np.eye(3) + np.sin(ang) * K + (1 - np.cos(ang)) * K @ K
        Q[:, :, k] = Q[:, :, k] @ R.T
    rod.set("x", x); rod.set("Q", Q)
    rod.refresh_strains()
    act = rng.uniform(0, 1, (4, N_ELEM)); act[3] = 0.0
    rod.set("muscle_activation", act)
    f_c, c_c = rod.muscle_probe()
    rest_len = rod.get("rest_lengths")
    layers = [{"kind": int(cfg.muscle_kind[m]), "ratio": ratio[m], "strength": strength[m], "activation": act[m]} for m in range(3)]
    f_np, c_np, forces = onp.muscle_equivalent_loads(
        Q, rod.get("sigma"), rod.get("kappa"), rod.get("tangents"), rod.get("radius"), radii, rest_len,
        0.5 * (rest_len[1:] + rest_len[:-1]), rod.get("dilatation"), rod.get("voronoi_dilatation"), layers,
        [cfg.muscle_fl_coef[k] for k in range(4)], form=form, current_radius=bool(current_radius), tm_law=tm_law)
    scale_f, scale_c = np.abs(f_np).max(), np.abs(c_np).max()
    assert scale_f > 1e-3 and scale_c > 1e-6 and np.abs(c_np[0]).max() > 1e-7 and np.abs(c_np[1]).max() > 1e-7
    np.testing.assert_allclose(f_c, f_np, rtol=0, atol=1e-12 * scale_f)
    np.testing.assert_allclose(c_c, c_np, rtol=0, atol=1e-12 * scale_c)
    np.testing.assert_allclose(rod.get("muscle_force")[:3], forces, rtol=1e-12, atol=1e-18)


def test_arm_push_env_host_logic_on_the_oracle_backend(oracle_built):
    """VecArmPushEnv / ArmPushEnv over the CPU test double: spaces, action validation, the inchworm moves the centre
    of mass forward, device-less auto-reset."""
    import gym_softrobot_amd as gsa
    from tests.oracle_backend import OracleBackend

    env = gsa.ArmPushEnv(backend=OracleBackend(_capi.arm_push_config(1)))
    assert env.action_space.n == 2 and env.observation_space.shape == (84,)
    ob, info = env.reset(seed=0)
    assert ob.dtype == np.float32 and ob.shape == (84,) and info == {} and env.observation_space.contains(ob)
    with pytest.raises(NotImplementedError, match="Action must be 1 or 0"):
        env.step(2)
    total = 0.0
    for a in (0, 0, 1, 1, 0, 0, 1, 1):
        ob, r, te, tr, info = env.step(a)
        assert isinstance(r, float) and isinstance(te, bool) and isinstance(tr, bool) and not te
        total += r
    assert total > 5e-3 and ob[82:].tolist() == [0.0, 1.0]            # the arm has inched forward; one-hot of action 1
    env.close()
    with pytest.raises(NotImplementedError, match="not available"):
        gsa.ArmPushEnv(mode="banana", backend=OracleBackend(_capi.arm_push_config(1)))
    vec = gsa.make_vec("OctoArmPush-v1", 2, final_time=0.06, autoreset=True, numpy_output=True,
                       backend=OracleBackend(_capi.arm_push_config(2, mode="continuous", final_time=0.06)))
    vec.reset(seed=0)
    trunc_seen = 0
    for t in range(6):
        o, r, te, tr, info = vec.step(np.array([[0.1, 0.9], [0.8, 0.2]], np.float32))
        trunc_seen += int(tr.sum())
    assert trunc_seen >= 2 and o.shape == (2, 84)
    vec.close()


# ---- the one-command pin, extended to the muscle arm ----------------------------------------------------------------
def test_pin_tooling_covers_the_muscle_envs(tmp_path, oracle_built):
    """tools/make_pyelastica_golden.py --muscle-envs records all six muscle envs where pyelastica AND coomm import; here
    the same record / replay code runs on oracle-made fixtures: exact self-replay, and a flipped recalled detail is not
    matched — three of the five switches by the envs that only ever drive the transverse layer (the push arm, the arm
    with a weight, CrawlEnv), ALL FIVE by OctoArmTwo and OctoReach, which drive the longitudinal layers too."""
    import sys

    sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tools"))
    import make_pyelastica_golden as gen
    import pyelastica_pin as pin

    with pytest.raises(SystemExit, match="import (elastica|coomm)"):
        gen.main(["--muscle-envs", "--out", str(tmp_path / "never")])
    assert gen.main(["--source", "oracle", "--out", str(tmp_path), "--envs", "--muscle-envs", "--seeds", "42", "--steps", "3"]) == 0
    files = pin.fixture_files(tmp_path, "oracle")
    assert [f.name for f in files] == [f"oracle_{e}_seed42.npz" for e in ("OctoArmPullWeight-v0", "OctoArmPush-v0", "OctoArmPush-v1",
                                                                          "OctoArmTwo-v0", "OctoCrawl-v0", "OctoReach-v0")]
    for f in files:
        fx = dict(np.load(f, allow_pickle=False))
        env_id = str(fx["env_id"])
        assert "sub1_x" not in fx and "step3_x" in fx            # no raw-substep records for the unactuated arm
        assert ("step3_head_x" in fx) == (env_id not in ("OctoArmPush-v0", "OctoArmPush-v1"))     # the rigid body's state too
        drv = pin.OracleDriver(env_id, None)
        assert pin.worst(pin.compare_case(drv, fx)) == 0.0
        drv.close()
        decided = {}
        for k, cands in pin.MUSCLE_SWITCHES.items():
            drv = pin.OracleDriver(env_id, {k: cands[1]})
            decided[k] = pin.worst(pin.compare_case(drv, fx)) > 1e-3
            drv.close()
        longitudinal = env_id in ("OctoArmTwo-v0", "OctoReach-v0")          # they drive the longitudinal layers
        assert decided == {"muscle_equiv_load_form": True, "muscle_position_current_radius": longitudinal,
                           "muscle_tm_length_law": True, "muscle_init_angle_rotates": longitudinal, "muscle_tm_sign": True}, env_id


def test_pull_weight_build_matches_the_executed_reference(oracle_built):
    """ArmPullWeightEnv._build executed from the reference's file (arm_push_env.py:520-618): the Cylinder, the
    BodyBoundaryCondition, the FixedJoint2Rigid, the damper, the sucker — against _capi.arm_pull_weight_config and what
    the oracle allocates from it."""
    r = json.loads((GOLD / "ref_muscle_build_records.json").read_text())["OctoArmPullWeight"]
    cfg = _capi.arm_pull_weight_config(1)
    init = r["init"]
    assert (init["step_skip"], init["time_step"], init["final_time"], init["mode"]) == (cfg.n_substeps, cfg.dt, cfg.final_time, cfg.arm_push_mode)
    assert init["obs_shape"] == [_capi.config_obs_dim(cfg)]
    cyl = r["cylinder"]
    assert (cyl["base_length"], cyl["base_radius"], cyl["density"]) == (cfg.head_length, cfg.head_radius, cfg.head_density)
    assert cyl["direction"] == [0.0, 0.0, 1.0] and cyl["normal"] == [0.0, 1.0, 0.0]
    np.testing.assert_allclose(np.asarray(cyl["start"]) + np.asarray(cyl["direction"]) * cyl["base_length"] / 2,
                               [cfg.head_center[i] for i in range(3)], rtol=0, atol=1e-18)
    assert r["order"] == ["append:FakeRod[0]", "damping:AnalyticalLinearDamper[0]", "append:Cylinder[1]",
                          "constrain:BodyBoundaryCondition[1]", "connect:FixedJoint2Rigid[1,0]",
                          "constrain:ControllableFixConstraint[0]", "forcing:ApplyMuscles[0]"]
    ops = {o["cls"]: o for o in r["ops"]}
    assert ops["AnalyticalLinearDamper"]["kwargs"] == {"damping_constant": cfg.damping_constant, "time_step": cfg.dt}
    j = ops["FixedJoint2Rigid"]["kwargs"]
    assert (j["k"], j["nu"], j["kt"], j["angle"], j["radius"]) == (cfg.joint_k, cfg.joint_nu, cfg.joint_kt, cfg.joint_angle0, cfg.head_radius)
    assert r["connect_indices"] == [-1, 0] and cfg.joint_angle_step == 0.0 and cfg.n_arm == 1
    assert ops["ControllableFixConstraint"]["kwargs"] == {"index": 0, "reduction_ratio": cfg.sucker_reduction_ratio}
    assert r["sucker"] == {"index": 0, "flag": True, "reduction_ratio": 0.9}
    assert r["time_step_kwarg"].startswith("TypeError")          # the subclass passes time_step itself (:518)
    rod = r["straight_rod"]
    assert (rod["density"], rod["youngs_modulus"], rod["shear_modulus"], rod["base_length"]) == (cfg.density, cfg.youngs_modulus, cfg.shear_modulus, cfg.base_length)
    np.testing.assert_array_equal(rod["base_radius"], _capi.arm_push_radii(N_ELEM))
    # the oracle's allocation from that config: Cylinder mass = rho pi r^2 L, reset observation = the reference's
    o = oracle_built.OracleOcto(cfg)
    radii = _capi.arm_push_radii(N_ELEM)
    o.pull_setup(radii, *_capi.es_muscle_layers(radii, 0.012))
    obs = o.reset_pull()
    np.testing.assert_array_equal(obs, np.load(GOLD / "ref_armpush.npz")["w_reset_obs"])
    h = o.head()
    np.testing.assert_allclose(h["mass"], 700.0 * np.pi * 0.015 ** 2 * 0.024, rtol=1e-15)
    np.testing.assert_allclose(h["x"], [-0.0135, 0.0, -0.012], rtol=0, atol=1e-18)


def test_pull_weight_env_drags_the_weight(oracle_built):
    """Host logic over the CPU test double: extend with the base held, then hold the tip and relax — the joint drags
    the rigid weight along; a `time_step` keyword raises like the reference's subclass does."""
    import gym_softrobot_amd as gsa
    from tests.oracle_backend import OracleBackend

    be = OracleBackend(_capi.arm_pull_weight_config(1))
    env = gsa.make_vec("OctoArmPullWeight-v0", 1, backend=be, numpy_output=True)
    assert env.cfg.n_substeps == 1000 and gsa.parity_label("OctoArmPullWeight-v0") is not None
    env.reset(seed=0)
    x0 = be.rods[0].head()["x"][0]
    for a in ([0.0, 0.6], [0.0, 0.6], [0.95, 0.0], [0.95, 0.0]):
        o, r, te, tr, _ = env.step(np.array([a], np.float32))
        assert not te[0] and not tr[0]
    assert be.rods[0].head()["x"][0] > x0 + 5e-3
    with pytest.raises(TypeError, match="time_step"):
        gsa.make_vec("OctoArmPullWeight-v0", 1, time_step=1e-5, backend=be)
    env.close()
