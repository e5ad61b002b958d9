"""SURVEY.md §8(f) N3 on the GPU: the COOMM muscle layers (SOFTROD_FEAT_COOMM_MUSCLES, csrc/softrod_muscle.hpp)
and OctoArmPush-v0 / -v1 (gym_softrobot/envs/octopus/arm_push_env.py:52-347) through the C-ABI against the
oracle's apply_muscles / env_step_push, both math modes, rtol 1e-5 (north_star's tolerance).

PARITY UNPINNED for the muscle law itself: COOMM (uv.lock:173-175) is not on disk, the oracle restates the
published model (Chang et al. 2023) — what these tests hold is HIP == this repo's restatement, plus two known
answers that do not depend on anyone's recollection of COOMM's source (a constant longitudinal force at an
offset bends a clamped rod into the arc of the end couple F r_m; a transverse layer stretches it uniformly)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-5


@pytest.fixture(scope="module")
def torch_gpu():
    import torch

    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    return torch


def _assert_state(be, rods, atol=1e-9):
    st = be.state_numpy()
    for i, r in enumerate(rods):
        for name in ("x", "v", "w", "Q"):
            np.testing.assert_allclose(st[name][i], r.get(name), rtol=RTOL, atol=atol, err_msg=f"{name} env {i}")


@pytest.mark.parametrize("math_mode", [0, 1], ids=["libm", "fast"])
@pytest.mark.parametrize("mode", ["discrete", "continuous"])
def test_arm_push_env_matches_oracle(torch_gpu, hip_lib, oracle_built, mode, math_mode):
    """OctoArmPush-v0 (discrete) / -v1 (continuous) on the tapered 12:1 arm: six env.steps of 500 substeps,
    an inchworm cycle in v0 (hold the base and extend, hold the tip and relax), moving suckers in v1."""
    import gym_softrobot_amd as gsa
    from tests.oracle_backend import OracleBackend

    N = 4
    env = gsa.make_vec("OctoArmPush-v0", N, mode=mode, math_mode=math_mode)
    assert ("ArmPush" in env.backend.kernel_tier()) == (math_mode == 1)
    ref = gsa.make_vec("OctoArmPush-v0", N, mode=mode,
                       backend=OracleBackend(gsa._capi.arm_push_config(N, mode=mode)), numpy_output=True)
    o, _ = env.reset(seed=0)
    o2, _ = ref.reset(seed=0)
    np.testing.assert_array_equal(o.cpu().numpy(), o2)
    rng = np.random.default_rng(3)
    for t in range(6):
        if mode == "discrete":
            # scripts that stay in the regime the restated law is meant for: alternations such as 0,0,1,1,0 drive
            # single elements to stretches of 1e3 and beyond in the ORACLE (the published cubic is unbounded above
            # l ~ 2.2), where no two evaluations can be compared
            a = np.array([[0, 0, 1, 1], [0, 1, 1, 0], [0, 0, 0, 1], [0, 1, 0, 0], [0, 0, 0, 1], [0, 1, 1, 0]][t],
                         np.float32).reshape(N, 1)
        else:
            a = rng.uniform(0.0, 1.0, (N, 2)).astype(np.float32)
            a[0, 0] = [0.0, 1.0, 0.999, 0.5, 0.0125, 0.3][t]            # the clip at both ends of the index range
        o, r, te, tr, info = env.step(a)
        o2, r2, te2, tr2, info2 = ref.step(a)
        torch_gpu.cuda.synchronize()
        np.testing.assert_allclose(o.cpu().numpy(), o2, rtol=RTOL, atol=2e-7, err_msg=f"obs step {t}")
        np.testing.assert_allclose(r.cpu().numpy(), r2, rtol=RTOL, atol=1e-9, err_msg=f"reward step {t}")
        np.testing.assert_array_equal(te.cpu().numpy(), te2)
        np.testing.assert_array_equal(tr.cpu().numpy(), tr2)
        np.testing.assert_array_equal(np.asarray(info["time"]), np.asarray(info2["time"]))
    assert not te.cpu().numpy().any() and max(np.abs(r_.get("v")).max() for r_ in ref.backend.rods) < 10.0
    _assert_state(env.backend, ref.backend.rods)
    st = env.backend.state()
    idx = st["sucker_index"][0].cpu().numpy()
    np.testing.assert_array_equal(idx, [int(r.get("sucker_index")[0]) for r in ref.backend.rods])
    x = env.backend.state_numpy()["x"]
    assert np.abs(x[:, 0, -1] - 0.2).max() > 5e-3          # the arm really extended / moved
    env.close()
    ref.close()


@pytest.mark.parametrize("math_mode", [0, 1], ids=["libm", "fast"])
def test_longitudinal_layers_bend_the_arm_in_3d(torch_gpu, hip_lib, oracle_built, math_mode):
    """All three layers at once with per-ELEMENT activations written through the state view
    (apply_activation with an array: arm_two_env.py:246-248, reach_env.py:176-179), longitudinal muscles turned
    out of the d1 axis so that the couple has components on d1 AND d2, the sucker half way along the arm:
    800 substeps against the oracle."""
    from gym_softrobot_amd import _capi
    from gym_softrobot_amd.backend import HipRodBackend

    N, n = 3, 40
    cfg = _capi.arm_push_config(N, math_mode=math_mode)
    cfg.damper_protocol = 1          # (the per-unit-mass protocol all but freezes the rotations of an arm this thin)
    cfg.sucker_index[0] = 17
    be = HipRodBackend(cfg, 0)
    radii = _capi.arm_push_radii(n)
    ratio, strength = _capi.es_muscle_layers(radii, 0.012)
    th = 0.7                                              # turn the antagonistic pair about d3
    c, s = np.cos(th), np.sin(th)
    for m in (0, 1):
        ratio[m, 0], ratio[m, 1] = c * ratio[m, 0] - s * ratio[m, 1], s * ratio[m, 0] + c * ratio[m, 1]
    be.set_radius_profile(radii)
    be.set_muscle_layers(ratio, strength)
    start, direction, normal = np.zeros(3), np.array([1.0, 0.0, 0.0]), np.array([0.0, 1.0, -0.0])
    be.reset_straight(start, direction, normal)
    rng = np.random.default_rng(11)
    act = rng.uniform(0.0, 1.0, (3, N, n))
    act[1] *= 0.3
    st = be.state()
    st["muscle_activation"][:3, :, :n] = torch_gpu.from_numpy(act).to(st["muscle_activation"].device)
    rods = []
    for i in range(N):
        c1 = cfg.copy()
        c1.n_envs = 1
        r = oracle_built.OracleRod(c1)
        r.set_radius_profile(radii)
        r.set_muscle_layers(ratio, strength)
        r.reset_straight(start, direction, normal)
        a4 = np.zeros((4, n))
        a4[:3] = act[:, i]
        r.set("muscle_activation", a4)
        rods.append(r)
    be.substeps(None, 800)
    for r in rods:
        r.substeps(0.0, 800)
    torch_gpu.cuda.synchronize()
    _assert_state(be, rods)
    x = be.state_numpy()["x"]
    assert np.abs(x[:, 1]).max() > 2e-3 and np.abs(x[:, 2]).max() > 2e-3       # bent out of both planes
    be.close()


@pytest.mark.parametrize("switch", ["muscle_equiv_load_form", "muscle_position_current_radius", "muscle_tm_length_law"])
@pytest.mark.parametrize("math_mode", [0, 1], ids=["libm", "fast"])
def test_hip_follows_every_recalled_muscle_switch(torch_gpu, hip_lib, oracle_built, math_mode, switch):
    """Each recalled COOMM detail is a softrod_config field honoured by the oracle AND both HIP kernels, so that
    whatever the muscle-env fixtures select one day needs no kernel work: flipped, HIP still equals the oracle —
    and differs from the default by far more than the tolerance."""
    from gym_softrobot_amd import _capi
    from gym_softrobot_amd.backend import HipRodBackend

    n = 40
    radii = _capi.arm_push_radii(n)
    ratio, strength = _capi.es_muscle_layers(radii, 0.012)
    start, direction, normal = np.zeros(3), np.array([1.0, 0.0, 0.0]), np.array([0.0, 1.0, -0.0])
    out = {}
    for flipped in (False, True):
        cfg = _capi.arm_push_config(1, math_mode=math_mode)
        cfg.damper_protocol = 1
        if flipped:
            setattr(cfg, switch, 1 - int(getattr(cfg, switch)))
        be = HipRodBackend(cfg, 0)
        be.set_radius_profile(radii)
        be.set_muscle_layers(ratio, strength)
        be.reset_straight(start, direction, normal)
        st = be.state()
        act = np.array([0.8, 0.1, 0.6])
        st["muscle_activation"][:3, :, :n] = torch_gpu.from_numpy(np.tile(act[:, None, None], (1, 1, n))).to(be.device)
        rod = oracle_built.OracleRod(cfg)
        rod.set_radius_profile(radii)
        rod.set_muscle_layers(ratio, strength)
        rod.reset_straight(start, direction, normal)
        for m in range(3):
            rod.apply_activation(m, act[m])
        be.substeps(None, 400)
        rod.substeps(0.0, 400)
        torch_gpu.cuda.synchronize()
        _assert_state(be, [rod])
        out[flipped] = be.state_numpy()["x"][0].copy()
        be.close()
    assert np.abs(out[True] - out[False]).max() > 1e-5 * np.abs(out[False]).max()


@pytest.mark.parametrize("math_mode", [0, 1], ids=["libm", "fast"])
def test_known_answers_on_the_gpu(torch_gpu, hip_lib, math_mode):
    """No oracle involved.  A uniform clamped rod, force-length law fl = 1 (a constant force F = u s):
      * a longitudinal layer at the offset x_m = r rho e_1: every cross-section carries the force -F along the
        muscle and the couple x_m x F, i.e. the equilibrium of the rod under the END couple F r_m and the axial
        end force -F (tests/elastica_chain.py's discrete equations: S sigma / e = -f, B kappa / eps^3 = -c_v):
        a circular arc, kappa = F r_m eps^3 / EI about d2, stretch e = 1 / (1 + F / EA);
      * a transverse layer (on the axis, negative strength): uniform stretch e = 1 / (1 - |F| / EA), straight."""
    from gym_softrobot_amd import _capi
    from gym_softrobot_amd.backend import HipRodBackend

    n, L, r0, E, rho = 12, 0.2, 0.01, 2e6, 700.0
    G = E / 1.5
    A, I = np.pi * r0 ** 2, np.pi * r0 ** 4 / 4
    feats = _capi.FEAT_FIXED_BC | _capi.FEAT_ANALYTICAL_DAMPER | _capi.FEAT_COOMM_MUSCLES
    for kind in ("longitudinal", "transverse"):
        cfg = _capi.softpendulum_config(2, n_elems=n, math_mode=math_mode)
        cfg.env_kind, cfg.features = _capi.ENV_NONE, feats
        cfg.base_length, cfg.base_radius, cfg.density, cfg.youngs_modulus, cfg.shear_modulus = L, r0, rho, E, G
        # uniform damper protocol at the bending mode's critical damping (the per-unit-mass protocol multiplies
        # omega by exp(-nu dt m / J) ~ 1e-20 per substep on a rod this thin: no rotation would ever settle)
        cfg.dt, cfg.damping_constant, cfg.damper_protocol = 5e-5, 45.0, 1
        _capi.muscle_defaults(cfg)
        cfg.n_muscles = 1
        cfg.muscle_kind[0] = _capi.MUSCLE_LONGITUDINAL if kind == "longitudinal" else _capi.MUSCLE_TRANSVERSE
        cfg.muscle_fl_degree = 0
        cfg.muscle_fl_coef[0] = 1.0
        be = HipRodBackend(cfg, 0)
        ratio = np.zeros((1, 3, n))
        if kind == "longitudinal":
            ratio[0, 0] = 0.6
        strength = np.full((1, n), 3.0 if kind == "longitudinal" else -25.0)
        be.set_muscle_layers(ratio, strength)
        be.reset_straight(np.zeros(3), np.array([1.0, 0.0, 0.0]), np.array([0.0, 1.0, 0.0]))
        u = np.array([1.0, 0.5])                                   # two envs, two activation levels
        st = be.state()
        st["muscle_activation"][0, :, :n] = torch_gpu.from_numpy(np.tile(u[:, None], (1, n))).to(be.device)
        be.substeps(None, 30000)                                   # 1.5 s: settled to 1e-12
        torch_gpu.cuda.synchronize()
        s = be.state_numpy()
        assert np.abs(s["v"]).max() < 1e-9
        for e_, act in enumerate(u):
            F = act * strength[0, 0]
            x = s["x"][e_]
            lens = np.sqrt(((x[:, 1:] - x[:, :-1]) ** 2).sum(axis=0))
            if kind == "transverse":
                stretch = 1.0 / (1.0 + F / (E * A))                # F < 0: longer
                np.testing.assert_allclose(lens / (L / n), stretch, rtol=1e-8)
                np.testing.assert_allclose(x[0], np.linspace(0, L * stretch, n + 1), rtol=1e-8, atol=1e-12)
                assert np.abs(x[1:]).max() < 1e-12
            else:
                stretch = 1.0 / (1.0 + F / (E * A))
                rm = 0.6 * r0 / np.sqrt(stretch)                   # x_m = radius ratio, radius = r0 / sqrt(e)
                kappa = F * rm * stretch ** 3 / (E * I)            # B kappa / eps^3 = F r_m
                np.testing.assert_allclose(lens[1:] / (L / n), stretch, rtol=1e-7)
                # joint angle kappa * D^ between neighbouring elements, about d2 = d3 x d1 = x cross y = z
                Q = s["Q"][e_]
                ang = np.arctan2(Q[2, 1], Q[2, 0])                 # heading of d3 in the x-y plane
                np.testing.assert_allclose(np.diff(ang), np.full(n - 1, kappa * (L / n)), rtol=1e-7)
                assert np.abs(x[2]).max() < 1e-12
        be.close()


def test_arm_push_nan_and_time_limit(torch_gpu, hip_lib):
    """NaN anywhere in the state: terminated, reward -20, np.nan_to_num on the observation
    (arm_push_env.py:298-318,335-338); `time > final_time` truncates without terminating (:321-325)."""
    import gym_softrobot_amd as gsa

    env = gsa.make_vec("OctoArmPush-v0", 3, final_time=0.06)
    env.reset(seed=0)
    st = env.backend.state()
    st["omega"][1, 1, 7] = float("nan")                    # NaN in omega only: still invalid (:305)
    a = np.zeros((3, 1), np.float32)
    o, r, te, tr, _ = env.step(a)
    torch_gpu.cuda.synchronize()
    assert te.cpu().numpy().tolist() == [False, True, False]
    assert r.cpu().numpy()[1] == -20.0 and np.isfinite(o.cpu().numpy()).all()
    for k in range(2):
        o, r, te, tr, info = env.step(a)
    torch_gpu.cuda.synchronize()
    assert tr.cpu().numpy().tolist() == [True, True, True] and te.cpu().numpy().tolist() == [False, True, False]
    env.close()


def test_arm_push_device_autoreset_equals_host_autoreset(torch_gpu, hip_lib):
    """NEXT_STEP auto-reset of the muscle arm on the device (fresh muscle objects, SuckerController(index=0) switched
    on again) is bit-identical to the host-driven one."""
    import gym_softrobot_amd as gsa

    N = 3
    host = gsa.make_vec("OctoArmPush-v1", N, final_time=0.06, autoreset=True)
    dev = gsa.make_vec("OctoArmPush-v1", N, final_time=0.06, autoreset="device")
    host.reset(seed=0)
    dev.reset(seed=0)
    acts = np.random.default_rng(2).uniform(0, 1, (12, N, 2)).astype(np.float32)
    for t in range(12):
        o1, r1, te1, tr1, _ = host.step(acts[t])
        o2, r2, te2, tr2, _ = dev.step(acts[t])
        torch_gpu.cuda.synchronize()
        np.testing.assert_array_equal(o1.cpu().numpy(), o2.cpu().numpy(), err_msg=f"step {t}")
        np.testing.assert_array_equal(r1.cpu().numpy(), r2.cpu().numpy())
        np.testing.assert_array_equal(tr1.cpu().numpy(), tr2.cpu().numpy())
    host.close()
    dev.close()


@pytest.mark.parametrize("math_mode", [0, 1], ids=["libm", "fast"])
def test_hip_replays_muscle_env_fixtures_of_the_pin_tooling(tmp_path, hip_lib, oracle_built, math_mode):
    """The record / replay harness of the one-command pin (tools/pyelastica_pin.py, MUSCLE_ENVS): fixtures the oracle
    makes for the six muscle envs are replayed by the HIP library within 1e-5 up to the strict horizon — what will
    run against COOMM-made fixtures the day they exist."""
    import sys
    from pathlib import Path

    sys.path.insert(0, str(Path(__file__).resolve().parents[1] / "tools"))
    import make_pyelastica_golden as gen
    import pyelastica_pin as pin

    assert gen.main(["--source", "oracle", "--out", str(tmp_path), "--envs", "--muscle-envs", "--seeds", "1", "--steps", "3"]) == 0
    for f in pin.fixture_files(tmp_path, "oracle"):
        fx = dict(np.load(f, allow_pickle=False))
        if math_mode == 0 and str(fx["env_id"]) not in ("OctoArmPush-v0", "OctoArmPush-v1"):
            continue                     # the rigid-body muscle envs exist for SOFTROD_MATH_FAST only (softrod_create says so)
        drv = pin.HipDriver(str(fx["env_id"]), None, math_mode=math_mode)
        dev = pin.compare_case(drv, fx)
        drv.close()
        assert pin.strict_worst(dev, str(fx["env_id"])) <= 1e-5, (f.name, max(dev, key=dev.get), max(dev.values()))


def test_arm_pull_weight_env_matches_oracle(torch_gpu, hip_lib, oracle_built):
    """OctoArmPullWeight-v0 (ArmPullWeightEnv, arm_push_env.py:516-618): the muscle arm joined to a rigid Cylinder by
    FixedJoint2Rigid, one wave per env on the rigid-body kernel (softrod_octo.hpp) with the tapered material table, the
    COOMM layers and the sucker: four env.steps of 1000 substeps against the oracle's two-body stepper — observations,
    rewards, flags, the arm's state and the weight's."""
    import gym_softrobot_amd as gsa
    from tests.oracle_backend import OracleBackend

    N = 3
    env = gsa.make_vec("OctoArmPullWeight-v0", N)
    assert "ArmPullWeight" in env.backend.kernel_tier()
    ref = gsa.make_vec("OctoArmPullWeight-v0", N, backend=OracleBackend(gsa._capi.arm_pull_weight_config(N)), numpy_output=True)
    o, _ = env.reset(seed=0)
    o2, _ = ref.reset(seed=0)
    np.testing.assert_array_equal(o.cpu().numpy(), o2)
    acts = np.array([[[0.0, 0.6], [0.3, 0.2], [0.0, 0.0]], [[0.0, 0.6], [0.3, 0.9], [0.5, 0.4]],
                     [[0.95, 0.0], [1.0, 0.1], [0.5, 0.0]], [[0.95, 0.0], [0.0, 0.5], [0.2, 0.3]]], np.float32)
    for t in range(4):
        o, r, te, tr, info = env.step(acts[t])
        o2, r2, te2, tr2, info2 = ref.step(acts[t])
        torch_gpu.cuda.synchronize()
        np.testing.assert_allclose(o.cpu().numpy(), o2, rtol=RTOL, atol=2e-7, err_msg=f"obs step {t}")
        np.testing.assert_allclose(r.cpu().numpy(), r2, rtol=RTOL, atol=1e-9, err_msg=f"reward step {t}")
        np.testing.assert_array_equal(te.cpu().numpy(), te2)
        np.testing.assert_array_equal(tr.cpu().numpy(), tr2)
    st = env.backend.state_numpy()
    head = env.backend.state()["head"].cpu().numpy()            # [20][N]: x 0..2, v 3..5, Q 6..14, w 15..17
    for i, q in enumerate(ref.backend.rods):
        arm = q.arm(0)
        for name in ("x", "v", "w", "Q"):
            np.testing.assert_allclose(st[name][i], arm.get(name), rtol=RTOL, atol=1e-9, err_msg=f"{name} env {i}")
        h = q.head()
        np.testing.assert_allclose(head[0:3, i], h["x"], rtol=RTOL, atol=1e-10)
        np.testing.assert_allclose(head[3:6, i], h["v"], rtol=RTOL, atol=1e-8)
        np.testing.assert_allclose(head[6:15, i].reshape(3, 3), h["Q"], rtol=RTOL, atol=1e-10)
        np.testing.assert_allclose(head[15:18, i], h["w"], rtol=RTOL, atol=1e-7)
    assert head[0, 0] > -0.0135 + 3e-3          # env 0 dragged its weight forward
    env.close()
    ref.close()
