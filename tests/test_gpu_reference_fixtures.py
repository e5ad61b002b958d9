"""The HIP path against the fixtures recorded by EXECUTING the reference's own env code
(tools/make_env_golden.py; CPU twin: tests/test_reference_fixtures.py).  Every case is one env
of a batch: the state the reference's step() saw after its substep loop is written into the
resident rows through softrod_state_view, the handle runs the env prologue + epilogue with
n_substeps = 0 (or, for OctoFlat, the recorded pre-loop state and the real 40 substeps), and the
observation / reward / flags / side effects must be the reference's.  Through the C-ABI."""
import json
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GOLD = Path(__file__).parent / "golden"
OBS_TOL = dict(rtol=2e-7, atol=1e-9)        # float32 observations: one ulp
RTOL = 1e-5                                 # north_star's tolerance where physics is in between


@pytest.fixture(scope="module")
def torch_gpu():
    import torch

    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    return torch


def _backend(cfg):
    from gym_softrobot_amd.backend import HipRodBackend

    return HipRodBackend(cfg, 0)


def _put(st, name, per_env):
    """per_env: (N, comps, k) -> resident rows [comps][N][stride]."""
    import torch

    a = np.ascontiguousarray(np.moveaxis(np.asarray(per_env, np.float64), 0, 1))
    st[name][:, :, : a.shape[2]] = torch.from_numpy(a).to(st[name].device)


def _inject_rods(be, z, pre="step_", extra=()):
    import torch

    st = be.state()
    n = len(z[pre + "time"])
    _put(st, "position", z[pre + "x"])
    _put(st, "velocity", z[pre + "v"])
    _put(st, "omega", z[pre + "w"])
    _put(st, "tangents", z[pre + "tangents"])
    _put(st, "director", z[pre + "Q"].reshape(n, 9, -1))
    st["time"][:] = torch.from_numpy(np.ascontiguousarray(z[pre + "time"])).to(st["time"].device)
    for name, arr in extra:
        _put(st, name, arr)
    return st


def _flags(t):
    return t.cpu().numpy().astype(bool)


# ---------------------------------------------------------------------------------------------
# SoftPendulum-v0
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("math_mode", [0, 1], ids=["libm", "fast"])
def test_softpendulum_reset_against_the_reference(torch_gpu, hip_lib, math_mode):
    import gym_softrobot_amd as gsa

    z = np.load(GOLD / "ref_softpendulum.npz")
    seeds = [int(s) for s in z["reset_seed"]]
    env = gsa.make_vec("SoftPendulum-v0", len(seeds), math_mode=math_mode)
    obs, _ = env.reset(seed=seeds)
    np.testing.assert_allclose(obs.cpu().numpy(), z["reset_obs"], **OBS_TOL)
    st = env.backend.state()
    bc = st["bc_targets"].cpu().numpy()                          # [12][N]: fixed_position, fixed_directors
    np.testing.assert_array_equal(bc[0:3].T, z["reset_bc_fixed_position"])
    np.testing.assert_array_equal(bc[3:12].T.reshape(-1, 3, 3), z["reset_bc_fixed_directors"])
    q = st["director"][:, :, 0].cpu().numpy().T.reshape(-1, 3, 3)
    # d3 = the normalised first edge (straight_rod's allocation; one ulp from `direction`), d1 = normal
    np.testing.assert_allclose(q[:, 2], z["reset_direction"], rtol=0, atol=3e-16)
    np.testing.assert_array_equal(q[:, 0], z["reset_normal"])
    env.close()


@pytest.mark.parametrize("math_mode", [0, 1], ids=["libm", "fast"])
def test_softpendulum_step_epilogue_against_the_reference(torch_gpu, hip_lib, math_mode):
    """soft_pendulum.py:163-166,196-251 on the states the reference's own step() saw."""
    from gym_softrobot_amd import _capi
    from gym_softrobot_amd.seeding import initial_angle, np_random

    z = np.load(GOLD / "ref_softpendulum.npz")
    n = len(z["step_label"])
    cfg = _capi.softpendulum_config(n, math_mode=math_mode)
    cfg.n_substeps = 0
    be = _backend(cfg)
    th = initial_angle(np_random(5)[0])                 # the fixture's episode (its boundary-condition targets)
    be.reset(np.full(n, th))
    st = _inject_rods(be, z)
    st["prev_action"][:, 0] = torch_gpu.from_numpy(z["step_prev_action_before"][:, 0].copy()).to(st["prev_action"].device)
    obs, rew, term, trunc = be.step(z["step_action"].reshape(n, 1))
    torch_gpu.cuda.synchronize()
    np.testing.assert_allclose(obs.cpu().numpy(), z["step_obs"], **OBS_TOL)
    np.testing.assert_allclose(rew.cpu().numpy(), z["step_reward"], rtol=1e-12, atol=1e-13)
    np.testing.assert_array_equal(_flags(term), z["step_terminated"])
    np.testing.assert_array_equal(_flags(trunc), z["step_truncated"])
    np.testing.assert_array_equal(st["time"].cpu().numpy(), z["step_info_time"])
    np.testing.assert_array_equal(st["prev_action"][:, 0].cpu().numpy(), z["step_prev_action_after"][:, 0])
    assert bool(z["step_terminated"].any()) and bool(z["step_truncated"].any()) and np.isnan(z["step_obs"]).any()
    be.close()


def test_pendulum_boundary_condition_against_the_reference_class(torch_gpu, hip_lib):
    """PendulumBoundaryConditions.constrain_values / constrain_rates (build.py:71-79) evaluated by
    the reference's class on arbitrary states; the fast kernel re-establishes exactly that at its
    entry, so a launch of zero substeps is one application of both."""
    from gym_softrobot_amd import _capi

    z = np.load(GOLD / "ref_softpendulum.npz")
    n = len(z["op_force"])
    cfg = _capi.softpendulum_config(n, math_mode=1)
    be = _backend(cfg)
    be.reset(np.full(n, 1.5))
    st = be.state()
    _put(st, "position", z["op_x_in"])
    _put(st, "velocity", z["op_v_in"])
    _put(st, "omega", z["op_w_in"])
    _put(st, "director", z["op_Q_in"].reshape(n, 9, -1))
    bc = np.concatenate([z["op_fixed_position"], z["op_fixed_directors"].reshape(n, 9)], axis=1).T    # [12][N]
    st["bc_targets"][:] = torch_gpu.from_numpy(np.ascontiguousarray(bc)).to(st["bc_targets"].device)
    be.substeps(None, 0)
    torch_gpu.cuda.synchronize()
    got = be.state_numpy()
    np.testing.assert_array_equal(got["x"], z["op_x_out"])
    np.testing.assert_array_equal(got["v"], z["op_v_out"])
    np.testing.assert_array_equal(got["w"], z["op_w_out"])
    np.testing.assert_array_equal(got["Q"], z["op_Q_out"])
    be.close()


# ---------------------------------------------------------------------------------------------
# SoftPendulum3D-v0
# ---------------------------------------------------------------------------------------------
def test_softpendulum3d_reset_against_the_reference(torch_gpu, hip_lib):
    import gym_softrobot_amd as gsa

    z = np.load(GOLD / "ref_softpendulum3d.npz")
    seeds = [int(s) for s in z["reset_seed"]]
    env = gsa.make_vec("SoftPendulum3D-v0", len(seeds))
    obs, _ = env.reset(seed=seeds)
    np.testing.assert_allclose(obs.cpu().numpy(), z["reset_obs"], **OBS_TOL)
    q = env.backend.state()["director"][:, :, 0].cpu().numpy().T.reshape(-1, 3, 3)
    np.testing.assert_allclose(q[:, 2], z["reset_direction"], rtol=0, atol=3e-16)
    np.testing.assert_array_equal(q[:, 0], z["reset_normal"])
    env.close()


@pytest.mark.parametrize("math_mode", [0, 1], ids=["libm", "fast"])
def test_softpendulum3d_step_epilogue_against_the_reference(torch_gpu, hip_lib, math_mode):
    """soft_pendulum_3d.py:99-174.  The fast kernel re-imposes the base position on node 0 at its
    entry, so it replays the cases whose state is consistent with the controller (the rollout);
    the libm kernel passes the state through untouched and replays all of them."""
    from gym_softrobot_amd import _capi

    z = np.load(GOLD / "ref_softpendulum3d.npz")
    keep = np.arange(len(z["step_label"]))
    if math_mode == 1:
        keep = np.array([i for i, l in enumerate(z["step_label"]) if str(l).startswith("rollout")])
    sub = {k: z[k][keep] for k in z.files if k.startswith("step_")}
    n = len(keep)
    cfg = _capi.softpendulum3d_config(n, math_mode=math_mode)
    cfg.n_substeps = 0
    be = _backend(cfg)
    be.reset_straight(np.zeros(3), np.array([0.0, 0.0, 1.0]), np.array([0.0, 1.0, 0.0]))
    st = _inject_rods(be, sub)
    dev = st["control"].device
    st["control"][:] = torch_gpu.from_numpy(np.ascontiguousarray(sub["step_ctrl_before"].T)).to(dev)
    st["prev_action"][:, 0:2] = torch_gpu.from_numpy(sub["step_prev_action_before"].copy()).to(dev)
    obs, rew, term, trunc = be.step(sub["step_action"])
    torch_gpu.cuda.synchronize()
    np.testing.assert_allclose(obs.cpu().numpy(), sub["step_obs"], **OBS_TOL)
    np.testing.assert_allclose(rew.cpu().numpy(), sub["step_reward"], rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(be.aux[:, 0].cpu().numpy(), sub["step_info_tilt"], rtol=1e-12, atol=1e-13)
    np.testing.assert_array_equal(_flags(term), sub["step_terminated"])
    np.testing.assert_array_equal(_flags(trunc), sub["step_truncated"])
    # commanded base position after set_action (clipped); its velocity divides by step_skip * dt,
    # which is zero in this zero-substep handle — checked with real steps below
    np.testing.assert_array_equal(st["control"][0:2].cpu().numpy().T, sub["step_ctrl_after"][:, 0:2])
    be.close()


def test_softpendulum3d_set_action_against_the_reference(torch_gpu, hip_lib):
    """set_action's controller update (soft_pendulum_3d.py:99-113) does not depend on the rod:
    real 400-substep steps from rest, controller preset as in the fixture."""
    from gym_softrobot_amd import _capi

    z = np.load(GOLD / "ref_softpendulum3d.npz")
    n = len(z["step_label"])
    be = _backend(_capi.softpendulum3d_config(n))
    be.reset_straight(np.zeros(3), np.array([0.0, 0.0, 1.0]), np.array([0.0, 1.0, 0.0]))
    st = be.state()
    st["control"][:] = torch_gpu.from_numpy(np.ascontiguousarray(z["step_ctrl_before"].T)).to(st["control"].device)
    be.step(z["step_action"])
    torch_gpu.cuda.synchronize()
    np.testing.assert_allclose(st["control"].cpu().numpy().T, z["step_ctrl_after"], rtol=1e-15, atol=0)
    lab = list(z["step_label"])
    assert z["step_ctrl_after"][lab.index("clip_hi")][0] == 0.5 and z["step_ctrl_after"][lab.index("clip_lo")][0] == -0.5
    be.close()


def test_softpendulum3d_rejects_actions_outside_the_box(torch_gpu, hip_lib):
    """soft_pendulum_3d.py:116-117; the fixture records that the reference raises ValueError."""
    import gym_softrobot_amd as gsa

    rec = json.loads((GOLD / "ref_build_records.json").read_text())["SoftPendulum3D-v0"]
    assert rec["bad_action_raises_ValueError"] == [True, True]
    env = gsa.make("SoftPendulum3D-v0")
    env.reset(seed=0)
    for bad in (np.array([1.5, 0.0], np.float32), np.array([0.0, 0.0, 0.0], np.float32)):
        with pytest.raises(ValueError):
            env.step(bad)
    env.close()


# ---------------------------------------------------------------------------------------------
# OctoArmSingle-v0
# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("math_mode", [0, 1], ids=["libm", "fast"])
def test_armsingle_step_epilogue_against_the_reference(torch_gpu, hip_lib, math_mode):
    """arm_single_env.py:186-235,252-316 on the states the reference's own step() saw."""
    import gym_softrobot_amd as gsa
    from gym_softrobot_amd import _capi

    z = np.load(GOLD / "ref_armsingle.npz")
    env = gsa.make_vec("OctoArmSingle-v0", 1, math_mode=math_mode)
    obs0, _ = env.reset(seed=0)
    np.testing.assert_allclose(obs0.cpu().numpy()[0], z["reset_obs"], **OBS_TOL)
    env.close()
    n = len(z["step_label"])
    cfg = _capi.arm_single_config(n, math_mode=math_mode)
    cfg.n_substeps = 0
    be = _backend(cfg)
    be.reset_straight(np.zeros(3), np.array([1.0, 0.0, 0.0]), np.array([0.0, 0.0, 1.0]))
    st = _inject_rods(be, z, extra=(("kappa", z["step_kappa"]),))
    dev = st["control"].device
    st["env_memory"][:, :49] = torch_gpu.from_numpy(z["step_prev_kappa_before"].copy()).to(dev)
    st["control"][0:2] = torch_gpu.from_numpy(np.ascontiguousarray(z["step_prev_com_before"].T)).to(dev)
    st["prev_action"][:, 0:7] = torch_gpu.from_numpy(z["step_prev_action_before"].copy()).to(dev)
    obs, rew, term, trunc = be.step(z["step_action"])
    torch_gpu.cuda.synchronize()
    np.testing.assert_allclose(obs.cpu().numpy(), z["step_obs"], rtol=2e-6, atol=2e-7)
    np.testing.assert_allclose(rew.cpu().numpy(), z["step_reward"], rtol=1e-9, atol=1e-12)
    np.testing.assert_array_equal(_flags(term), z["step_terminated"])
    np.testing.assert_array_equal(_flags(trunc), z["step_truncated"])
    # set_action: rest_kappa[0, :] = interp1d(cubic)(action), applied as the constant basis matrix
    rk = st["rest_kappa"][:, :, :49].cpu().numpy()
    np.testing.assert_allclose(np.moveaxis(rk, 0, 1), z["step_rest_kappa"], rtol=1e-12, atol=1e-12)
    np.testing.assert_array_equal(st["env_memory"][:, :49].cpu().numpy(), z["step_prev_kappa_after"])
    ok = ~np.isnan(z["step_obs"]).any(axis=1)
    np.testing.assert_allclose(st["control"][0:2].cpu().numpy().T[ok], z["step_prev_com_after"][ok], rtol=1e-12, atol=1e-13)
    be.close()


# ---------------------------------------------------------------------------------------------
# OctoFlat-v0
# ---------------------------------------------------------------------------------------------
FLAT_FPS = 357


def test_octoflat_reset_against_the_reference(torch_gpu, hip_lib):
    import gym_softrobot_amd as gsa

    z = np.load(GOLD / "ref_octoflat.npz")
    seeds = [int(s) for s in z["reset_seed"]]
    env = gsa.make_vec("OctoFlat-v0", len(seeds), numpy_output=True)
    obs, _ = env.reset(seed=seeds)
    d = env.split_obs(obs)
    np.testing.assert_array_equal(env.targets, z["reset_target"])
    np.testing.assert_allclose(d["individual"], z["reset_individual"], **OBS_TOL)
    np.testing.assert_allclose(d["shared"], z["reset_shared"], **OBS_TOL)
    env.close()


def test_octoflat_step_from_the_recorded_pre_state(torch_gpu, hip_lib):
    """FlatEnv.step (flat_env.py:288-408): from the recorded pre-loop state the HIP path runs
    set_action (zero-padded cubic spline per arm), the 40 substeps and the epilogue, and must
    return what the reference's step() returned for the oracle's post-loop state (the 40
    substeps in between are this repo's physics on both sides: rtol 1e-5)."""
    import gym_softrobot_amd as gsa

    z = np.load(GOLD / "ref_octoflat.npz")
    keep = np.array([i for i, l in enumerate(z["step_label"]) if str(l).startswith("rollout")])
    n = len(keep)
    env = gsa.make_vec("OctoFlat-v0", n, numpy_output=True, recording_fps=FLAT_FPS)
    assert int(env.cfg.n_substeps) == 40
    env.reset(seed=0)
    be = env.backend
    st = be.state()
    seg = st["arm_stride"]
    dev = st["position"].device
    T = torch_gpu.from_numpy
    for j, i in enumerate(keep):
        for a in range(8):
            lo = a * seg
            st["position"][:, j, lo : lo + 11] = T(z["step_pre_x"][i][a].copy()).to(dev)
            st["velocity"][:, j, lo : lo + 11] = T(z["step_pre_v"][i][a].copy()).to(dev)
            st["omega"][:, j, lo : lo + 10] = T(z["step_pre_w"][i][a].copy()).to(dev)
            st["director"][:, j, lo : lo + 10] = T(np.ascontiguousarray(z["step_pre_Q"][i][a].reshape(9, 10))).to(dev)
            st["kappa"][:, j, lo : lo + 9] = T(z["step_pre_kappa"][i][a].copy()).to(dev)
            st["rest_kappa"][:, j, lo : lo + 9] = T(z["step_pre_rest_kappa"][i][a].copy()).to(dev)
        head = np.concatenate([z["step_pre_head_x"][i], z["step_pre_head_v"][i], z["step_pre_head_Q"][i].ravel(),
                               z["step_pre_head_w"][i], z["step_target"][i]])
        st["head"][0:20, j] = T(head).to(dev)
        st["time"][j] = float(z["step_time"][i]) - 40 * float(env.cfg.dt)
        st["prev_action"][j, :24] = T(z["step_prev_action_before"][i].copy()).to(dev)
    obs, rew, term, trunc, info = env.step(z["step_action"][keep])
    d = env.split_obs(obs)
    np.testing.assert_allclose(d["individual"], z["step_individual"][keep], rtol=RTOL, atol=2e-6)
    np.testing.assert_allclose(d["shared"], z["step_shared"][keep], rtol=RTOL, atol=2e-6)
    np.testing.assert_allclose(rew, z["step_reward"][keep], rtol=RTOL, atol=1e-6)
    np.testing.assert_array_equal(term, z["step_terminated"][keep])
    np.testing.assert_array_equal(trunc, z["step_truncated"][keep])
    rk = be.octo_state_numpy()["rest_kappa"]
    np.testing.assert_allclose(rk, z["step_rest_kappa"][keep], rtol=1e-12, atol=1e-12)
    env.close()


def test_octoflatlite_step_from_the_recorded_pre_state(torch_gpu, hip_lib):
    """OctoFlatLite-v0 (n_arm = 1, n_action = 8): reset observation, then the reference's step()
    outputs from the recorded pre-loop states through the real 40 substeps."""
    import gym_softrobot_amd as gsa

    z = np.load(GOLD / "ref_octoflatlite.npz")
    seeds = [int(s) for s in z["reset_seed"]]
    env = gsa.make_vec("OctoFlat-v0", len(seeds), numpy_output=True, n_arm=1, n_action=8)
    obs, _ = env.reset(seed=seeds)
    d = env.split_obs(obs)
    np.testing.assert_allclose(d["individual"], z["reset_individual"], **OBS_TOL)
    np.testing.assert_allclose(d["shared"], z["reset_shared"], **OBS_TOL)
    env.close()
    n = len(z["step_label"])
    env = gsa.make_vec("OctoFlat-v0", n, numpy_output=True, recording_fps=FLAT_FPS, n_arm=1, n_action=8)
    env.reset(seed=0)
    st = env.backend.state()
    dev = st["position"].device
    T = torch_gpu.from_numpy
    for i in range(n):
        st["position"][:, i, :11] = T(z["step_pre_x"][i][0].copy()).to(dev)
        st["velocity"][:, i, :11] = T(z["step_pre_v"][i][0].copy()).to(dev)
        st["omega"][:, i, :10] = T(z["step_pre_w"][i][0].copy()).to(dev)
        st["director"][:, i, :10] = T(np.ascontiguousarray(z["step_pre_Q"][i][0].reshape(9, 10))).to(dev)
        st["kappa"][:, i, :9] = T(z["step_pre_kappa"][i][0].copy()).to(dev)
        st["rest_kappa"][:, i, :9] = T(z["step_pre_rest_kappa"][i][0].copy()).to(dev)
        head = np.concatenate([z["step_pre_head_x"][i], z["step_pre_head_v"][i], z["step_pre_head_Q"][i].ravel(),
                               z["step_pre_head_w"][i], z["step_target"][i]])
        st["head"][0:20, i] = T(head).to(dev)
        st["time"][i] = float(z["step_time"][i]) - 40 * float(env.cfg.dt)
        st["prev_action"][i, :8] = T(z["step_prev_action_before"][i].copy()).to(dev)
    obs, rew, term, trunc, _ = env.step(z["step_action"])
    d = env.split_obs(obs)
    np.testing.assert_allclose(d["individual"], z["step_individual"], rtol=RTOL, atol=2e-6)
    np.testing.assert_allclose(d["shared"], z["step_shared"], rtol=RTOL, atol=2e-6)
    np.testing.assert_allclose(rew, z["step_reward"], rtol=RTOL, atol=1e-6)
    np.testing.assert_array_equal(term, z["step_terminated"])
    np.testing.assert_array_equal(trunc, z["step_truncated"])
    env.close()
