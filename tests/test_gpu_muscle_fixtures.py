"""The reference's own ArmPushEnv code, executed (tools/make_muscle_env_golden.py -> tests/golden/ref_armpush.npz),
replayed through the HIP library: set_action (the sucker's index with Python indexing, the layers' activations),
prev_cm_pos, the NaN check, reward, truncation and get_state with np.nan_to_num of
gym_softrobot/envs/octopus/arm_push_env.py:225-347, by state-view injection into a handle built with n_substeps = 0
(softrod_step is then the prologue and the epilogue on the resident state; prev_cm_pos is what the control rows hold).
tests/test_muscles.py replays the same fixtures through the oracle on the CPU.  Nothing of COOMM's muscle law is
involved (it is not on disk): these fixtures pin the env code around it."""
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
GOLD = Path(__file__).resolve().parent / "golden"
N_ELEM = 40


@pytest.mark.parametrize("math_mode", [0, 1], ids=["libm", "fast"])
@pytest.mark.parametrize("mode", ["discrete", "continuous"])
def test_arm_push_step_replays_the_executed_reference(hip_lib, mode, math_mode):
    import torch

    from gym_softrobot_amd import _capi
    from gym_softrobot_amd.backend import HipRodBackend

    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    z = np.load(GOLD / "ref_armpush.npz")
    p = "d_" if mode == "discrete" else "c_"
    labels = [str(s) for s in z[p + "step_label"]]
    keep = [i for i, s in enumerate(labels) if s != "nan_alpha"]      # alpha_collection is not part of the state (test_muscles.py)
    N = len(keep)
    cfg = _capi.arm_push_config(N, mode=mode, math_mode=math_mode)
    cfg.n_substeps = 0
    be = HipRodBackend(cfg, 0)
    radii = _capi.arm_push_radii(N_ELEM)
    be.set_radius_profile(radii)
    be.set_muscle_layers(*_capi.es_muscle_layers(radii, 0.012))
    start, direction, normal = np.zeros(3), np.array([1.0, 0.0, 0.0]), np.array([0.0, 1.0, -0.0])
    be.reset_straight(start, direction, normal)
    # the reset observation is the reference's get_state on the freshly built arm
    torch.cuda.synchronize()
    np.testing.assert_array_equal(be.observe(None).cpu().numpy()[0], z[p + "reset_obs"])
    st = be.state()
    dev = be.device

    def put(name, arr, width):                   # (N, comps, width) -> rows [comps][N][64]
        t = torch.from_numpy(np.ascontiguousarray(np.moveaxis(arr, 0, 1))).to(dev)
        st[name][:, :, :width] = t

    put("position", z[p + "step_x"][keep], N_ELEM + 1)
    put("velocity", z[p + "step_v"][keep], N_ELEM + 1)
    put("omega", z[p + "step_w"][keep], N_ELEM)
    put("director", z[p + "step_Q"][keep].reshape(N, 9, N_ELEM), N_ELEM)
    st["time"][:] = torch.from_numpy(z[p + "step_time"][keep]).to(dev)
    # prev_cm_pos: the centre of mass of the PRE-step state, with the masses of the allocation
    vol = np.pi * radii ** 2 * (0.2 / N_ELEM)
    mass = np.zeros(N_ELEM + 1)
    mass[:-1] += 0.5 * 700.0 * vol
    mass[1:] += 0.5 * 700.0 * vol
    pre = z[p + "step_pre_x"][keep]
    com = (pre[:, :2] * mass).sum(axis=2) / mass.sum()
    st["control"][:2] = torch.from_numpy(com.T.copy()).to(dev)
    # layers 0, 1 hold 0.25 beforehand: the continuous mode must leave them alone, the discrete one overwrite them
    st["muscle_activation"][:2, :, :N_ELEM] = 0.25
    act = z[p + "step_action"][keep].astype(np.float32)
    a = act[:, :1] if mode == "discrete" else act
    obs, rew, term, trunc = be.step(a)
    torch.cuda.synchronize()
    obs, rew = obs.cpu().numpy(), rew.cpu().numpy()
    for k, i in enumerate(keep):
        np.testing.assert_array_equal(obs[k], z[p + "step_obs"][i], err_msg=labels[i])
        np.testing.assert_allclose(rew[k], z[p + "step_reward"][i], rtol=1e-12, atol=1e-15, err_msg=labels[i])
    np.testing.assert_array_equal(term.cpu().numpy().astype(bool), z[p + "step_terminated"][keep])
    np.testing.assert_array_equal(trunc.cpu().numpy().astype(bool), z[p + "step_truncated"][keep])
    np.testing.assert_array_equal(st["sucker_index"][0].cpu().numpy(), z[p + "step_sucker_index"][keep])
    got = st["muscle_activation"][:3, :, :N_ELEM].cpu().numpy()
    want = z[p + "step_activations"][keep]                     # NaN: apply_activation was not called on that layer
    for m in range(3):
        exp = np.where(np.isfinite(want[:, m]), want[:, m], 0.25)
        np.testing.assert_array_equal(got[m], np.tile(exp[:, None], (1, N_ELEM)), err_msg=f"layer {m}")
    assert {"nan_x", "nan_Q", "nan_w", "inf_v0", "time_just_past"} <= {labels[i] for i in keep}
    be.close()
