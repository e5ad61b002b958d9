"""Product-library hygiene (VERDICT r4 "next" #7):

* the A/B switches of the measurement builds (SOFTROD_OCTO_ONE_WAVE, SOFTROD_NO_WINDOW, ...) change the
  kernel tier ONLY together with SOFTROD_DEBUG_SWITCHES=1 — a stray variable in a product process does
  nothing (softrod_kernel_tier, ABI v15, reports what softrod_create selected);
* softrod_autoreset_enable frees what it allocated when it fails mid-way and can be retried.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch_gpu():
    import torch

    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    return torch


def test_stray_switches_do_not_change_the_kernel_tier(torch_gpu, hip_lib, monkeypatch):
    from gym_softrobot_amd import _capi
    from gym_softrobot_amd.backend import HipRodBackend

    octo, arm = _capi.octo_flat_config(4), _capi.arm_single_config(4, n_elems=100)
    plain = {}
    for name, cfg in (("octo", octo), ("arm", arm)):
        be = HipRodBackend(cfg, device=0)
        plain[name] = be.kernel_tier()
        be.close()
    assert plain["octo"] == "softrod_octo_step_kernel<zup,2 waves,4 envs/wg>"
    assert plain["arm"].startswith("softrod_step_window_kernel<ArmSingle,4 rods/wg>")
    stray = {"SOFTROD_OCTO_ONE_WAVE": "1", "SOFTROD_OCTO_ONE_ENV_PER_BLOCK": "1", "SOFTROD_NO_WINDOW": "1",
             "SOFTROD_WINDOW_PAIRED": "0", "SOFTROD_WINDOW_REFRESH": "1"}
    for k, v in stray.items():
        monkeypatch.setenv(k, v)
    for name, cfg in (("octo", octo), ("arm", arm)):                 # stray variables alone: nothing changes
        be = HipRodBackend(cfg, device=0)
        assert be.kernel_tier() == plain[name]
        be.close()
    monkeypatch.setenv("SOFTROD_DEBUG_SWITCHES", "1")                 # with the gate they are honoured
    be = HipRodBackend(octo, device=0)
    assert "octo1w" in be.kernel_tier()
    be.close()
    be = HipRodBackend(arm, device=0)
    assert "window" not in be.kernel_tier() and "epl=2" in be.kernel_tier()
    be.close()
    monkeypatch.delenv("SOFTROD_NO_WINDOW")
    be = HipRodBackend(arm, device=0)
    assert "1 rod/wg,s_barrier" in be.kernel_tier() and "refresh=1 " in be.kernel_tier()
    be.close()


def test_tiers_of_the_registered_envs(torch_gpu, hip_lib):
    import gym_softrobot_amd as gsa

    want = {"SoftPendulum-v0": "softrod_step_fast_kernel<SoftPendulum,epl=1>",
            "SoftPendulum3D-v0": "softrod_step_fast_kernel<SoftPendulum3D,epl=1>",
            "OctoArmSingle-v0": "softrod_step_fast_kernel<ArmSingle,epl=1>",
            "OctoFlat-v0": "softrod_octo_step_kernel<zup,2 waves,4 envs/wg>"}
    for env_id, tier in want.items():
        env = gsa.make_vec(env_id, 2, device=0)
        assert env.backend.kernel_tier() == tier
        env.close()


@pytest.mark.parametrize("fail_at", [1, 3, 7, 12, 15, 19, 20])
def test_autoreset_enable_cleans_up_after_a_failure_and_can_be_retried(torch_gpu, hip_lib, monkeypatch, fail_at):
    """The fail_at-th HIP call of the set-up fails (injected; honoured only under SOFTROD_DEBUG_SWITCHES=1):
    the call reports it, nothing stays allocated (free device memory is back to what it was, repeatedly),
    the handle still steps WITHOUT auto-reset exactly like an untouched one, and a retry succeeds and
    auto-resets exactly like a handle enabled at the first attempt."""
    import gym_softrobot_amd as gsa
    from gym_softrobot_amd import _capi

    n, depth = 64, 512                           # ~ 64 * 512 * 18 * 8 B = 4.7 MB per ring: visible in hipMemGetInfo
    env = gsa.make_vec("SoftPendulum-v0", n, device=0, final_time=0.1)      # truncation on the third step
    ref = gsa.make_vec("SoftPendulum-v0", n, device=0, final_time=0.1)
    env.reset(seed=3)
    ref.reset(seed=3)
    torch_gpu.cuda.synchronize()
    monkeypatch.setenv("SOFTROD_DEBUG_SWITCHES", "1")
    monkeypatch.setenv("SOFTROD_DEBUG_FAIL_AUTORESET_CALL", str(fail_at))
    free0 = torch_gpu.cuda.mem_get_info(0)[0]
    for _ in range(4):
        with pytest.raises(_capi.SoftrodError, match="injected failure at call"):
            env.backend.autoreset_enable(depth)
    torch_gpu.cuda.synchronize()
    free1 = torch_gpu.cuda.mem_get_info(0)[0]
    assert free0 - free1 < 2 << 20, f"{(free0 - free1) / 2**20:.1f} MiB still allocated after four failed attempts"
    acts = np.random.default_rng(0).uniform(-22, 22, (6, n, 1)).astype(np.float32)
    for t in range(2):                             # the failed attempts left the stepping path alone
        a, b = env.step(acts[t]), ref.step(acts[t])
        for x, y in zip(a[:4], b[:4]):
            np.testing.assert_array_equal(x.cpu().numpy(), y.cpu().numpy())
    monkeypatch.delenv("SOFTROD_DEBUG_FAIL_AUTORESET_CALL")
    monkeypatch.delenv("SOFTROD_DEBUG_SWITCHES")
    env.backend.autoreset_enable(4)                # the retry
    ref.backend.autoreset_enable(4)
    with pytest.raises(_capi.SoftrodError, match="already enabled"):
        env.backend.autoreset_enable(4)
    th = np.linspace(1.5, 1.6, n)
    for be in (env.backend, ref.backend):
        be.queue_push(th[:, None], np.ones(n, np.int32))
    restarted = False
    for t in range(2, 6):                          # step 3 truncates every env, step 4 restarts it from the queue
        a, b = env.backend.step(acts[t]), ref.backend.step(acts[t])
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x.cpu().numpy(), y.cpu().numpy())
        if t == 2:
            assert a[3].cpu().numpy().all()
        if t == 3:
            restarted = not a[3].cpu().numpy().any()
    assert restarted, "the retried auto-reset did not restart the truncated envs"
    consumed, underflow = env.backend.queue_status()
    assert (np.asarray(consumed) == 1).all() and int(underflow) == 0
    env.close()
    ref.close()
