"""On-device policy-in-the-loop (VERDICT r4 "next" #5): obs -> 2 x 64 MLP (torch, the
env's stream) -> actions -> softrod_step with device-side auto-reset, no host synchronisation anywhere
in the loop.  Stream-ordering proof: the rollout is BIT-IDENTICAL to the same loop with a device
synchronise after every call — so nothing in the asynchronous path reads a buffer before the kernel
that fills it has run, or overwrites one (the observation buffer is a view the next step reuses)
before its reader has.  Mirrors the rollout of /root/reference/examples/soft_pendulum_3d/train_ppo.py:26-38
and /root/reference/gym_softrobot/debug/make.py:14-23."""
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parents[1]


@pytest.fixture(scope="module")
def bench_mod(hip_lib):
    import torch

    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    sys.path.insert(0, str(ROOT))
    import bench

    return bench


def _rollout(bench, env_id, n, steps, sync, amax, stream=None, **kw):
    import torch

    import gym_softrobot_amd as gsa

    ctx = torch.cuda.stream(stream) if stream is not None else torch.cuda.stream(torch.cuda.current_stream())
    with ctx:
        env = gsa.make_vec(env_id, n, device=0, autoreset="device", **kw)
        obs, _ = env.reset(seed=11)
        policy = bench.make_policy(torch, env.backend.device, env.obs_dim, env.action_dim, amax)
        rec = []
        bench.policy_loop(torch, env, policy, obs, steps, sync_every_call=sync, record=rec)
        torch.cuda.synchronize()
        consumed = int(env.backend.queue_status()[0].sum())
        env.close()
    return rec, consumed


@pytest.mark.parametrize("env_id,n,steps,amax,kw", [
    ("SoftPendulum-v0", 1024, 140, 22.0, {}),                          # truncation on step 126: restarts inside the loop
    ("SoftPendulum-v0", 256, 24, 22.0, {"final_time": 0.19}),          # 5-step episodes: many restarts, queue top-ups
    ("SoftPendulum3D-v0", 256, 30, 1.0, {}),
    ("OctoArmSingle-v0", 128, 12, 6.0, {}),
], ids=["pendulum-episode", "pendulum-short-episodes", "pendulum3d", "arm"])
def test_async_policy_loop_equals_the_synchronised_one(bench_mod, env_id, n, steps, amax, kw):
    a, ca = _rollout(bench_mod, env_id, n, steps, False, amax, **kw)
    b, cb = _rollout(bench_mod, env_id, n, steps, True, amax, **kw)
    assert len(a) == len(b) == steps and ca == cb
    for t, (x, y) in enumerate(zip(a, b)):
        for name, u, v in zip(("obs", "reward", "terminated", "truncated"), x, y):
            assert np.array_equal(u.numpy(), v.numpy(), equal_nan=True), f"step {t + 1}: {name} differs"
    if "final_time" in kw or steps > 126:
        assert ca > 0, "the loop was meant to cross episode ends"
        assert any(bool(x[3].any()) for x in a)


def test_policy_loop_on_a_side_stream(bench_mod):
    """The same proof on a NON-DEFAULT (non-blocking) torch stream: softrod_step takes the caller's current
    stream, and every library-side helper (queue top-ups, status reads) must order itself on it."""
    import torch

    s = torch.cuda.Stream()
    a, ca = _rollout(bench_mod, "SoftPendulum-v0", 512, 135, False, 22.0, stream=s)
    b, cb = _rollout(bench_mod, "SoftPendulum-v0", 512, 135, True, 22.0)
    assert ca == cb and ca > 0
    for t, (x, y) in enumerate(zip(a, b)):
        for u, v in zip(x, y):
            assert np.array_equal(u.numpy(), v.numpy(), equal_nan=True), f"step {t + 1}"


@pytest.mark.parametrize("env_id,n,steps,amax,kw", [
    ("SoftPendulum-v0", 1024, 140, 22.0, {"autoreset": "device"}),                      # crosses the episode end on step 126
    ("SoftPendulum-v0", 256, 30, 22.0, {"autoreset": "device", "final_time": 0.19}),    # 5-step episodes: restarts + top-ups between replays
    ("SoftPendulum3D-v0", 256, 20, 1.0, {}),
    ("OctoFlat-v0", 8, 2, 22.0, {}),
], ids=["pendulum-episode", "pendulum-short-episodes", "pendulum3d-no-autoreset", "octoflat"])
def test_policy_step_as_one_hip_graph_equals_the_eager_loop(bench_mod, env_id, n, steps, amax, kw):
    """`capture_policy_step`: policy + softrod_step captured ONCE into a HIP graph (north_star: hipGraphs for the
    launch-bound inner loop) and replayed per env.step — bit-identical to the eager loop, auto-reset and its queue
    top-ups (outside the graph, between replays) included."""
    import torch

    import gym_softrobot_amd as gsa

    def eager():
        env = gsa.make_vec(env_id, n, device=0, **kw)
        obs, _ = env.reset(seed=5)
        policy = bench_mod.make_policy(torch, env.backend.device, env.obs_dim, env.action_dim, amax)
        rec = []
        bench_mod.policy_loop(torch, env, policy, obs, steps, record=rec)
        env.close()
        return rec

    def graphed():
        env = gsa.make_vec(env_id, n, device=0, **kw)
        env.reset(seed=5)
        policy = bench_mod.make_policy(torch, env.backend.device, env.obs_dim, env.action_dim, amax)
        replay = env.capture_policy_step(policy)
        rec = []
        for _ in range(steps):
            out = replay()
            rec.append(tuple(x.detach().cpu().clone() for x in out))
        torch.cuda.synchronize()
        env.close()
        return rec

    a, b = eager(), graphed()
    assert len(a) == len(b) == steps
    for t, (x, y) in enumerate(zip(a, b)):
        for name, u, v in zip(("obs", "reward", "terminated", "truncated"), x, y):
            assert np.array_equal(u.numpy(), v.numpy().astype(u.numpy().dtype), equal_nan=True), f"step {t + 1}: {name} differs"
    if kw.get("autoreset") and (steps > 126 or "final_time" in kw):
        assert any(bool(x[3].any()) for x in b), "the graphed loop was meant to cross episode ends"


def test_capture_refuses_host_autoreset(bench_mod):
    import torch

    import gym_softrobot_amd as gsa

    env = gsa.make_vec("SoftPendulum-v0", 4, device=0, autoreset=True)
    env.reset(seed=0)
    with pytest.raises(NotImplementedError):
        env.capture_policy_step(bench_mod.make_policy(torch, env.backend.device, 4, 1, 22.0))
    env.close()
