"""OctoCrawl-v0 / OctoArmTwo-v0 / OctoReach-v0 (the muscle octopus of octopus/build_muscle_octopus.py) on the GPU against
the oracle: the C body (oracle/octoflat_oracle.inc.c oracle_mocto_*) under the NumPy env code of tests/oracle_mocto.py.
PARITY UNPINNED underneath (the restated COOMM muscle law); what these tests hold is HIP == oracle at rtol 1e-5 on the
same inputs: set_action's mapping (sucker index / ratios, per-element activations), 800 substeps of 8 (2) tapered muscle
arms joined to the head, get_state's layout, reward and flags."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-5
IDS = {"OctoCrawl-v0": "ENV_CRAWL", "OctoArmTwo-v0": "ENV_ARM_TWO", "OctoReach-v0": "ENV_REACH"}


@pytest.fixture(scope="module")
def torch_gpu():
    import torch

    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    return torch


def _pair(env_id, n, **kw):
    import gym_softrobot_amd as gsa
    from tests.oracle_backend import OracleBackend

    kind = getattr(gsa._capi, IDS[env_id])
    env = gsa.make_vec(env_id, n, **kw)
    cfg = gsa._capi.muscle_octopus_config(kind, n, final_time=kw.get("final_time"))
    ref = gsa.make_vec(env_id, n, backend=OracleBackend(cfg), numpy_output=True, **kw)
    return env, ref


def _actions(env_id, rng, n, dim):
    a = rng.uniform(0.0, 1.0, (n, dim)).astype(np.float32)
    if env_id == "OctoReach-v0":
        a *= 0.6                       # every element of every layer driven at random: keep the restated cubic in its range
    return a


@pytest.mark.parametrize("env_id", list(IDS))
def test_env_steps_match_the_oracle(torch_gpu, hip_lib, oracle_built, env_id):
    n, T = 3, 3
    env, ref = _pair(env_id, n)
    assert "softrod_mocto_action_kernel" in env.backend.kernel_tier()
    o, _ = env.reset(seed=3)
    o2, _ = ref.reset(seed=3)
    # (kappa of the freshly built straight arms: the constructor's inv_rotate leaves +-4e-14 in the oracle, the device reset
    # writes the zero it stands for)
    np.testing.assert_allclose(o.cpu().numpy(), o2, rtol=0, atol=1e-12)
    rng = np.random.default_rng(5)
    for t in range(T):
        a = _actions(env_id, rng, n, env.action_dim)
        o, r, te, tr, info = env.step(a)
        o2, r2, te2, tr2, info2 = ref.step(a)
        torch_gpu.cuda.synchronize()
        np.testing.assert_allclose(o.cpu().numpy(), o2, rtol=RTOL, atol=5e-6, err_msg=f"obs step {t}")
        np.testing.assert_allclose(r.cpu().numpy(), r2, rtol=RTOL, atol=2e-6, err_msg=f"reward step {t}")
        np.testing.assert_array_equal(te.cpu().numpy(), te2)
        np.testing.assert_array_equal(tr.cpu().numpy(), tr2)
        np.testing.assert_allclose(np.asarray(info["time"]), np.asarray(info2["time"]), rtol=0, atol=0)
    # the body itself: every arm's state and the head's
    st = env.backend.state()
    ne, na, seg = env.n_elems, env.n_arm, int(st["arm_stride"])
    assert seg == 32
    x = st["position"].cpu().numpy()            # [3][n][64 nw]
    v = st["velocity"].cpu().numpy()
    w = st["omega"].cpu().numpy()
    Q = st["director"].cpu().numpy()
    head = st["head"].cpu().numpy()
    sidx = st["sucker_index"].cpu().numpy()
    srat = st["sucker_ratio"].cpu().numpy()
    mact = st["muscle_activation"].cpu().numpy()
    for i, q in enumerate(ref.backend.rods):
        for a in range(na):
            arm = q.arm(a)
            sl = slice(a * seg, a * seg + ne + 1)
            se = slice(a * seg, a * seg + ne)
            np.testing.assert_allclose(x[:, i, sl], arm.get("x"), rtol=RTOL, atol=1e-8, err_msg=f"x env {i} arm {a}")
            np.testing.assert_allclose(v[:, i, sl], arm.get("v"), rtol=RTOL, atol=2e-6, err_msg=f"v env {i} arm {a}")
            np.testing.assert_allclose(w[:, i, se], arm.get("w"), rtol=RTOL, atol=2e-4, err_msg=f"w env {i} arm {a}")
            np.testing.assert_allclose(Q[:, i, se].reshape(3, 3, ne), arm.get("Q"), rtol=RTOL, atol=1e-7, err_msg=f"Q env {i} arm {a}")
            np.testing.assert_array_equal(sidx[:, i * na + a], arm.get("sucker_index").astype(np.int32))
            np.testing.assert_allclose(srat[:, i * na + a], arm.get("sucker_ratio"), rtol=1e-15, atol=0)
            for m in range(3):
                np.testing.assert_allclose(mact[m, i, se], arm.get("muscle_activation")[m], rtol=1e-12, atol=1e-15)
        h = q.head()
        np.testing.assert_allclose(head[0:3, i], h["x"], rtol=RTOL, atol=1e-9)
        np.testing.assert_allclose(head[3:6, i], h["v"], rtol=RTOL, atol=1e-6)
        np.testing.assert_allclose(head[6:15, i].reshape(3, 3), h["Q"], rtol=RTOL, atol=1e-9)
    if env_id == "OctoReach-v0":                 # OneEndFixedBC: the head has not moved at all
        np.testing.assert_array_equal(head[0:3, :].T, np.tile([0.0, 0.0, -0.013], (n, 1)))
        np.testing.assert_array_equal(head[3:6], 0.0)
    else:
        assert np.abs(head[0:2]).max() > 1e-7    # the arms drag the head
    env.close()
    ref.close()


@pytest.mark.parametrize("env_id", list(IDS))
def test_device_autoreset_and_truncation_follow_the_host_double(torch_gpu, hip_lib, oracle_built, env_id):
    """final_time shortened to two env.steps: the third step truncates, NEXT_STEP auto-reset restarts the env on the device
    (fresh arms, suckers, activations; ReachEnv's next target from the staged draws) exactly as the host double does."""
    n = 2
    env, ref = _pair(env_id, n, final_time=0.07, autoreset="device")
    o, _ = env.reset(seed=11)
    o2, _ = ref.reset(seed=11)
    rng = np.random.default_rng(2)
    seen_trunc = False
    for t in range(6):
        a = _actions(env_id, rng, n, env.action_dim)
        o, r, te, tr, info = env.step(a)
        o2, r2, te2, tr2, info2 = ref.step(a)
        torch_gpu.cuda.synchronize()
        np.testing.assert_allclose(o.cpu().numpy(), o2, rtol=RTOL, atol=5e-6, err_msg=f"obs step {t}")
        np.testing.assert_allclose(r.cpu().numpy(), r2, rtol=RTOL, atol=2e-6, err_msg=f"reward step {t}")
        np.testing.assert_array_equal(te.cpu().numpy(), te2)
        np.testing.assert_array_equal(tr.cpu().numpy(), tr2)
        seen_trunc = seen_trunc or bool(tr2.any())
    assert seen_trunc
    np.testing.assert_allclose(env.targets, ref.targets, rtol=0, atol=0)
    env.close()
    ref.close()


FIX = {"OctoCrawl-v0": "crawl_", "OctoArmTwo-v0": "armtwo_", "OctoReach-v0": "reach_"}


@pytest.mark.parametrize("env_id", list(IDS))
def test_env_code_replays_the_executed_reference(torch_gpu, hip_lib, env_id):
    """The reference's own CrawlEnv / ArmTwoEnv / ReachEnv code, executed (tools/make_muscle_octopus_golden.py ->
    tests/golden/ref_muscle_octopus.npz), replayed through the HIP library: one env per fixture row in a handle of
    n_substeps = 0 — softrod_step is then set_action (softrod_mocto_action_kernel) and get_state / reward / flags
    (softrod_mocto_epilogue_kernel) on the state installed through the state view.  tests/test_muscle_octopus.py replays the
    same rows through the oracle's env code on the CPU."""
    from pathlib import Path

    import gym_softrobot_amd as gsa
    from gym_softrobot_amd import _capi
    from gym_softrobot_amd.backend import HipRodBackend

    torch = torch_gpu
    z = np.load(Path(__file__).resolve().parent / "golden" / "ref_muscle_octopus.npz")
    p = FIX[env_id]
    kind = getattr(_capi, IDS[env_id])
    labels = [str(s) for s in z[p + "step_label"]]
    N = len(labels)
    cfg = _capi.muscle_octopus_config(kind, N)
    cfg.n_substeps = 0
    be = HipRodBackend(cfg, 0)
    radii = _capi.muscle_octopus_radii(20)
    be.set_radius_profile(radii)
    be.set_muscle_layers(*_capi.es_muscle_layers(radii, 0.013))
    tgt = np.zeros((N, 4))                       # x, y, z and the episode's final_time (0: the config's)
    tgt[:, :3] = z[p + "reset_target"] if kind == _capi.ENV_REACH else [5.0, 0.0, 0.0]
    be.reset_octo(tgt)
    torch.cuda.synchronize()
    # the reset observation is the reference's get_state on the freshly built body
    np.testing.assert_allclose(be.observe(None).cpu().numpy()[0], z[p + "reset_obs"], rtol=0, atol=1e-12)
    st = be.state()
    dev, na, n, seg = be.device, int(cfg.n_arm), 20, int(st["arm_stride"])

    def put(name, arr, width):                   # (N, na, comps, width) -> rows [comps][N][na * seg]
        for a in range(na):
            t = torch.from_numpy(np.ascontiguousarray(np.moveaxis(arr[:, a], 0, 1))).to(dev)
            st[name][: t.shape[0], :, a * seg: a * seg + width] = t

    put("position", z[p + "step_x"], n + 1)
    put("velocity", z[p + "step_v"], n + 1)
    put("kappa", z[p + "step_kappa"], n - 1)
    head = st["head"]
    head[0:3] = torch.from_numpy(z[p + "step_hx"].T.copy()).to(dev)
    head[3:6] = torch.from_numpy(z[p + "step_hv"].T.copy()).to(dev)
    head[6:15] = torch.from_numpy(z[p + "step_hQ"].reshape(N, 9).T.copy()).to(dev)
    st["time"][:] = torch.from_numpy(z[p + "step_time"]).to(dev)
    st["env_aux"][3:5] = torch.from_numpy(z[p + "step_pre_hx"][:, :2].T.copy()).to(dev)      # xposbefore
    st["prev_action"][:, : be.action_dim] = torch.from_numpy(z[p + "step_prev_action_before"].astype(np.float32)).to(dev)
    st["prev_kappa"][:] = torch.from_numpy(z[p + "step_prev_kappa_before"].reshape(N, -1).astype(np.float32)).to(dev)
    obs, rew, term, trunc = be.step(z[p + "step_action"].astype(np.float32))
    torch.cuda.synchronize()
    obs, rew = obs.cpu().numpy(), rew.cpu().numpy()
    for k, label in enumerate(labels):
        np.testing.assert_array_equal(obs[k], z[p + "step_obs"][k], err_msg=label)
        want = z[p + "step_reward"][k]
        assert (np.isnan(want) and np.isnan(rew[k])) or rew[k] == pytest.approx(want, rel=1e-12, abs=1e-14), (label, rew[k], want)
    np.testing.assert_array_equal(term.cpu().numpy().astype(bool), z[p + "step_terminated"])
    np.testing.assert_array_equal(trunc.cpu().numpy().astype(bool), z[p + "step_truncated"])
    ns = int(cfg.n_suckers)
    sidx = st["sucker_index"].cpu().numpy().reshape(4, N, na)
    srat = st["sucker_ratio"].cpu().numpy().reshape(4, N, na)
    for j in range(ns):
        np.testing.assert_array_equal(sidx[j], z[p + "step_sucker_index"][:, :, j])
        np.testing.assert_array_equal(srat[j], z[p + "step_sucker_ratio"][:, :, j])
    mact = st["muscle_activation"].cpu().numpy()                # [4][N][na * seg]
    acts = z[p + "step_activations"]                            # (N, na, 3, n); NaN: the layer received nothing
    for a in range(na):
        for m in range(3):
            got = mact[m, :, a * seg: a * seg + n]
            want = np.where(np.isfinite(acts[:, a, m]), acts[:, a, m], 0.0)
            np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-15, err_msg=f"arm {a} layer {m}")
    if kind == _capi.ENV_ARM_TWO:
        np.testing.assert_array_equal(st["prev_kappa"].cpu().numpy().reshape(N, na, n - 1), z[p + "step_prev_kappa_after"])
    be.close()


@pytest.mark.parametrize("env_id,n", [("OctoCrawl-v0", 256), ("OctoArmTwo-v0", 512), ("OctoReach-v0", 128)])
def test_a_whole_batch_one_step(torch_gpu, hip_lib, oracle_built, env_id, n):
    """Hundreds of envs (several workgroups per CU, every XCD), one env.step each under its own random action: every
    env's observation and reward against the oracle — a mapping bug (an env reading its neighbour's sucker row, a wave
    its neighbour's activation row) would show as a pattern over the env index, not as noise."""
    env, ref = _pair(env_id, n)
    env.reset(seed=21)
    ref.reset(seed=21)
    a = _actions(env_id, np.random.default_rng(8), n, env.action_dim)
    o, r, te, tr, _ = env.step(a)
    o2, r2, te2, tr2, _ = ref.step(a)
    torch_gpu.cuda.synchronize()
    o, r = o.cpu().numpy(), r.cpu().numpy()
    band = 5e-6 + RTOL * np.abs(o2)
    per_env = (np.abs(o - o2) / band).max(axis=1)
    assert per_env.max() <= 1.0, (int(per_env.argmax()), float(per_env.max()), np.nonzero(per_env > 1.0)[0][:10])
    np.testing.assert_allclose(r, r2, rtol=RTOL, atol=2e-6)
    np.testing.assert_array_equal(te.cpu().numpy(), te2)
    np.testing.assert_array_equal(tr.cpu().numpy(), tr2)
    env.close()
    ref.close()


def test_crawl_random_final_time_on_the_device(torch_gpu, hip_lib):
    """config_random_final_time: every env's own final_time travels with its reset record (env_aux row 5) and truncation
    follows it — one env.step (0.04 s) from a clock set just short of each env's draw."""
    import gym_softrobot_amd as gsa

    n = 4
    env = gsa.make_vec("OctoCrawl-v0", n, config_random_final_time=True, numpy_output=True)
    env.reset(seed=0)
    z = np.load(__import__("pathlib").Path(__file__).resolve().parent / "golden" / "ref_muscle_octopus.npz")
    np.testing.assert_array_equal(env.final_times[:2], z["crawl_random_final_time"][:2])
    st = env.backend.state()
    np.testing.assert_array_equal(st["env_aux"][5].cpu().numpy(), env.final_times)
    back = np.array([0.02, 0.06, 0.039, 5.0])          # 0.039: 0.04 later the clock is PAST the draw by 1 ms
    st["time"][:] = torch_gpu.from_numpy(env.final_times - back).to(env.backend.device)
    o, r, te, tr, info = env.step(np.zeros((n, 24), np.float32))
    assert list(tr) == [True, False, True, False] and not te.any()
    env.close()
    plain = gsa.make_vec("OctoCrawl-v0", 2, numpy_output=True)
    plain.reset(seed=0)
    np.testing.assert_array_equal(plain.backend.state()["env_aux"][5].cpu().numpy(), [0.0, 0.0])
    plain.close()


@pytest.mark.parametrize("env_id", list(IDS))
def test_ten_steps_stay_on_the_oracle_trajectory(torch_gpu, hip_lib, oracle_built, env_id):
    """No contact, no friction: the muscle octopus is smooth dynamics, so trajectory parity holds over many env.steps —
    ten here (8000 substeps), observations and rewards at 1e-5 throughout."""
    n, T = 2, 10
    env, ref = _pair(env_id, n)
    env.reset(seed=5)
    ref.reset(seed=5)
    rng = np.random.default_rng(17)
    worst = 0.0
    for t in range(T):
        a = _actions(env_id, rng, n, env.action_dim) * 0.7
        o, r, te, tr, _ = env.step(a)
        o2, r2, te2, tr2, _ = ref.step(a)
        torch_gpu.cuda.synchronize()
        o = o.cpu().numpy()
        worst = max(worst, float((np.abs(o - o2) / (5e-6 + RTOL * np.abs(o2))).max()))
        np.testing.assert_allclose(o, o2, rtol=RTOL, atol=5e-6, err_msg=f"obs step {t}")
        np.testing.assert_allclose(r.cpu().numpy(), r2, rtol=RTOL, atol=2e-6, err_msg=f"reward step {t}")
        np.testing.assert_array_equal(te.cpu().numpy(), te2)
    assert worst <= 1.0
    env.close()
    ref.close()


def test_symmetric_actuation_leaves_the_head_where_it_is(torch_gpu, hip_lib):
    """Known answer for the whole body: eight identical arms at 45-degree spacing, the same transverse activation in
    all of them, suckers released (ratio 0): the joint loads on the head cancel by symmetry — it stays put while every
    arm extends by the same amount."""
    import gym_softrobot_amd as gsa

    env = gsa.make_vec("OctoCrawl-v0", 2, numpy_output=True)
    env.reset(seed=0)
    a = np.tile(np.array([0.0, 0.5, 0.0], np.float32), (2, 8))          # location 0, activation 0.5, reduction ratio 0
    for _ in range(3):
        env.step(a)
    st = env.backend.octo_state_numpy()
    assert np.abs(st["head_x"][:, :2]).max() < 1e-9 and np.abs(st["head_v"][:, :2]).max() < 1e-7
    tips = st["x"][0, :, :, 20]                                           # (arm, 3)
    bases = st["x"][0, :, :, 0]
    reach = np.linalg.norm(tips - bases, axis=1)
    assert reach.min() > 0.2505 and reach.max() - reach.min() < 1e-9     # every arm longer than its rest length, all alike
    env.close()
