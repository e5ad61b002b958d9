"""BASELINE.json configs[2] and the per-GPU share of configs[4] at their FULL sizes (config[1] at
4096 envs is tests/test_gpu_parity.py::test_full_size_batch_properties): size-independent
properties of the whole batch — finite, bitwise repeatable, independent of the batch an env sits
in — plus spot parity of a few envs against the oracle over SEVERAL steps.  EVERY env of these
batches against the oracle (stepped with OpenMP over rods) for the first step(s) is
tests/test_gpu_full_batch_parity.py."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
RTOL = 1e-5


@pytest.fixture(scope="module")
def torch_gpu():
    import torch

    assert torch.cuda.is_available(), "-m gpu tests need the MI355X"
    return torch


def test_config3_4096_arms_of_100_elements(torch_gpu, hip_lib, oracle_built):
    """OctoArmSingle-style reach, 4096 envs x 100 elements, 714 substeps per env.step: the
    two-window kernel + the epilogue launch (softrod_window.hpp)."""
    import gym_softrobot_amd as gsa

    n, T = 4096, 2
    acts = np.random.default_rng(2).uniform(-6, 6, (T, n, 7)).astype(np.float32)

    def rollout(idx):
        env = gsa.make_vec("OctoArmSingle-v0", len(idx), n_elems=100)
        assert env.backend.cfg.n_elem == 100
        env.reset(seed=0)
        out = []
        for t in range(T):
            o, r, te, tr, _ = env.step(acts[t, idx])
            out.append((o.cpu().numpy().copy(), r.cpu().numpy().copy(), te.cpu().numpy().copy(), tr.cpu().numpy().copy()))
        st = env.backend.state_numpy()
        env.close()
        return out, st

    full, st = rollout(np.arange(n))
    again, _ = rollout(np.arange(n))
    for a, b in zip(full, again):                                    # bitwise run-to-run (K7)
        for x, y in zip(a, b):
            np.testing.assert_array_equal(x, y)
    obs, rew, term, trunc = full[-1]
    assert np.isfinite(obs).all() and np.isfinite(rew).all() and not trunc.any()
    # a few arms may end their episode early (|omega|_F > 250 is `invalid`, arm_single_env.py:270;
    # 100 elements whip harder than 50): that is the env, and the oracle must agree on which
    assert term.mean() < 0.05
    np.testing.assert_array_equal(obs[:, 16:23], acts[-1])            # _prev_action = the action just taken
    np.testing.assert_array_equal(obs[:, 23:25], np.broadcast_to(np.float32([1.0, 0.0]), (n, 2)))
    Q = st["Q"]
    QQt = np.einsum("eimk,ejmk->eijk", Q, Q)
    assert np.abs(QQt - np.eye(3)[None, :, :, None]).max() < 1e-11  # directors stay orthonormal
    spots = np.array([0, 1, 63, 64, 2047, 4095])
    few, _ = rollout(spots)                                           # batch independence (K8)
    for t in range(T):
        for x, y in zip(full[t], few[t]):
            np.testing.assert_array_equal(x[spots], y)
    ended = [int(i) for i in np.nonzero(full[0][2] | full[1][2])[0][:2]]
    for i in [0, 2047, 4095] + ended:                                 # spot parity vs the oracle
        r = oracle_built.OracleRod(gsa._capi.arm_single_config(1, n_elems=100))
        r.reset_arm()
        for t in range(T):
            o, rw, te, tr = r.env_step_arm(acts[t, i])
            assert te == bool(full[t][2][i]) and tr == bool(full[t][3][i])
        np.testing.assert_allclose(obs[i], o, rtol=RTOL, atol=2e-6)
        np.testing.assert_allclose(rew[i], rw, rtol=RTOL, atol=1e-7)
        np.testing.assert_allclose(st["x"][i], r.get("x"), rtol=RTOL, atol=1e-8)


def test_config5_share_1024_octoflat_envs(torch_gpu, hip_lib, oracle_built):
    """Octopus multi-arm, the per-GPU share of configs[4]: 1024 envs x 8 arms x 10 elements + head,
    2857 substeps per env.step, one workgroup of two wavefronts per env."""
    import gym_softrobot_amd as gsa

    n = 1024
    acts = np.random.default_rng(3).uniform(-22, 22, (n, 24)).astype(np.float32)

    def one_step(idx):
        env = gsa.make_vec("OctoFlat-v0", len(idx), numpy_output=True)
        env.reset(seed=[int(i) for i in idx])
        o, r, te, tr, _ = env.step(acts[idx])
        res = (o.copy(), r.copy(), te.copy(), tr.copy())
        tg = env.targets.copy()
        env.close()
        return res, tg

    full, targets = one_step(np.arange(n))
    again, _ = one_step(np.arange(n))
    for x, y in zip(full, again):
        np.testing.assert_array_equal(x, y)
    obs, rew, term, trunc = full
    assert obs.shape == (n, 461) and np.isfinite(obs).all() and np.isfinite(rew).all() and not trunc.any()
    spots = np.array([0, 1, 255, 256, 1023])
    few, _ = one_step(spots)
    for x, y in zip(full, few):
        np.testing.assert_array_equal(x[spots], y)
    for i in (0, 256, 1023):
        o = oracle_built.OracleOcto(gsa._capi.octo_flat_config(1))
        o.reset(targets[i])
        ob, rw, te, tr = o.env_step(acts[i])
        flat = np.concatenate([ob["individual"].ravel(), ob["shared"]])
        np.testing.assert_allclose(obs[i], flat, rtol=RTOL, atol=2e-6)
        np.testing.assert_allclose(rew[i], rw, rtol=RTOL, atol=1e-6)
        assert bool(term[i]) == te and bool(trunc[i]) == tr


def test_config4_total_batch_32768_softpendulum_envs_on_one_gpu(torch_gpu, hip_lib, oracle_built):
    """BASELINE configs[3]'s TOTAL batch (8 x 4096 envs) resident on one GPU (453 MB of state): the
    envs the eight shards would own are the same envs — env i seeded i whatever batch it sits in —
    so the first and last envs of every shard must equal, bit for bit, a small batch holding just
    them; spot parity against the oracle at both ends; finite everywhere."""
    import gym_softrobot_amd as gsa
    from gym_softrobot_amd.seeding import initial_angle, np_random

    n, T = 32768, 3
    acts = np.random.default_rng(1).uniform(-22, 22, (T, n, 1)).astype(np.float32)
    env = gsa.make_vec("SoftPendulum-v0", n)
    obs0, _ = env.reset(seed=0)
    full = []
    for t in range(T):
        o, r, te, tr, _ = env.step(acts[t])
        full.append((o.cpu().numpy().copy(), r.cpu().numpy().copy()))
    assert np.isfinite(full[-1][0]).all() and np.isfinite(full[-1][1]).all()
    env.close()
    spots = np.array([s * 4096 + k for s in range(8) for k in (0, 4095)])      # both ends of every shard
    for i in spots[[0, 7, 15]]:                                                   # each spot: its own env, seeded i
        small = gsa.make_vec("SoftPendulum-v0", 1)
        small.reset(seed=int(i))
        for t in range(T):
            o, r, _, _, _ = small.step(acts[t, i:i + 1])
            np.testing.assert_array_equal(o.cpu().numpy()[0], full[t][0][i])
            assert float(r.cpu().numpy()[0]) == full[t][1][i]
        small.close()
    for i in (0, n - 1):
        rod = oracle_built.OracleRod(gsa._capi.softpendulum_config(1))
        rod.reset_pendulum(initial_angle(np_random(int(i))[0]))
        for t in range(T):
            o, r, te, tr = rod.env_step(acts[t, i, 0])
        np.testing.assert_allclose(full[-1][0][i], o, rtol=RTOL, atol=1e-7)
        np.testing.assert_allclose(full[-1][1][i], r, rtol=RTOL, atol=1e-9)


def test_config5_full_8192_octoflat_envs_on_one_gpu(torch_gpu, hip_lib, oracle_built):
    """BASELINE configs[4] at its REAL size: all 8192 OctoFlat envs (65 536 arms + 8192 heads, 2048
    workgroups of four envs: eight resident rounds at one workgroup per CU) resident on ONE GPU — the batch
    the eight shards would own, env i seeded i whatever batch it sits in.  One whole env.step of 2857
    substeps: finite everywhere, bitwise repeatable, the first and last envs of every 1024-env shard equal
    — bit for bit — a small batch holding just them (batch independence at the shard ends), and three
    spot envs against the oracle for that first step (the horizon at which trajectory parity holds for
    OctoFlat: DESIGN.md §3; the regimes after it are tests/test_gpu_ensemble_parity.py's).
    Reference: /root/reference/gym_softrobot/envs/octopus/flat_env.py:58-61,78,315-408."""
    import gym_softrobot_amd as gsa

    n = 8192
    acts = np.random.default_rng(5).uniform(-22, 22, (n, 24)).astype(np.float32)

    def one_step(idx):
        env = gsa.make_vec("OctoFlat-v0", len(idx), numpy_output=True)
        assert env.backend.kernel_tier() == "softrod_octo_step_kernel<zup,2 waves,4 envs/wg>"
        env.reset(seed=[int(i) for i in idx])
        o, r, te, tr, info = env.step(acts[idx])
        res = (o.copy(), r.copy(), te.copy(), tr.copy())
        tg, st = env.targets.copy(), env.backend.octo_state_numpy()
        env.close()
        return res, tg, st

    full, targets, st = one_step(np.arange(n))
    again, _, st2 = one_step(np.arange(n))
    for x, y in zip(full, again):                                      # bitwise run-to-run
        np.testing.assert_array_equal(x, y)
    for k in ("x", "v", "w", "Q", "head_x", "head_v", "head_Q", "head_w"):
        np.testing.assert_array_equal(st[k], st2[k])
    obs, rew, term, trunc = full
    assert obs.shape == (n, 461) and np.isfinite(obs).all() and np.isfinite(rew).all()
    assert not trunc.any() and not term.any()
    np.testing.assert_allclose(st["time"], 2857 * 7.0e-5, rtol=1e-12)
    Q = st["Q"]                                                         # (N, A, 3, 3, n): directors stay orthonormal
    QQt = np.einsum("eaimk,eajmk->eaijk", Q, Q)
    assert np.abs(QQt - np.eye(3)[None, None, :, :, None]).max() < 1e-10
    hq = st["head_Q"]                                                   # BodyBoundaryCondition: d3 = e_z, d1, d2 unit in the plane
    assert np.abs(hq[:, 2] - np.array([0.0, 0.0, 1.0])).max() == 0.0 and np.abs(hq[:, :2, 2]).max() == 0.0
    assert np.abs(np.linalg.norm(hq[:, :2, :2], axis=2) - 1.0).max() < 1e-14
    assert np.abs(st["head_x"][:, 2]).max() == 0.0 and np.abs(st["head_v"][:, 2]).max() == 0.0
    # the arm-crossing penalty did its work somewhere in the batch (reward = forward - 0.02 * crossings)
    assert (rew < -0.05).any()
    spots = np.array([s * 1024 + k for s in range(8) for k in (0, 1023)])    # both ends of every shard
    few, _, _ = one_step(spots)
    for x, y in zip(full, few):
        np.testing.assert_array_equal(x[spots], y)
    for i in (0, 4095, 8191):
        o = oracle_built.OracleOcto(gsa._capi.octo_flat_config(1))
        o.reset(targets[i])
        ob, rw, te, tr = o.env_step(acts[i])
        flat = np.concatenate([ob["individual"].ravel(), ob["shared"]])
        np.testing.assert_allclose(obs[i], flat, rtol=RTOL, atol=2e-6)
        np.testing.assert_allclose(rew[i], rw, rtol=RTOL, atol=1e-6)
        assert bool(term[i]) == te and bool(trunc[i]) == tr
