"""The reference's own env tests (tests/envs/test_envs.py, tests/envs/test_determinism.py), run
over every env this package registers — the same assertions, on `gym_softrobot_amd.make`:
observations inside the declared space and of its dtype, scalar reward, bool flags, dict info;
same seed and same sampled actions twice -> identical observations, rewards and flags."""
import numpy as np
import pytest

import gym_softrobot_amd as gsa
from gym_softrobot_amd.spaces import Box

pytestmark = pytest.mark.gpu
ENV_IDS = sorted(gsa.registered())


def _equal(a, b, prefix=""):
    if isinstance(a, dict):
        assert sorted(a) == sorted(b), prefix
        for k in a:
            _equal(a[k], b[k], f"{prefix}{k}: ")
    else:
        np.testing.assert_array_equal(np.asarray(a), np.asarray(b), err_msg=prefix)


def _check_obs(space, ob, what):
    assert space.contains(ob), f"{what}: {ob!r} not in space"
    if isinstance(space, Box):
        assert ob.dtype == space.dtype, f"{what} dtype: {ob.dtype}, expected: {space.dtype}"


@pytest.mark.parametrize("env_id", ENV_IDS)
def test_env(hip_lib, env_id):                       # tests/envs/test_envs.py:19-59
    env = gsa.make(env_id)
    ob_space, act_space = env.observation_space, env.action_space
    ob, info = env.reset()
    assert isinstance(info, dict)
    _check_obs(ob_space, ob, "Reset observation")
    a = act_space.sample()
    observation, reward, terminated, truncated, _info = env.step(a)
    _check_obs(ob_space, observation, "Step observation")
    assert np.isscalar(reward), f"{reward} is not a scalar for {env}"
    assert isinstance(terminated, bool)
    assert isinstance(truncated, bool)
    assert isinstance(_info, dict)
    env.close()


@pytest.mark.parametrize("env_id", ENV_IDS)
def test_determinism(hip_lib, env_id):               # tests/envs/test_determinism.py:7-60
    runs = []
    for _ in range(2):
        env = gsa.make(env_id)
        initial, _ = env.reset(seed=0)
        env.action_space.seed(0)
        actions = [env.action_space.sample() for _ in range(3)]
        responses = [env.step(a) for a in actions]
        env.close()
        runs.append((initial, actions, responses))
    (i1, a1, r1), (i2, a2, r2) = runs
    for x, y in zip(a1, a2):
        _equal(x, y, "action ")
    _equal(i1, i2, "initial observation ")
    for k, ((o1, w1, t1, x1, _), (o2, w2, t2, x2, _)) in enumerate(zip(r1, r2)):
        _equal(o1, o2, f"[{k}] ")
        assert w1 == w2 and t1 == t2 and x1 == x2, f"[{k}]"


@pytest.mark.parametrize("env_id", ENV_IDS)
def test_make_vec_covers_every_registered_id(hip_lib, env_id):
    """`make_vec(id, n)` exists for every id `make(id)` knows (gym_softrobot/__init__.py:6-15,27-30,
    60-63,74-80 — OctoFlatLite-v0 included), is configured exactly like the single env (same
    softrod_config but for the batch size), and its env 0 reproduces the single env of that seed."""
    import torch

    n = 3
    vec = gsa.make_vec(env_id, n, numpy_output=True)
    one = gsa.make(env_id)
    c = vec.cfg.copy()
    c.n_envs = 1
    assert bytes(c) == bytes(one._vec.cfg), "make_vec and make configure different physics"
    obs, _ = vec.reset(seed=5)
    ob1, _ = one._vec.reset(seed=5)
    np.testing.assert_array_equal(np.asarray(obs[0]), np.asarray(ob1[0]))
    a = np.random.default_rng(0).uniform(vec.action_low, vec.action_high, (n, vec.action_dim)).astype(np.float32)
    if getattr(vec, "mode", None) == 0:          # OctoArmPush-v0: Discrete(2) — anything else raises, like arm_push_env.py:267
        a = np.round(a)
    o, r, te, tr, _ = vec.step(a)
    o1, r1, te1, tr1, _ = one._vec.step(a[:1])
    torch.cuda.synchronize()
    assert o.shape == (n, vec.obs_dim) and r.shape == (n,) and np.isfinite(o).all()
    np.testing.assert_array_equal(np.asarray(o[0]), np.asarray(o1[0]))
    assert float(r[0]) == float(r1[0]) and bool(te[0]) == bool(te1[0]) and bool(tr[0]) == bool(tr1[0])
    vec.close()
    one.close()
