"""The Gymnasium-present path, exercised without Gymnasium (VERDICT r4 "next" #6).

`gym_softrobot_amd` registers its ids into `gymnasium.registry`, derives its envs from `gymnasium.Env`
and uses `gymnasium.spaces` WHEN Gymnasium is importable — branches that never ran in the build image.
tests/fake_gymnasium.py stands in for it (installed into sys.modules of a fresh interpreter BEFORE the
package is imported).  CPU: the eight ids of gym_softrobot/__init__.py:6-15,27-30,37-46,60-63,74-76 appear under
`gym_softrobot_amd/`, with the reference's kwargs, pointing at the HIP env classes, which are
`gymnasium.Env`s with `gymnasium.spaces`.  GPU: `gymnasium.make(...)` through that registry returns the
HIP env, and the single-env facade passes the reference's own API assertions
(/root/reference/tests/envs/test_envs.py:27-47) and determinism test (test_determinism.py:7-60)."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]
IDS = ["OctoArmPullWeight-v0", "OctoArmPush-v0", "OctoArmPush-v1", "OctoArmSingle-v0", "OctoArmTwo-v0", "OctoCrawl-v0", "OctoFlat-v0",
       "OctoFlatLite-v0", "OctoReach-v0", "SoftArmTracking-v0", "SoftPendulum-v0", "SoftPendulum3D-v0"]
MUSCLE_IDS = ("OctoArmPush", "OctoArmPullWeight", "OctoArmTwo", "OctoCrawl", "OctoReach")       # registered with the COOMM caveat

PRELUDE = f"""
import json, sys
sys.path.insert(0, {str(ROOT)!r}); sys.path.insert(0, {str(ROOT / 'tests')!r})
import fake_gymnasium
gymnasium = fake_gymnasium.install()
import gym_softrobot_amd as gsa
"""


def _run(body, timeout=600):
    out = subprocess.run([sys.executable, "-c", PRELUDE + body], capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    return json.loads(out.stdout.strip().splitlines()[-1])


def test_ids_are_registered_into_gymnasium_with_the_reference_kwargs():
    res = _run("""
from gym_softrobot_amd import registration, spaces
from gym_softrobot_amd.envs import base
reg = gymnasium.registry
ours = sorted(k for k in reg if k.startswith("gym_softrobot_amd/"))
out = {"ids": ours, "have": spaces.HAVE_GYMNASIUM,
       "box_is_gymnasium": spaces.Box is gymnasium.spaces.Box, "dict_is_gymnasium": spaces.Dict is gymnasium.spaces.Dict,
       "base_is_gymnasium_env": base.GymEnv is gymnasium.Env,
       "entry": {k: reg[k].entry_point.__name__ for k in ours},
       "kwargs": {k: reg[k].kwargs for k in ours},
       "max_steps": [reg[k].max_episode_steps for k in ours],
       "vec": {k: reg[k].vector_entry_point.__name__ for k in ours},
       "subclass": all(issubclass(reg[k].entry_point, gymnasium.Env) for k in ours),
       "own_registry": gsa.registered()}
# registering the package twice (a re-import under another name, importlib.reload) must not raise
import importlib
importlib.reload(gsa)
out["after_reload"] = sorted(k for k in gymnasium.registry if k.startswith("gym_softrobot_amd/"))
print(json.dumps(out))
""")
    assert res["ids"] == [f"gym_softrobot_amd/{i}" for i in IDS] == res["after_reload"]
    assert res["own_registry"] == IDS
    assert res["have"] and res["box_is_gymnasium"] and res["dict_is_gymnasium"] and res["base_is_gymnasium_env"] and res["subclass"]
    assert res["entry"] == {
        "gym_softrobot_amd/OctoArmPullWeight-v0": "ArmPullWeightEnv",
        "gym_softrobot_amd/OctoArmPush-v0": "ArmPushEnv", "gym_softrobot_amd/OctoArmPush-v1": "ArmPushEnv",
        "gym_softrobot_amd/OctoArmSingle-v0": "ArmSingleEnv", "gym_softrobot_amd/OctoFlat-v0": "FlatEnv",
        "gym_softrobot_amd/OctoArmTwo-v0": "ArmTwoEnv", "gym_softrobot_amd/OctoCrawl-v0": "CrawlEnv", "gym_softrobot_amd/OctoReach-v0": "ReachEnv",
        "gym_softrobot_amd/OctoFlatLite-v0": "FlatEnv", "gym_softrobot_amd/SoftArmTracking-v0": "SoftArmTrackingEnv",
        "gym_softrobot_amd/SoftPendulum-v0": "SoftPendulumEnv", "gym_softrobot_amd/SoftPendulum3D-v0": "SoftPendulum3DEnv"}
    # gym_softrobot/__init__.py:11-15: OctoFlatLite = FlatEnv(n_arm=1, n_action=8); the others carry no kwargs
    assert res["kwargs"]["gym_softrobot_amd/OctoFlatLite-v0"] == {"n_arm": 1, "n_action": 8}
    assert res["kwargs"]["gym_softrobot_amd/OctoArmPush-v1"] == {"mode": "continuous"}     # gym_softrobot/__init__.py:42-46
    assert res["kwargs"]["gym_softrobot_amd/OctoArmPullWeight-v0"] == {"mode": "continuous"}   # gym_softrobot/__init__.py:48-52
    assert all(v == {} for k, v in res["kwargs"].items() if "Lite" not in k and "Push-v1" not in k and "PullWeight" not in k)
    assert res["max_steps"] == [None] * 12          # no TimeLimit wrapper: truncation is the env's own (soft_pendulum.py:226-229)
    # Gymnasium 1.0's make_vec(..., vectorization_mode="vector_entry_point") gets the batched HIP classes
    assert res["vec"] == {
        "gym_softrobot_amd/OctoArmPullWeight-v0": "VecArmPullWeightEnv",
        "gym_softrobot_amd/OctoArmPush-v0": "VecArmPushEnv", "gym_softrobot_amd/OctoArmPush-v1": "VecArmPushEnv",
        "gym_softrobot_amd/OctoArmSingle-v0": "VecArmSingleEnv", "gym_softrobot_amd/OctoFlat-v0": "VecOctoFlatEnv",
        "gym_softrobot_amd/OctoArmTwo-v0": "VecArmTwoEnv", "gym_softrobot_amd/OctoCrawl-v0": "VecCrawlEnv", "gym_softrobot_amd/OctoReach-v0": "VecReachEnv",
        "gym_softrobot_amd/OctoFlatLite-v0": "VecOctoFlatEnv", "gym_softrobot_amd/SoftArmTracking-v0": "VecSoftArmTrackingEnv",
        "gym_softrobot_amd/SoftPendulum-v0": "VecSoftPendulumEnv", "gym_softrobot_amd/SoftPendulum3D-v0": "VecSoftPendulum3DEnv"}


def test_muscle_envs_are_registered_with_the_unpinned_label():
    """the COOMM muscle envs must not be registered without the caveat."""
    import gym_softrobot_amd as gsa

    for i in IDS:
        label = gsa.parity_label(i)
        assert (label is not None and "parity-unpinned" in label and "COOMM" in label) == i.startswith(MUSCLE_IDS), i
    assert "parity-unpinned" in gsa.VecArmPushEnv.parity_label and "parity-unpinned" in gsa.ArmPushEnv.parity_label


def test_without_gymnasium_the_package_registry_alone_serves_make():
    out = subprocess.run([sys.executable, "-c", f"""
import sys
sys.path.insert(0, {str(ROOT)!r})
import gym_softrobot_amd as gsa
from gym_softrobot_amd import spaces, registration
assert not spaces.HAVE_GYMNASIUM and registration._gymnasium is None and "gymnasium" not in sys.modules
assert gsa.registered() == {IDS!r}
try:
    gsa.make("NoSuchEnv-v0")
except KeyError as exc:
    assert "registered" in str(exc)
else:
    raise SystemExit("unknown id accepted")
print("OK")
"""], capture_output=True, text=True, timeout=300, cwd=ROOT)
    assert out.returncode == 0 and "OK" in out.stdout, out.stderr[-2000:]


@pytest.mark.gpu
@pytest.mark.parametrize("env_id", IDS)
def test_gymnasium_make_returns_the_hip_env_and_passes_the_reference_api_assertions(hip_lib, env_id):
    res = _run(f"""
import numpy as np
from gymnasium.spaces import Box
env = gymnasium.make("gym_softrobot_amd/{env_id}")
assert isinstance(env, gymnasium.Env) and env.spec.id == "gym_softrobot_amd/{env_id}"
assert type(env.unwrapped._vec.backend).__name__ == "HipRodBackend"          # the HIP path, nothing else
tier = env.unwrapped._vec.backend.kernel_tier()
ob_space, act_space = env.observation_space, env.action_space
assert isinstance(act_space, gymnasium.spaces.Space) and isinstance(ob_space, gymnasium.spaces.Space)
# /root/reference/tests/envs/test_envs.py:27-47
ob, info = env.reset()
assert isinstance(info, dict)
assert ob_space.contains(ob), f"Reset observation: {{ob!r}} not in space"
if isinstance(ob_space, Box):
    assert ob.dtype == ob_space.dtype
a = act_space.sample()
observation, reward, terminated, truncated, _info = env.step(a)
assert ob_space.contains(observation), f"Step observation: {{observation!r}} not in space"
assert np.isscalar(reward), f"{{reward}} is not a scalar"
assert isinstance(terminated, bool) and isinstance(truncated, bool) and isinstance(_info, dict)
if isinstance(ob_space, Box):
    assert observation.dtype == ob_space.dtype
env.close()
# /root/reference/tests/envs/test_determinism.py:7-60 through gymnasium.make
def flat(x):
    return np.concatenate([np.ravel(x[k]) for k in sorted(x)]) if isinstance(x, dict) else np.ravel(x)
runs = []
for _ in range(2):
    env = gymnasium.make("gym_softrobot_amd/{env_id}")
    initial, _ = env.reset(seed=0)
    env.action_space.seed(0)
    acts = [env.action_space.sample() for _ in range(3)]
    resp = [env.step(a) for a in acts]
    env.close()
    runs.append((initial, acts, resp))
(i1, a1, r1), (i2, a2, r2) = runs
assert np.array_equal(flat(i1), flat(i2))
for x, y in zip(a1, a2):
    assert np.array_equal(flat(x), flat(y))
for (o1, w1, t1, u1, _), (o2, w2, t2, u2, _) in zip(r1, r2):
    assert np.array_equal(flat(o1), flat(o2), equal_nan=True) and w1 == w2 and t1 == t2 and u1 == u2
# the same seed through the package's own make() gives the same numbers: one env, two front doors
env = gsa.make("{env_id}")
j, _ = env.reset(seed=0)
env.close()
assert np.array_equal(flat(j), flat(i1))
# Gymnasium 1.0's vector front door: make_vec -> the batched HIP env; env 0 of the batch is the single env
vec = gymnasium.make_vec("gym_softrobot_amd/{env_id}", num_envs=3, vectorization_mode="vector_entry_point", numpy_output=True)
assert vec.num_envs == 3 and type(vec.backend).__name__ == "HipRodBackend"
vo, _ = vec.reset(seed=0)
assert np.array_equal(np.asarray(vo)[0], flat(i1).astype(np.float32)) or "{env_id}".startswith("OctoFlat")
acts = np.stack([vec.single_action_space.sample() for _ in range(3)])
o, r, te, tr, info = vec.step(acts)
assert np.asarray(o).shape[0] == 3 and np.asarray(r).shape == (3,) and isinstance(info, dict)
vec.close()
print(json.dumps({{"tier": tier}}))
""")
    assert "kernel" in res["tier"]
