"""CPU stand-in for bench.py's device layer — TEST INFRASTRUCTURE, imported by bench.py only when
SOFTROD_BENCH_TEST_SHIM names it (tests/test_bench_launch.py).  It builds the local vec env on
the oracle-backed test double (tests/oracle_backend.py) with a handful of tiny rods, so that the
argument handling, the self-launch (`python bench.py --gpus 2` -> child torchrun), the sharding
and the rank-0 JSON relay of bench.py run on a box without a GPU, over gloo.  Lines produced this
way carry "data": "TEST-SHIM"; they are not measurements."""
import gym_softrobot_amd as gsa
from gym_softrobot_amd import _capi
from tests.oracle_backend import OracleBackend


def make_vec(env_id, n_local, **extra):
    assert env_id == "SoftPendulum-v0", "the shim covers the headline workload only"
    kw = dict(time_step=1e-4, recording_fps=2000, n_elems=8)      # 5 substeps per env.step
    kw.update(extra)
    autoreset = kw.pop("autoreset", False)
    cfg = _capi.softpendulum_config(n_local, **kw)
    return gsa.VecSoftPendulumEnv(n_local, backend=OracleBackend(cfg), autoreset=autoreset, **kw)
