"""Runs bench.py's own `main()` on a box WITHOUT a GPU — TEST INFRASTRUCTURE (tests/test_bench_launch.py).

bench.py carries no switch that swaps its backend: this launcher lives under tests/,
patches what it needs in ITS process and then calls `bench.main(script=<this file>)`, so that the
self-launch (`--gpus 2` -> child `torch.distributed.run` of this file), the argument handling, the
sharding, the packed all-gather (over gloo) and the rank-0 JSON relay are bench.py's code, while the
rods are stepped by the oracle-backed double of tests/oracle_backend.py (a handful of tiny rods).
The line such a run prints says "data": "TEST-DOUBLE (not a measurement)" — bench.py labels any
backend that is not HipRodBackend that way."""
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main() -> int:
    if "--gpus" in sys.argv and "WORLD_SIZE" not in os.environ and int(sys.argv[sys.argv.index("--gpus") + 1]) > 1:
        import bench                               # the parent of a self-launch: nothing to patch, no torch

        return bench.main(script=__file__)

    import torch

    import bench
    import gym_softrobot_amd as gsa
    from gym_softrobot_amd import _capi
    from tests.oracle_backend import OracleBackend

    def make_vec(env_id, n_local, device=0, math_mode=_capi.MATH_FAST, **extra):
        if env_id == "OctoFlat-v0":            # configs[4]'s shape: 8 arms + head per env, 12 substeps per env.step
            cfg = _capi.octo_flat_config(n_local)
            cfg.n_substeps = 12
            env = gsa.VecOctoFlatEnv(n_local, backend=OracleBackend(cfg), autoreset=extra.pop("autoreset", False))
            env.cfg.n_substeps = 12
            return env
        assert env_id == "SoftPendulum-v0", "the double covers the headline workload and OctoFlat only"
        kw = dict(time_step=1e-4, recording_fps=2000, n_elems=8)      # 5 substeps per env.step
        if os.environ.get("SOFTROD_TEST_EPISODE_STEPS"):              # short episodes: truncation inside the run
            kw["final_time"] = int(os.environ["SOFTROD_TEST_EPISODE_STEPS"]) * 5e-4 - 1e-9
        kw.update(extra)
        autoreset = kw.pop("autoreset", False)
        cfg = _capi.softpendulum_config(n_local, **kw)
        return gsa.VecSoftPendulumEnv(n_local, backend=OracleBackend(cfg), autoreset=autoreset, **kw)

    gsa.make_vec = make_vec
    torch.cuda.is_available = lambda: True
    torch.cuda.set_device = lambda *_a, **_k: None
    torch.cuda.synchronize = lambda *_a, **_k: None
    os.environ["SOFTROD_BENCH_DIST_BACKEND"] = "gloo"
    return bench.main(script=__file__)


if __name__ == "__main__":
    sys.exit(main())
