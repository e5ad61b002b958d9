"""bench.py must be launchable exactly as the driver launches it: `python bench.py --gpus N ...`
for every N, with no torchrun around it (ADVICE r1 / VERDICT r1 "missing" #4).  These tests run
that command line on CPU through tests/bench_cpu_launcher.py, which patches an oracle-backed
double over gloo into ITS process and calls bench.main(); everything else — argument handling, the
child torch.distributed.run, sharding, the all-gather, the rank-0 JSON relay, the return code — is
bench.py's own code, and bench.py itself carries no backend switch."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parents[1]


def _run(args, extra_env=None, timeout=300, script="bench_cpu_launcher.py"):
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    env["PYTHONPATH"] = str(ROOT) + os.pathsep + env.get("PYTHONPATH", "")
    env["OMP_NUM_THREADS"] = "1"
    env.update(extra_env or {})
    return subprocess.run([sys.executable, str(ROOT / "tests" / script)] + args, cwd=str(ROOT), env=env,
                          capture_output=True, text=True, timeout=timeout)


def _json_lines(out):
    return [json.loads(l) for l in out.splitlines() if l.startswith("{")]


def _p2p_trial(stderr):
    """The trial's figures: written to stderr AFTER rank 0 printed the line (bench.py)."""
    tag = "bench.py: p2p_trial: "
    hits = [l[len(tag):] for l in stderr.splitlines() if l.startswith(tag)]
    assert len(hits) == 1, stderr[-2000:]
    return json.loads(hits[0])


def test_single_process_line_has_the_contract_fields(oracle_built):
    p = _run(["--gpus", "1", "--steps", "3", "--warmup", "1", "--envs-per-gpu", "4", "--no-cpu-baseline"])
    assert p.returncode == 0, p.stderr[-2000:]
    (line,) = _json_lines(p.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
              "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["n_gpus"] == 1 and line["steps"] == 3 and line["warmup"] == 1
    assert line["data"].startswith("TEST-DOUBLE") and line["config"]["envs_total"] == 4
    assert line["windows"]["count"] == 5 and len(line["windows"]["value"]) == 5      # 1 + 5*3 steps fit an episode
    assert line["value"] == sorted(line["windows"]["value"])[2]                       # the median window
    assert line["roofline"]["frac"] is None and line["roofline"]["frac_withheld"]     # not the HIP library: no pricing
    assert line["roofline"]["bound"] == "fp64_valu" and "frac" in line["roofline"] and "traffic" in line["roofline"]


def test_gpus_2_launches_itself_and_prints_one_line(oracle_built):
    """The driver's multi-GPU command: no WORLD_SIZE in the environment, --gpus 2."""
    p = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--envs-per-gpu", "4"])
    assert p.returncode == 0, p.stderr[-3000:]
    (line,) = _json_lines(p.stdout)
    assert line["n_gpus"] == 2 and line["config"]["envs_total"] == 8 and line["scaling"] == "weak"
    # the gathered rows of both ranks reached rank 0: same last-step checksum as one process stepping all 8
    q = _run(["--gpus", "1", "--steps", "3", "--warmup", "1", "--envs-per-gpu", "8", "--no-cpu-baseline"])
    assert q.returncode == 0, q.stderr[-2000:]
    (one,) = _json_lines(q.stdout)
    assert line["config"]["last_step_checksum"] == one["config"]["last_step_checksum"]


def test_gpus_2_with_the_p2p_transport_falls_back_on_a_box_without_a_gpu(oracle_built):
    """`--transport p2p` where its set-up cannot work (no device memory to map): every rank stays with
    the collective, the line says which transport ran, and the rows are the same."""
    p = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--envs-per-gpu", "4", "--transport", "p2p"])
    assert p.returncode == 0, p.stderr[-3000:]
    (line,) = _json_lines(p.stdout)
    assert line["config"]["transport"] == "rccl" and "all_gather" in line["config"]["sharding"]
    q = _run(["--gpus", "2", "--steps", "3", "--warmup", "1", "--envs-per-gpu", "4"])
    (ref,) = _json_lines(q.stdout)
    assert line["config"]["last_step_checksum"] == ref["config"]["last_step_checksum"]


# ---- the driver's 8-GPU command lines, rehearsed on CPU: 8 gloo ranks, a handful of tiny rods each ----

def _eight_vs_one(extra_args, per_rank_envs, extra_env=None, strong=False):
    a = ["--steps", "4", "--warmup", "1", "--windows", "1"] + extra_args
    total = per_rank_envs * 8
    p = _run(["--gpus", "8", "--envs-per-gpu", str(total if strong else per_rank_envs)] + a, extra_env, timeout=600)
    assert p.returncode == 0, p.stderr[-3000:]
    (line,) = _json_lines(p.stdout)
    q = _run(["--gpus", "1", "--envs-per-gpu", str(total), "--no-cpu-baseline"] + a,
             extra_env, timeout=600)
    assert q.returncode == 0, q.stderr[-2000:]
    (one,) = _json_lines(q.stdout)
    line["_stderr"] = p.stderr
    assert line["n_gpus"] == 8 and line["config"]["envs_total"] == total == one["config"]["envs_total"]
    # the rows of all 8 ranks reached rank 0 and are what one process stepping the whole batch returns
    assert line["config"]["last_step_checksum"] == one["config"]["last_step_checksum"]
    assert line["config"]["non_finite_envs_at_end"] == one["config"]["non_finite_envs_at_end"] == 0
    # what every rank measured is in the line
    ranks = line["per_rank"]["ranks"]
    assert [r["rank"] for r in ranks] == list(range(8))
    for r in ranks:
        assert r["kernel_ms_avg"] > 0 and 0 < r["efficiency_vs_n1_kernel"] <= 1.0 + 1e-9
        assert abs(r["exchange_us_per_step"] - (line["ms_per_step"] - r["kernel_ms_avg"]) * 1e3) < 1e-6
    return line, one


def test_gpus_8_softpendulum_weak_scaling_with_the_p2p_trial(oracle_built):
    """BASELINE configs[3]'s shape: `python bench.py --gpus 8 ...` as the driver runs it, plus `--p2p-trial`
    (opt-in since round 5).  The default transport is measured, then every rank's child runs the job again over transport p2p —
    which on a box without a GPU falls back to the collective on every rank, says so, and returns the
    same rows.  The trial runs AFTER the line is out, bounded, in children that die with their parents."""
    line, one = _eight_vs_one(["--p2p-trial"], 2)
    assert line["scaling"] == "weak" and line["config"]["transport"] == "rccl"
    assert "p2p_trial" not in line            # the headline is printed BEFORE the experimental transport runs
    t = _p2p_trial(line["_stderr"])
    assert t["returncode"] == 0 and t["transport"] == "rccl" and t["transport_fallback_reason"]
    assert t["last_step_checksum"] == line["config"]["last_step_checksum"]
    assert len(t["per_rank"]["ranks"]) == 8


def test_gpus_8_strong_scaling(oracle_built):
    line, _ = _eight_vs_one(["--scaling", "strong"], 2, strong=True)
    assert line["scaling"] == "strong" and "bench.py: p2p_trial:" not in line["_stderr"]


def test_gpus_8_octoflat(oracle_built):
    """BASELINE configs[4]'s shape: 8 arms + head per env, odd observation width (461) in the packed rows."""
    line, _ = _eight_vs_one(["--env", "OctoFlat-v0"], 1)
    assert "OctoFlat-v0" in line["config"]["workload"]


def test_gpus_8_device_autoreset_restarts_on_every_rank(oracle_built):
    """3-step episodes: every env is truncated and restarts from its staged record inside the run, on
    all 8 ranks; restarted steps are not counted as work, and the rows equal one process's."""
    line, one = _eight_vs_one(["--autoreset", "device"], 2, {"SOFTROD_TEST_EPISODE_STEPS": "3"})
    assert line["config"]["autoreset"] == "device"
    assert line["config"]["episode_restarts_not_counted"] == one["config"]["episode_restarts_not_counted"] >= 16


def test_child_failure_propagates(oracle_built):
    # 3 envs do not split over 2 ranks: every rank exits non-zero before any line is printed
    p = _run(["--gpus", "2", "--steps", "2", "--warmup", "1", "--envs-per-gpu", "3", "--scaling", "strong",
              "--env", "SoftPendulum-v0"])
    assert p.returncode != 0
    assert not _json_lines(p.stdout)


def test_bench_py_has_no_backend_switch_and_refuses_a_box_without_a_gpu():
    """bench.py on its own: no env var swaps its device layer, and without a GPU it
    stops instead of falling back."""
    text = (ROOT / "bench.py").read_text()
    assert "SHIM" not in text and "importlib" not in text
    import torch

    if torch.cuda.is_available():
        pytest.skip("a GPU is visible")
    p = _run(["--gpus", "1", "--steps", "2", "--warmup", "1", "--envs-per-gpu", "4"], script="../bench.py")
    assert p.returncode != 0 and "no CPU fallback" in (p.stderr + p.stdout)
    assert not _json_lines(p.stdout)


def test_a_hung_child_is_killed_after_the_launch_timeout(tmp_path):
    """self_launch must not block forever on a hung child: the ranks run in their own
    process group and are killed as a group at SOFTROD_BENCH_LAUNCH_TIMEOUT."""
    sys.path.insert(0, str(ROOT))
    import bench

    hang = tmp_path / "hang.py"
    hang.write_text("import time\ntime.sleep(600)\n")
    args = bench.parse_args(["--gpus", "2"])
    os.environ["SOFTROD_BENCH_LAUNCH_TIMEOUT"] = "8"
    try:
        import time

        t0 = time.time()
        rc = bench.self_launch(args, script=hang, argv=["--gpus", "2"])
        assert rc == 124 and time.time() - t0 < 60
    finally:
        del os.environ["SOFTROD_BENCH_LAUNCH_TIMEOUT"]


def test_mismatched_world_is_refused():
    p = _run(["--gpus", "3"], extra_env={"WORLD_SIZE": "2", "RANK": "0"})
    assert p.returncode != 0 and "WORLD_SIZE" in (p.stderr + p.stdout)


def test_usable_cpus_reads_the_cgroup_quota(monkeypatch, tmp_path):
    sys.path.insert(0, str(ROOT))
    import bench

    n = bench.usable_cpus()
    assert 1 <= n <= len(os.sched_getaffinity(0))


def test_preheat_waits_out_the_ramp_and_the_pull_backs():
    """bench.preheat on a scripted clock: a 45 ms ramp, pull-backs 55 and 110 ms into the load (what
    the pool's boxes do, profiles/README.md r3f), then a steady 0.303 ms.  It must not stop on the
    first plateau (the floor of 120 ms of kernel time), must stop once three groups agree to 0.5 %,
    and must give up at the cap when the clock never settles."""
    sys.path.insert(0, str(ROOT))
    import numpy as np
    import torch

    import bench

    class FakeBackend:
        def __init__(self, law):
            self.law, self.busy, self.n, self.pending = law, 0.0, 0, []

        def set_timing(self, n):
            self.pending = []

        def launch(self):
            ms = self.law(self.busy)
            self.busy += ms
            self.pending.append(ms)

        def kernel_times_ms(self):
            out, self.pending = np.asarray(self.pending, np.float32), []
            return out

    class FakeEnv:
        def __init__(self, law):
            self.backend = FakeBackend(law)

        def reset(self, seed=None):
            pass

        def step(self, a):
            self.backend.launch()

    def pool_box(busy_ms):
        if busy_ms < 45:
            return 0.38 - 0.08 * busy_ms / 45
        for start in (55.0, 110.0):
            if start <= busy_ms < start + 20:
                return 0.30 + 0.06 * (1 - (busy_ms - start) / 20)
        return 0.303

    acts = torch.zeros((4, 8, 1))
    env, info = bench.preheat(lambda: FakeEnv(pool_box), acts, cap_ms=600.0)
    assert info["stable"] and 130.0 <= info["kernel_ms_total"] <= 200.0, info
    assert abs(info["settled_kernel_ms"] - 0.303) < 1e-3
    # a clock that never settles: the cap ends it
    env, info = bench.preheat(lambda: FakeEnv(lambda b: 0.30 + 0.05 * ((int(b / 6) % 2))), acts, cap_ms=300.0)
    assert not info["stable"] and 300.0 <= info["kernel_ms_total"] < 320.0
    assert bench.preheat(lambda: FakeEnv(pool_box), acts, cap_ms=0.0) == (None, {"launches": 0})
