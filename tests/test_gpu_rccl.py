"""The multi-GPU code path of bench.py on the real RCCL backend.  A 1-GPU box cannot hold two
ranks (RCCL refuses two ranks on one device), so this is a world of ONE, launched the way the
driver launches N ranks (torch.distributed.run), with the packed all-gather forced on: process
group creation with device_id, the asynchronous all_gather_into_tensor on alternating buffers,
the barrier and the max/sum all-reduces all run through librccl.  World size 2 is covered on CPU
over gloo (tests/test_distributed_gloo.py)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    import socket

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


ROOT = Path(__file__).resolve().parents[1]


def _bench(extra_env, *args):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(ROOT / "bench.py"),
           "--gpus", "1", "--no-cpu-baseline", *args]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_through_rccl_in_a_world_of_one(hip_lib):
    with_gather = _bench({"SOFTROD_BENCH_FORCE_DIST": "1"}, "--steps", "12", "--warmup", "2", "--envs-per-gpu", "512")
    plain = _bench({}, "--steps", "12", "--warmup", "2", "--envs-per-gpu", "512")
    for d in (with_gather, plain):
        assert d["n_gpus"] == 1 and d["steps"] == 12 and d["value"] > 0
        assert d["config"]["non_finite_envs_at_end"] == 0
    # same envs, same actions: the gathered run must integrate exactly what the plain one does
    assert with_gather["config"]["envs_total"] == plain["config"]["envs_total"] == 512
    assert with_gather["config"]["last_step_checksum"] == plain["config"]["last_step_checksum"]
