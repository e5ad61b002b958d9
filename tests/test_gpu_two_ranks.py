"""BASELINE configs[3] / configs[4] in their stated form — a batch SHARDED over ranks — on the real HIP
backend (VERDICT r2 "weak" #3, "next" #1).  The pool's box has ONE GPU and RCCL refuses two ranks on
one device, so the two ranks share cuda:0 and the packed rows travel over gloo
(SOFTROD_BENCH_ALL_RANKS_ON_DEVICE0=1, SOFTROD_BENCH_DIST_BACKEND=gloo): self-launch, sharding, the
kernel-packed rows, the overlapped all-gather buffers, device-side auto-reset across ranks and the
rank-0 JSON relay all run on real kernels with two real processes.  What stays for an 8-GPU node:
RCCL's own transport between DIFFERENT devices (xGMI) — tests/test_gpu_rccl.py runs RCCL in a world
of one."""
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
TWO_ON_ONE = {"SOFTROD_BENCH_ALL_RANKS_ON_DEVICE0": "1", "SOFTROD_BENCH_DIST_BACKEND": "gloo"}


def _bench(extra_env, *args, timeout=900):
    """`python bench.py ...` exactly as the driver calls it: no torchrun around it, no WORLD_SIZE."""
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--no-cpu-baseline", *args], env=env,
                         capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    line = json.loads(lines[0])
    assert "p2p_trial" not in line            # printed before the experimental transport runs
    tag = "bench.py: p2p_trial: "
    trials = [json.loads(l[len(tag):]) for l in out.stderr.splitlines() if l.startswith(tag)]
    if trials:
        line["_p2p_trial"] = trials[-1]
    line["_stderr"] = out.stderr
    return line


def _check_pair(two, one, total, steps):
    for d, n in ((two, 2), (one, 1)):
        assert d["n_gpus"] == n and d["steps"] == steps and d["value"] > 0 and d["data"] == "synthetic"
        assert d["config"]["envs_total"] == total
    assert two["scaling"] == "weak" and "all_gather" in two["config"]["sharding"]
    # same envs, same actions, same number of steps: the rows of BOTH ranks reached rank 0 and hold
    # exactly what one process stepping the whole batch returns
    assert two["config"]["last_step_checksum"] == one["config"]["last_step_checksum"]
    assert two["config"]["non_finite_envs_at_end"] == one["config"]["non_finite_envs_at_end"]


def test_bench_two_ranks_softpendulum(hip_lib):
    """configs[3]'s shape at world 2: 2 x 2048 envs against 1 x 4096."""
    a = ("--steps", "12", "--warmup", "2")
    two = _bench(TWO_ON_ONE, "--gpus", "2", "--envs-per-gpu", "2048", "--p2p-trial", *a)
    one = _bench({}, "--gpus", "1", "--envs-per-gpu", "4096", *a)
    _check_pair(two, one, 4096, 12)
    assert two["windows"]["count"] == one["windows"]["count"] == 5
    assert one["roofline"]["frac"] is None or 0.2 < one["roofline"]["frac"] < 1.0
    # what every rank measured, and the second measurement over transport p2p (each rank's child process)
    ranks = two["per_rank"]["ranks"]
    assert [r["rank"] for r in ranks] == [0, 1] and all(0 < r["efficiency_vs_n1_kernel"] <= 1.0 for r in ranks)
    t = two["_p2p_trial"]
    assert t["returncode"] == 0 and t["transport"] == "p2p" and t["exchange_memory"] in ("uncached", "fine-grained"), t
    assert t["last_step_checksum"] == two["config"]["last_step_checksum"] and t["value"] > 0
    # configs[2], configs[4]'s share, the libm kernel, and (round 6) the muscle arm WITH its parity label
    assert "secondary" not in two and "secondary" in one and len(one["secondary"]) == 4
    assert "parity-unpinned" in one["secondary"][3]["parity_label"] and "OctoArmPush-v1" in one["secondary"][3]["workload"]
    for sec in one["secondary"]:
        assert sec["value"] > 0 and sec["non_finite_envs_at_end"] == 0 and sec["kernel_ms_avg"] > 0
    assert one["secondary"][2]["math_mode"] == "libm" and one["secondary"][2]["value"] < one["value"]
    assert one["policy_in_loop"]["value"] > 0 and one["sustained"]["seconds"] >= 2.0
    assert 0.85 < one["sustained"]["ratio_to_value"] < 1.1, one["sustained"]
    assert one["pcie_inclusive"]["value"] > 0 and one["pcie_inclusive"]["ms_per_step"] > one["roofline"]["kernel_ms_avg"]


def test_bench_two_ranks_default_sizes_keep_every_gpu_busy(hip_lib):
    """`python bench.py --gpus 2 --steps K --warmup W` at the DEFAULT batch (2 x 4096 envs: the driver's command):
    after the windows every rank runs the >= 2 s `sustained` leg on its own GPU with no exchange (what the driver's
    GPU-busy samples see); rank 0's figure is in the line, every rank's on stderr; no p2p trial unless asked for."""
    two = _bench(TWO_ON_ONE, "--gpus", "2", "--steps", "10", "--warmup", "2")
    assert two["n_gpus"] == 2 and two["config"]["envs_total"] == 8192 and two["config"]["transport"] == "rccl"
    su = two["sustained"]
    assert su["seconds"] >= 2.0 and su["value"] > 0 and "rank 0" in su["note"], su
    tag = "bench.py: sustained rank "
    ranks = sorted(int(l[len(tag):].split(":")[0]) for l in two["_stderr"].splitlines() if l.startswith(tag))
    assert ranks == [0, 1], two["_stderr"][-1500:]
    assert "_p2p_trial" not in two


def test_bench_two_ranks_strong_scaling(hip_lib):
    """SURVEY §8(d) cfg 4 asks for the curve at fixed total N as well: `--scaling strong` splits
    --envs-per-gpu envs over the ranks (2 x 1024 here) and must return what one rank returns for
    the same 2048 envs."""
    a = ("--steps", "10", "--warmup", "2", "--envs-per-gpu", "2048", "--preheat", "30")
    two = _bench(TWO_ON_ONE, "--gpus", "2", "--scaling", "strong", "--no-p2p-trial", *a)
    one = _bench({}, "--gpus", "1", *a)
    assert two["scaling"] == "strong" and two["n_gpus"] == 2 and two["config"]["envs_total"] == 2048
    assert two["config"]["last_step_checksum"] == one["config"]["last_step_checksum"]
    assert two["config"]["non_finite_envs_at_end"] == one["config"]["non_finite_envs_at_end"] == 0


def test_bench_two_ranks_octoflat(hip_lib):
    """configs[4]'s shape at world 2: 2 x 512 OctoFlat envs (8 arms + head each) against 1 x 1024."""
    a = ("--env", "OctoFlat-v0", "--steps", "3", "--warmup", "1", "--windows", "1", "--preheat", "40")
    two = _bench(TWO_ON_ONE, "--gpus", "2", "--envs-per-gpu", "512", "--no-p2p-trial", *a)
    one = _bench({}, "--gpus", "1", "--envs-per-gpu", "1024", *a)
    _check_pair(two, one, 1024, 3)


def test_bench_two_ranks_device_autoreset_across_the_shard_boundary(hip_lib):
    """140 steps: every env is truncated on step 126 and restarts on 127 from its staged record, on
    both ranks; the restarted envs' rows are gathered like any other."""
    a = ("--steps", "140", "--warmup", "2", "--autoreset", "device", "--preheat", "20")
    two = _bench(TWO_ON_ONE, "--gpus", "2", "--envs-per-gpu", "1024", "--no-p2p-trial", *a)
    one = _bench({}, "--gpus", "1", "--envs-per-gpu", "2048", *a)
    _check_pair(two, one, 2048, 140)
    assert two["config"]["autoreset"] == "device"
    assert two["config"]["episode_restarts_not_counted"] == one["config"]["episode_restarts_not_counted"] >= 2048


def _worker(*args, timeout=900):
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), str(ROOT / "tests" / "two_rank_hip_worker.py"), *args]
    out = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-3000:])
    assert "TWO-RANK-OK" in out.stdout, out.stdout[-2000:]
    return out.stdout


@pytest.mark.parametrize("env_id,per,t1,t2", [("SoftPendulum-v0", 96, 5, 6), ("SoftPendulum3D-v0", 32, 3, 3),
                                               ("OctoArmSingle-v0", 16, 2, 2), ("OctoFlat-v0", 6, 2, 2)])
def test_sharded_env_two_hip_ranks_with_a_masked_reset_mid_rollout(hip_lib, env_id, per, t1, t2):
    """ShardedVecEnv(overlap=True) over two HIP ranks, a masked reset between two stretches of steps
    (envs on both sides of the shard boundary), every gathered step bit-equal to one process."""
    _worker(env_id, str(per), str(t1), str(t2))


@pytest.mark.parametrize("env_id,per,t1,t2", [("SoftPendulum-v0", 96, 5, 6), ("OctoFlat-v0", 6, 2, 2)])
def test_sharded_env_two_hip_ranks_p2p_transport(hip_lib, env_id, per, t1, t2):
    """transport="p2p": no collective per step — every rank copies its packed rows into its block of
    every peer's buffer (IPC-mapped device memory; here both ranks sit on cuda:0, so the mapping is
    exercised and the copies are local), a barrier at sync().  Bit-equal to one process on every
    gathered step, masked reset included."""
    _worker(env_id, str(per), str(t1), str(t2), "off", "p2p")


def test_bench_two_ranks_p2p_transport(hip_lib):
    a = ("--steps", "12", "--warmup", "2", "--preheat", "30")
    two = _bench(TWO_ON_ONE, "--gpus", "2", "--envs-per-gpu", "1024", "--transport", "p2p", *a)
    one = _bench({}, "--gpus", "1", "--envs-per-gpu", "2048", *a)
    assert two["config"]["transport"] == "p2p" and "peer copies" in two["config"]["sharding"]
    assert two["config"]["last_step_checksum"] == one["config"]["last_step_checksum"]


def test_sharded_env_two_hip_ranks_device_autoreset(hip_lib):
    """Device-side NEXT_STEP auto-reset on both ranks (3-step episodes, so every env restarts several
    times) against ONE process with host-driven auto-reset: the same draws, the same rows."""
    out = _worker("SoftPendulum-v0", "64", "7", "8", "device")
    flagged = int(out.split("flagged=")[1].split()[0])
    assert flagged >= 128 * 3


def test_exchange_buffers_are_uncached_device_memory(hip_lib):
    """transport="p2p" never hands a peer an ordinary (L2-cached) allocation to store into:
    softrod_exchange_alloc returns uncached (or fine-grained) device memory, zeroed, with an IPC handle."""
    import torch

    import gym_softrobot_amd as gsa

    env = gsa.make_vec("SoftPendulum-v0", 4, device=0)
    be = env.backend
    t, ptr, handle, kind = be.exchange_alloc(1024)
    assert kind in ("uncached", "fine-grained") and len(handle) == 64 and any(handle) and ptr == t.data_ptr()
    assert t.shape == (1024,) and float(t.abs().sum().item()) == 0.0
    t[:] = torch.arange(1024, device=t.device, dtype=torch.float32)
    torch.cuda.synchronize()
    assert float(t.sum().item()) == 1023 * 1024 / 2
    # the tagged scatter into it: rows, then the generation word behind them
    env.reset(seed=0)
    packed = be.step_packed(torch.zeros(4, device=t.device))
    w = packed.shape[1]
    be.scatter_rows(packed, [ptr], 0, tag_word=4 * w + 0, tag=77)
    torch.cuda.synchronize()
    assert torch.equal(t[: 4 * w].view(4, w), packed) and int(t[4 * w : 4 * w + 1].view(torch.int32).item()) == 77
    del t
    be.exchange_free(ptr)
    env.close()


def test_sharded_env_p2p_reports_its_exchange_memory_and_checks_generations(hip_lib):
    """World of one over the p2p transport (gloo group of one rank): the set-up self-test passes, the
    buffers are uncached, sync() verifies the generation word, and a corrupted word is caught."""
    out = subprocess.run([sys.executable, "-c", """
import os, sys
sys.path.insert(0, %r)
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=%r, RANK="0", WORLD_SIZE="1")
import torch, torch.distributed as dist
import gym_softrobot_amd as gsa
from gym_softrobot_amd.distributed import ShardedVecEnv, P2PError
dist.init_process_group("gloo")
local = gsa.make_vec("SoftPendulum-v0", 64, device=0)
env = ShardedVecEnv(local, 64, overlap=True, force_collective=True, transport="p2p")
assert env.transport == "p2p", env._p2p_error
assert env.exchange_memory in ("uncached", "fine-grained")
env.reset(seed=0)
ref = gsa.make_vec("SoftPendulum-v0", 64, device=0)
ref.reset(seed=0)
a = torch.linspace(-20, 20, 64, device="cuda")
for t in range(5):
    o, r, te, tr, _ = env.step(a)
    env.sync()
    o2, r2, _, _, _ = ref.step(a)
    torch.cuda.synchronize()
    assert torch.equal(o, o2) and torch.equal(r, r2)
# the completion contract (ADVICE r4): with depth 2 the rows of every step stay intact only with a sync()
# after every step; a second step without one is refused unless the caller says it reads nothing in between
assert env.p2p_sync_interval() == 1 and env.p2p_sync_interval(every_step=False) == 1
env.step(a)
try:
    env.step(a)
    raise SystemExit("two steps without sync() went through")
except P2PError as exc:
    assert "sync()" in str(exc)
env.sync()
env.p2p_enforce_sync_interval = False
env.step(a); env.step(a); env.step(a)
env.sync()                                  # checks the generation words of BOTH buffers written since
env.p2p_enforce_sync_interval = True
ref.step(a); ref.step(a); ref.step(a); ref.step(a)
o, r, te, tr, _ = env.step(a)
env.sync()
o2, r2, _, _, _ = ref.step(a)
torch.cuda.synchronize()
assert torch.equal(o, o2) and torch.equal(r, r2)
env._tags[env._last_k][0] = 12345          # a peer that did not deliver
try:
    env.sync()
    raise SystemExit("a wrong generation word went unnoticed")
except P2PError:
    pass
env._tags[env._last_k][0] = env._gen_of[env._last_k]
env.close(); ref.close()
dist.destroy_process_group()
print("P2P-GEN-OK")
""" % (str(ROOT), str(_free_port()))], capture_output=True, text=True, timeout=600,
                         env=dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0"))
    assert out.returncode == 0 and "P2P-GEN-OK" in out.stdout, (out.stdout[-1500:], out.stderr[-3000:])


def test_driver_line_carries_cpu_baseline_and_parity_vs_oracle(hip_lib):
    """`python bench.py --steps 20 --warmup 5` as the driver runs it at N = 1: one JSON line with the
    contract's keys, `roofline`, `secondary`, `pcie_inclusive`, and `cpu_baseline` on the 4096 rods
    with BASELINE.md §3's parity figures (HIP vs the oracle after 1 / 3 / 100 env.steps) inside 1e-5."""
    env = dict(os.environ)
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--steps", "20", "--warmup", "5"], env=env,
                         capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-3000:]
    (line,) = [json.loads(l) for l in out.stdout.splitlines() if l.startswith("{")]
    assert line["metric"] == "env_steps_per_sec" and line["n_gpus"] == 1 and line["dtype"] == "f64"
    assert line["methodology_version"] == 5 and line["windows"]["count"] == 5
    assert line["value"] == sorted(line["windows"]["value"])[2] and line["single_window"]["value"] == line["windows"]["value"][0]
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and "4096 rods" in cb["sample"]
    pv = cb["parity_vs_oracle"]
    assert pv["steps"] == [1, 3, 100] and pv["within_tolerance"] and pv["flags_equal"], pv
    assert max(pv["max_rel_obs"]) <= 1e-5 and max(pv["max_rel_reward"]) <= 1e-5
    r = line["roofline"]
    assert r["bound"] == "fp64_valu" and (r["frac"] is None or 0.5 < r["frac"] < 1.05)
    assert r["frac"] is None or (r["frac_cycle_weighted"] is not None and 0.5 < r["frac_cycle_weighted"] <= 1.0)
    assert [s["baseline_config"][:10] for s in line["secondary"]] == ["configs[2]", "configs[4]", "configs[1]", "none: BASE"]
    assert "parity-unpinned" in line["secondary"][3]["parity_label"]          # the COOMM muscle arm never without its caveat
    assert line["secondary"][2]["math_mode"] == "libm"
    su = line["sustained"]
    assert su["seconds"] >= 2.0 and 0.85 < su["ratio_to_value"] < 1.1 and {"near_start", "near_end"} <= set(su["sensors"]), su
    assert line["policy_in_loop"]["value"] > 0 and line["policy_in_loop"]["ms_per_step"] > line["roofline"]["kernel_ms_avg"]
