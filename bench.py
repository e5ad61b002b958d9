#!/usr/bin/env python3
"""bench.py — env-steps/s of N parallel SoftPendulum-v0 on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one env.step() of every env of the job: 4096 envs x 50 elements per GPU
(BASELINE.json configs[1]; with N GPUs the batch is 4096*N envs sharded contiguously,
configs[3] at N=8 — weak scaling), i.e. per GPU 4096 x 400 PositionVerlet substeps in
one kernel launch, plus (N>1) one packed RCCL all-gather of the per-env outputs.
Inputs (state, pre-staged float32 actions) are resident in HBM when the timed region
starts.  Rank 0 prints ONE JSON line.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment launches itself:
the parent starts `python -m torch.distributed.run --nproc-per-node N bench.py ...` as a CHILD
process before anything in the parent touches the GPU (never an exec), relays rank 0's JSON
line and exits with the child's return code.  Launched under torchrun (WORLD_SIZE set) it
runs as one rank, as before.

roofline: the kernel is register-resident (HBM is touched once per env.step, not once per
substep), so what bounds it is the fp64 VALU issue rate: a wave64 fp64 instruction occupies
its SIMD for 4 cycles (1024 SIMDs x 2.4 GHz / 4 = 614.4 G wave-instructions/s = 78.6 TFLOP/s
of FMAs).  `roofline.achieved` = VALU wave-instructions per launch (SQ_INSTS_VALU of this
workload's kernel from the tracked PMC pass, profiles/valu_counts.json, which names the
rocprofv3 output it came from) / the step kernel's average duration measured live with HIP
events on the launch stream (softrod_set_timing / softrod_kernel_times_ms); `frac` =
achieved / peak.  `roofline.traffic` = measured HBM bytes per launch (FETCH_SIZE/WRITE_SIZE
passes, profiles/hbm_traffic.json); `roofline.hbm` carries the measured-traffic bandwidth
and, labelled as a model, SURVEY.md §8(d)'s algorithmic-bytes streaming figure (rods x
substeps x 2*(18n+6)*8 B / kernel time), which a fused kernel exceeds by construction.

Clock: after an idle period the shader clock ramps for ~45 ms under load (0.38 -> 0.297 ms per
launch over 140 launches) and is then pulled back twice, ~55 and ~110 ms after the load began (a
power-controller transient: 0.30 -> 0.35-0.37 ms, gone 20 ms later), before it holds (0.303 ms).  Before the declared warm-up an un-timed PRE-HEAT
steps a scratch batch of the same shape with zero actions until the HIP-event kernel time is
stable (groups of ~6 ms of kernel time, three groups within 0.5 %, after 120 and within 600 ms of
kernel time); the measured batch is untouched by it, so the timed window stays inside one episode.  `steps`/`warmup` echo the
arguments; as many consecutive windows of K steps as fit one episode together with the warm-up
(120 steps; at most five) are timed, each bracketed like the contract says, and `value` /
`ms_per_step` are the MEDIAN window's (all windows are listed under `windows`): on some boxes the
power controller pulls the clock back for ~20 ms every ~55 ms, and one 6 ms window can fall
entirely inside or outside such a dip.

Staleness: profiles/valu_counts.json and hbm_traffic.json entries carry the source hash of the
library they were measured on (softrod_source_hash); when it differs from the loaded library's,
`roofline.frac` / `traffic` are null and `frac_withheld` says why.

cpu_baseline: the repo's fp64 C oracle (a port/restatement, NOT PyElastica — see
oracle/softrod_oracle.c) timed on this box's host cores with OpenMP over rods, rank 0,
N=1 only, on a bounded sample of the same workload: the SAME 4096 rods (SURVEY §8(d) /
BASELINE.md §3), as many env.steps as fit ~10 s (at least 2), plus one rod on one thread for
1 s.  `cores` = the CPUs the process may really use (cgroup quota, else affinity), which is
also the OpenMP thread count.  `cpu_baseline.parity_vs_oracle`: max |d obs| / |obs| of
the HIP path against that oracle after 1 / 3 / 100 env.steps on the first 16 envs of the batch.

secondary (N=1, default): after the headline windows the same process measures BASELINE
configs[2] (OctoArmSingle-v0, 100 elements x 4096 envs) and configs[4]'s per-GPU share
(OctoFlat-v0 x 1024 envs), 10 timed steps each on the warm clock, and reports them under
`secondary` with their own roofline blocks (hash-matched tables); the headline keys are untouched.
`pcie_inclusive` (same condition): the headline workload driven from pinned HOST buffers in a closed
loop — what a caller that hands over NumPy arrays gets; never `value`.

N > 1: `per_rank` lists every rank's step-kernel time, what a step costs beyond it (the exchange:
collective or peer copies, plus launch gaps) and `efficiency_vs_n1_kernel` = kernel time / step
time.  `--p2p-trial` (opt-in since round 5: the transport has never run between different devices, and
the driver's N = 1, 2, 4, 8 runs share one node): AFTER rank 0 has printed the line every rank runs the same
windows once more over transport "p2p" in a child process of its own, bounded by 90 s (a crash, a hang
or a fall-back of the experimental transport cannot cost the headline measurement: it is out already).
The trial's figures go to stderr as `bench.py: p2p_trial: {...}` and to gpurun_out/p2p_trial_N<world>.json.

sustained (default; round 5; N > 1: every rank on its own GPU, no exchange, rank 0's figure in the line):
after everything else ONE more leg keeps the GPU under continuous load
for >= 8 s — ~27 000 env.steps of the headline workload with device-side NEXT_STEP auto-reset (episodes
end every 125 steps and restart from staged draws without a host round trip) — so that an outside
observer sampling the GPU (the driver's smi samples) sees it busy; its env-steps/s is reported beside,
never instead of, the median-window `value`, together with the shader clock and socket power read
from sysfs (or rocm-smi) near the start and the end of the leg.  `policy_in_loop` (same condition): the
closed RL loop on the device — observations -> a 2 x 64 tanh MLP (torch, the same stream, no host
synchronisation) -> actions -> softrod_step, with device auto-reset — i.e. what examples/
soft_pendulum_3d/train_ppo.py's rollout costs per env.step when the policy lives next to the envs.

methodology_version 5 (round 5: `secondary` gains the libm-mode SoftPendulum entry, `sustained`,
`policy_in_loop`; the headline windows are those of version 4).  3 = round 3: clock pre-heat + median of up to five windows;
figures of rounds 1-2 (one window, no pre-heat) are not like for like.  `single_window` in the
line is window 0 alone, whatever the window count.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

ENVS_PER_GPU = 4096
N_ELEM = 50
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: HBM3E 8 TB/s
N_SIMD = 1024                # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9             # MI355X_MICROARCH.md peak engine clock
CYCLES_PER_FP64_WAVE_INSTR = 4.0   # wave64 fp64 op: 16 lanes per clock per SIMD
VALU_PEAK_GINSTR = N_SIMD * CLOCK_HZ / CYCLES_PER_FP64_WAVE_INSTR / 1e9   # 614.4
METHODOLOGY_VERSION = 5
AMAX = {"SoftPendulum-v0": 22.0, "SoftPendulum3D-v0": 1.0, "OctoArmSingle-v0": 6.0,
        "OctoFlat-v0": 22.0, "SoftArmTracking-v0": 1.0, "OctoArmPush-v0": 1.0, "OctoArmPush-v1": 1.0, "OctoArmPullWeight-v0": 1.0,
        "OctoCrawl-v0": 1.0, "OctoArmTwo-v0": 1.0, "OctoReach-v0": 1.0}
MUSCLE_OCTOPUS_ENVS = ("OctoCrawl-v0", "OctoArmTwo-v0", "OctoReach-v0")          # n_arm muscle arms on a rigid head
MUSCLE_ENVS = ("OctoArmPush-v0", "OctoArmPush-v1", "OctoArmPullWeight-v0") + MUSCLE_OCTOPUS_ENVS      # COOMM muscles: parity unpinned
HEADED_ENVS = ("OctoFlat-v0",) + MUSCLE_OCTOPUS_ENVS                             # several rods per env


def random_actions(np, env_id: str, shape, amax: float, seed: int = 1):
    """Synthetic actions of the env's own action space: uniform in [-amax, amax]; OctoArmPush-v1: uniform in [0, 1]
    (Box(0, 1), arm_push_env.py:113-115); OctoArmPush-v0: 0 / 1 (Discrete(2), :101) held for four env.steps each (an
    inchworm stroke lasts several steps; flipping every step drives the restated muscle law out of its range)."""
    rng = np.random.default_rng(seed)
    if env_id in MUSCLE_OCTOPUS_ENVS:          # Box(0, 1) per arm; 0.6 keeps the restated cubic force-length law in its range
        return (rng.uniform(0.0, 0.6, shape) * (1.0 if amax else 0.0)).astype(np.float32)
    if env_id in ("OctoArmPush-v1", "OctoArmPullWeight-v0"):
        return (rng.uniform(0.0, 1.0, shape) * (1.0 if amax else 0.0)).astype(np.float32)
    if env_id == "OctoArmPush-v0":
        T = shape[0]
        strokes = rng.integers(0, 2, ((T + 3) // 4,) + tuple(shape[1:]))
        return (np.repeat(strokes, 4, axis=0)[:T] * (1 if amax else 0)).astype(np.float32)
    return rng.uniform(-amax, amax, shape).astype(np.float32)
TAPER_RATIO = 12.0           # base : tip radius of the tapered arm (octopus/arm_push_env.py:160-165: 0.012 : 0.001)


def algorithmic_bytes_per_rod_substep(n_elem: int, sizeof_real: int = 8) -> int:
    return 2 * (18 * n_elem + 6) * sizeof_real  # 14 496 B for n = 50, fp64


def usable_cpus() -> int:
    """CPUs this process can really run on: the cgroup CPU quota when there is one (a box that
    shows 256 CPUs in its affinity mask may grant a dozen), else the affinity mask."""
    n = len(os.sched_getaffinity(0))
    for path in ("/sys/fs/cgroup/cpu.max", "/sys/fs/cgroup/cpu/cpu.cfs_quota_us"):
        try:
            txt = Path(path).read_text().split()
            if path.endswith("cpu.max"):
                if txt[0] != "max":
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]) + 0.5)))
            else:
                q = int(txt[0])
                if q > 0:
                    per = int(Path("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read_text())
                    n = min(n, max(1, int(q / per + 0.5)))
            break
        except (OSError, ValueError, IndexError):
            continue
    return n


def cpu_baseline(cfg, cores: int, n_rods: int = ENVS_PER_GPU, budget_s: float = 10.0):
    """Time the C oracle (OpenMP over rods) on the workload's own batch: `n_rods` rods (4096:
    SURVEY §8(d)), env.steps until ~budget_s of wall time (at least 2)."""
    import numpy as np

    os.environ["OMP_NUM_THREADS"] = str(cores)
    from gym_softrobot_amd.seeding import initial_angle, np_random
    from oracle import oracle_c

    oracle_c.build()
    batch = oracle_c.OracleBatch(cfg, n_rods, omp=True)
    batch.reset([initial_angle(np_random(i)[0]) for i in range(n_rods)])
    acts = np.random.default_rng(1).uniform(-22, 22, (64, n_rods)).astype(np.float32)
    # warm-up on a slice of the time budget: one step of 4096 rods is ~1-3 s on 8-16 threads
    t0 = time.perf_counter()
    batch.env_step(acts[0])
    per_step = time.perf_counter() - t0
    t0 = time.perf_counter()
    steps = 0
    while steps < 2 or (time.perf_counter() - t0 + per_step < budget_s and steps < 60):
        batch.env_step(acts[1 + steps])
        steps += 1
    dt = time.perf_counter() - t0
    # one rod on one thread: the threaded figure divided by this one is the parallel speed-up the
    # box really delivered (SMT siblings and quotas make `cores` an upper bound)
    one = oracle_c.OracleRod(cfg)
    one.reset_pendulum(initial_angle(np_random(0)[0]))
    one.env_step(float(acts[0, 0]))
    t1 = time.perf_counter()
    k = 0
    while time.perf_counter() - t1 < 1.0:
        one.env_step(float(acts[1 + k % 60, 0]))
        k += 1
    dt1 = time.perf_counter() - t1
    value, single = n_rods * steps / dt, k / dt1
    return {
        "value": value,
        "unit": "env-steps/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{n_rods} rods x {steps} env.steps (400 substeps, 50 elements, fp64 C oracle, "
                  f"OpenMP {cores} threads, {dt:.1f} s; the batch the GPU line steps)",
        "single_thread_value": single,
        "single_thread_sample": f"1 rod x {k} env.steps on 1 thread, {dt1:.1f} s",
        "measured_parallel_speedup": value / single,
        "affinity_cpus": len(os.sched_getaffinity(0)),
    }


def parity_vs_oracle(gsa, torch, device, math_mode, cfg, n_envs: int = 16, steps=(1, 3, 100)):
    """BASELINE.md §3's last item: max |d obs| / |obs| of the HIP path against the CPU oracle.  After env.step 1 over
    the WHOLE benchmark batch (all ENVS_PER_GPU envs; the oracle steps them with OpenMP over rods, about a second),
    then after 3 / 100 env.steps on the first `n_envs` envs (same seeds i -> theta0_i, same action script; an env's
    trajectory does not depend on the batch it is stepped in — bitwise, tests/test_gpu_parity.py).  The oracle is the
    CHECKER here, as in smoke()."""
    import numpy as np

    from gym_softrobot_amd.seeding import initial_angle, np_random
    from oracle import oracle_c

    T = max(steps)
    N = ENVS_PER_GPU
    acts = np.random.default_rng(1).uniform(-22, 22, (T, N, 1)).astype(np.float32)[:, :, 0]
    env = gsa.make_vec("SoftPendulum-v0", N, device=device, math_mode=math_mode)
    env.reset(seed=0)
    c1 = cfg.copy()
    c1.n_envs = 1
    thetas = [initial_angle(np_random(i)[0]) for i in range(N)]
    batch = oracle_c.OracleBatch(c1, N, omp=True)
    batch.reset(thetas)
    out = {"envs": n_envs, "steps": list(steps), "max_rel_obs": [], "max_rel_reward": [], "flags_equal": True,
           "tolerance": 1e-5,
           "definition": "max over envs and entries of |hip - oracle| / max(|oracle|, 1e-3); oracle = this repo's "
                         "fp64 restatement of PyElastica (parity against PyElastica itself is unpinned)"}

    def rel(h, r):
        return np.abs(h - r) / np.maximum(np.abs(r), 1e-3)

    # step 1: every env of the batch
    o, r, te, tr, _ = env.step(torch.from_numpy(acts[0].copy()).to(env.backend.device))
    ro, rr, rte, rtr = batch.env_step(acts[0])
    torch.cuda.synchronize()
    ho, hr = o.cpu().numpy().astype(np.float64), r.cpu().numpy()
    per_env = np.maximum(rel(ho, ro.astype(np.float64)).max(axis=1), rel(hr, rr))
    worst = int(per_env.argmax())
    flags1 = bool((te.cpu().numpy().astype(bool) == rte).all() and (tr.cpu().numpy().astype(bool) == rtr).all())
    out["whole_batch_step1"] = {"envs": N, "max_rel_obs": float(rel(ho, ro.astype(np.float64)).max()),
                                "max_rel_reward": float(rel(hr, rr).max()), "flags_equal": flags1,
                                "worst_env": worst, "worst_env_xcd": worst % 8,
                                "bit_identical_obs_envs": int((o.cpu().numpy() == ro).all(axis=1).sum())}
    rods = batch.rods[:n_envs]
    for t in range(T):
        if t > 0:
            o, r, te, tr, _ = env.step(torch.from_numpy(acts[t].copy()).to(env.backend.device))
            ref = [rod.env_step(acts[t, i]) for i, rod in enumerate(rods)]
        else:
            ref = [(ro[i], rr[i], rte[i], rtr[i]) for i in range(n_envs)]
        if (t + 1) in steps:
            torch.cuda.synchronize()
            ho, hr = o.cpu().numpy().astype(np.float64)[:n_envs], r.cpu().numpy()[:n_envs]
            po = np.stack([x[0] for x in ref]).astype(np.float64)
            pr = np.array([x[1] for x in ref])
            out["max_rel_obs"].append(float(rel(ho, po).max()))
            out["max_rel_reward"].append(float(rel(hr, pr).max()))
            out["flags_equal"] = out["flags_equal"] and bool(
                (te.cpu().numpy().astype(bool)[:n_envs] == np.array([x[2] for x in ref])).all()
                and (tr.cpu().numpy().astype(bool)[:n_envs] == np.array([x[3] for x in ref])).all())
    w = out["whole_batch_step1"]
    out["within_tolerance"] = bool(max(out["max_rel_obs"] + out["max_rel_reward"] + [w["max_rel_obs"], w["max_rel_reward"]])
                                   <= out["tolerance"] and out["flags_equal"] and w["flags_equal"])
    env.close()
    return out


def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 20 warm-up + 5 windows of 20 env.steps stay inside one SoftPendulum episode (truncation
    # fires on step 126).  Five windows, not one of 100 steps: `value` is the median window, so ONE host
    # hiccup (r4b: a 1.5 ms env.step() call idled the GPU, the clock dropped, the next 20 launches ran
    # 3-10 % slow and a single 100-step window read 12.9 M where the five-window runs read 13.97 M)
    # cannot set the figure
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs-per-gpu", type=int, default=None,
                    help=f"default {ENVS_PER_GPU}; OctoFlat-v0: 1024 (BASELINE configs[4]: 8192 envs on 8 GPUs)")
    ap.add_argument("--math-mode", choices=["fast", "libm"], default="fast")
    ap.add_argument("--env", default="SoftPendulum-v0",
                    choices=["SoftPendulum-v0", "SoftPendulum3D-v0", "OctoArmSingle-v0", "OctoFlat-v0", "SoftArmTracking-v0",
                             "OctoArmPush-v0", "OctoArmPush-v1", "OctoArmPullWeight-v0", "OctoCrawl-v0", "OctoArmTwo-v0", "OctoReach-v0"],
                    help="headline metric is SoftPendulum-v0; the others are the widened §8 rows")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak",
                    help="weak (default, BASELINE configs[3]): --envs-per-gpu envs on every GPU; "
                         "strong: that many envs in total, split over the GPUs (SURVEY.md §8d cfg 4)")
    ap.add_argument("--autoreset", choices=["auto", "off", "host", "device"], default="auto",
                    help="NEXT_STEP auto-reset of finished envs.  auto (default): off while warmup + steps "
                         "stay inside one SoftPendulum episode (125 steps), else 'device' (staged reset "
                         "records, no host read) — the reference's own loop resets a truncated env, and "
                         "a pendulum driven by random forces for more than ~7 s of simulated time blows up")
    ap.add_argument("--actions", choices=["random", "zero"], default="random",
                    help="random (default): uniform in the action box; zero: SURVEY.md §8(d)'s zero-action run")
    ap.add_argument("--n-elems", type=int, default=None, help="elements per rod (env default if omitted)")
    ap.add_argument("--taper", action="store_true",
                    help=f"OctoArmSingle-v0 only: a TAPERED arm, radius falling {TAPER_RATIO:g}:1 from base to tip "
                         "like the reference's muscle arms (octopus/arm_push_env.py:160-179) — the per-lane "
                         "material-table instantiation of the step kernel")
    ap.add_argument("--no-secondary", action="store_true",
                    help="N=1 headline run: skip the two secondary workloads (BASELINE configs[2], configs[4]'s share)")
    ap.add_argument("--no-sustained", action="store_true",
                    help="N=1 headline run: skip the closing >= 8 s continuous-load leg")
    ap.add_argument("--p2p-trial", action="store_true",
                    help="N>1 with the default transport: AFTER the line is printed, run the same windows once more over "
                         "the experimental transport p2p in child processes (bounded by 90 s; figures on stderr).  "
                         "Opt-in since round 5: the transport has never run between different devices, and the "
                         "driver's N = 1, 2, 4, 8 runs share one node")
    ap.add_argument("--no-p2p-trial", action="store_true", help=argparse.SUPPRESS)    # (round 4's switch: the default now)
    ap.add_argument("--trial-child", action="store_true", help=argparse.SUPPRESS)
    ap.add_argument("--transport", choices=["rccl", "p2p"], default="rccl",
                    help="N > 1: how the packed rows reach the other ranks.  rccl (default, BASELINE's north_star): "
                         "one all_gather_into_tensor per step; p2p: every rank copies its rows into its block of "
                         "every peer's buffer (IPC-mapped, xGMI point to point), no collective call per step")
    ap.add_argument("--windows", type=int, default=0,
                    help="timed windows of --steps steps each (value = the median window); 0 (default): as many as "
                         "fit one SoftPendulum episode together with the warm-up (120 steps), at most 5, at least 1")
    ap.add_argument("--preheat", type=float, default=600.0,
                    help="cap of the un-timed pre-heat on a scratch batch, in ms of kernel time (0 = none); it "
                         "ends earlier once the kernel time has been stable for three groups after 120 ms")
    return ap.parse_args(argv)


def self_launch(args, script=None, argv=None) -> int:
    """`python bench.py --gpus N` (N > 1, not under torchrun): start the N ranks as a CHILD
    `python -m torch.distributed.run` — this parent has not imported torch or touched the GPU,
    and it never exec()s — relay rank 0's JSON line on stdout (everything else of the child's
    stdout goes to stderr) as soon as it arrives, and return the child's exit code.  The child runs
    in its own process group and is killed as a group when it exceeds SOFTROD_BENCH_LAUNCH_TIMEOUT
    seconds (default 1800) or when this parent is interrupted; a job killed at the limit AFTER its line
    was relayed (a hung p2p trial, a hung shutdown) still counts as measured (return code 0).  `script` (tests only): the file the ranks run,
    default this one."""
    import signal
    import socket
    import subprocess
    import threading

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port),
           str(Path(script or __file__).resolve())] + list(sys.argv[1:] if argv is None else argv)
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True, env=env, start_new_session=True, bufsize=1)
    limit = float(os.environ.get("SOFTROD_BENCH_LAUNCH_TIMEOUT", "1800"))
    timed_out = threading.Event()

    def kill_tree():
        """torchrun starts its workers in process groups of their own: killing torchrun's group alone would
        orphan them (and they hold the stdout pipe open).  Collect the descendants FIRST (by parent pid,
        from /proc), then kill the group and every one of them."""
        kids, todo = [], [proc.pid]
        try:
            ppid = {}
            for d in os.listdir("/proc"):
                if d.isdigit():
                    try:
                        ppid[int(d)] = int(Path("/proc", d, "stat").read_text().rsplit(")", 1)[1].split()[1])
                    except (OSError, ValueError, IndexError):
                        pass
            while todo:
                cur = todo.pop()
                for pid, par in ppid.items():
                    if par == cur and pid not in kids:
                        kids.append(pid)
                        todo.append(pid)
        except OSError:
            pass
        try:
            os.killpg(proc.pid, signal.SIGKILL)
        except OSError:
            pass
        for pid in kids:
            try:
                os.kill(pid, signal.SIGKILL)
            except OSError:
                pass

    def watchdog():
        try:
            proc.wait(timeout=limit)
        except subprocess.TimeoutExpired:
            timed_out.set()
            kill_tree()

    threading.Thread(target=watchdog, daemon=True).start()
    n_json = 0
    try:
        # rank 0's line is relayed THE MOMENT it arrives, not when the child job ends: what follows it in
        # the ranks (the p2p trial, the shutdown of the process group) cannot hold it back or lose it
        for ln in proc.stdout:
            txt = ln.strip()
            is_line = False
            if txt.startswith("{") and '"metric"' in txt:
                try:
                    json.loads(txt)
                    is_line = True
                except ValueError:
                    pass
            if is_line:
                n_json += 1
                print(txt, flush=True)
            else:
                sys.stderr.write(ln)
        proc.wait()
    except BaseException as exc:           # KeyboardInterrupt, a broken pipe: do not leave ranks behind
        kill_tree()
        proc.wait()
        sys.stderr.write(f"bench.py: child launch aborted ({type(exc).__name__})\n")
        raise
    if timed_out.is_set():
        sys.stderr.write(f"bench.py: child launch exceeded {limit:g} s and was killed"
                         + (" (the headline line had been relayed)" if n_json else "") + "\n")
        return 0 if n_json == 1 else 124
    rc = proc.returncode
    if rc == 0 and n_json != 1:
        sys.stderr.write(f"bench.py: expected one JSON line from rank 0, saw {n_json}\n")
        return 1
    return rc


def load_profile_table(name: str, key: str):
    """profiles/<name>.json[key] or None — tracked rocprofv3 evidence keyed by workload."""
    f = ROOT / "profiles" / name
    try:
        return json.loads(f.read_text()).get(key)
    except (OSError, ValueError):
        return None


def fresh_or_none(rec, lib_hash):
    """A profile-table entry is used only when it was measured on the library that is loaded now
    (entry["source_hash"] == softrod_source_hash()).  Returns (entry or None, reason or None)."""
    if rec is None:
        return None, "no tracked profile entry for this workload"
    h = rec.get("source_hash")
    if lib_hash is None:
        return None, "the stepping backend is not libsoftrod_hip.so"
    if h != lib_hash:
        return None, (f"profile entry was measured on source hash {h}, the loaded library is {lib_hash}: "
                      "re-run tools/profile_all.sh")
    return rec, None


def useful_lane_fraction(cfg, octo: bool, n_waves: int) -> float:
    """Lanes that carry a node / lanes launched, for one env: 51/64 (SoftPendulum, 50 elements),
    101/128 (100 elements on two slots per lane or two windows), 88/128 (OctoFlat: 8 arms x 11)."""
    n_nodes = int(cfg.n_elem) + 1
    if octo:
        return int(cfg.n_arm) * n_nodes / (64.0 * n_waves)
    return n_nodes / (64.0 * (1 if n_nodes <= 64 else 2))


def preheat(make_scratch, acts_dev, cap_ms: float, min_ms: float = 120.0, agree=None):
    """Un-timed launches on a SCRATCH batch of the measured shape until the step kernel's HIP-event
    duration is STABLE.  What the pool's boxes do under this load (4096 SoftPendulum envs, per-launch
    times of five runs in profiles/README.md "bench.py on the settled clock"): the clock ramps for
    ~45 ms (0.38 -> 0.297 ms per launch over 140 launches), and ~55 and ~110 ms after the load began
    the power controller pulls it back (group means 0.305 0.344 0.348 0.329 0.316 0.309 0.304 ... 0.303
    0.323 0.371 0.340 0.330 0.317 0.309 0.304 0.303 0.303), after which it holds.  All of this is a
    matter of TIME under load, not of launches, so the pre-heat is too: groups
    of ~6 ms of kernel time; stable = the last three group means within 0.5 % of each other, after at
    least `min_ms` and at most `cap_ms` of kernel time.  The scratch batch is stepped with ZERO
    actions and never reset on the way (driven at random for more than an episode a pendulum blows
    up and its steps get slower; a reset idles the GPU for ~10 ms of host work, enough for the
    clock to fall back: 0.297 -> 0.365 ms measured).  It stays allocated until the end of the run;
    the warm-up of the measured batch follows at once.  With several ranks (`agree`: a reduction
    over the ranks) everybody heats until EVERY rank is stable or one has reached the cap, so that
    all of them leave together: a rank that finished early would sit idle in the first collective
    of the warm-up and lose the clock it had just reached."""
    import numpy as np

    if cap_ms <= 0:
        return None, {"launches": 0}
    scratch = make_scratch()
    scratch.reset(seed=10_000_019)
    be = scratch.backend
    zero = acts_dev[0] * 0
    be.set_timing(2)
    scratch.step(zero)
    scratch.step(zero)
    first = be.kernel_times_ms()
    G = int(min(20, max(2, round(6.0 / max(float(first[1]), 1e-3)))))
    groups, n, busy = [], 2, float(first.sum())

    def stable():
        if len(groups) < 3:
            return False
        last = [float(np.mean(g)) for g in groups[-3:]]
        return max(last) <= 1.005 * min(last)

    def go_on():
        done, capped = (busy >= min_ms and stable()), busy >= cap_ms
        if agree is not None:
            done, capped = agree(done, capped)
        return not (done or capped)

    while go_on():
        be.set_timing(G)
        for k in range(G):
            scratch.step(zero)
        kt = be.kernel_times_ms()
        n += G
        busy += float(kt.sum())
        groups.append([float(x) for x in kt])
    be.set_timing(0)
    return scratch, {"launches": n, "kernel_ms_total": busy, "first_kernel_ms": float(first[0]),
                     "settled_kernel_ms": float(np.mean(groups[-1])) if groups else float(first[1]),
                     "group_launches": G, "group_means_ms": [round(float(np.mean(g)), 4) for g in groups],
                     "stable": stable(),
                     "what": "scratch batch of the same env/size, zero actions, not the measured batch"}


def taper_profile(base_radius: float, n_elem: int):
    """Element radii of the tapered arm: node radii falling linearly base -> base / TAPER_RATIO, element
    value = mean of its two nodes (octopus/arm_push_env.py:163-165, scaled to this arm's base radius so
    that the base rests on the plane like the uniform arm does)."""
    import numpy as np

    edge = np.linspace(base_radius, base_radius / TAPER_RATIO, n_elem + 1)
    return (edge[:-1] + edge[1:]) / 2


def workload_name(env_id: str, cfg, n_local: int, world: int = 1, taper: bool = False, libm: bool = False) -> str:
    octo = env_id in HEADED_ENVS
    taper = taper or env_id in MUSCLE_ENVS          # the muscle arm is tapered 12:1 by construction
    return (f"{env_id}, {n_local} envs x " + (f"{int(cfg.n_arm)} arms x " if octo else "")
            + f"{int(cfg.n_elem)} elements per GPU " + ("tapered " if taper else "") + ("libm kernel " if libm else "")
            + ("+ COOMM muscle layers, PARITY UNPINNED " if env_id in MUSCLE_ENVS else "")
            + (f"(BASELINE configs[1]; x{world} GPUs)" if env_id == "SoftPendulum-v0"
               else "(widened row of SURVEY §8; not the headline metric)"))


def profile_key(env_id: str, n_elem: int, taper: bool = False, libm: bool = False) -> str:
    taper = taper or env_id in MUSCLE_ENVS      # (workload_name says "tapered": tools/update_profile_tables.py keys on it)
    return f"{env_id}|n_elem={n_elem}" + ("|taper" if taper else "") + ("|libm" if libm else "")


def roofline_block(env_id, cfg, n_local, kernel_ms, math_mode, lib_hash, hip, backend, taper=False):
    """The `roofline` object of one workload: priced against the tracked rocprofv3 tables, used only
    when they were measured on THIS env / size / build of the library."""
    nsub = int(cfg.n_substeps)
    octo = env_id in HEADED_ENVS
    rods_per_env = int(cfg.n_arm) if octo else 1
    rod_substeps = n_local * rods_per_env * nsub
    # OctoFlat: n_arm rods + the rigid head (x, v, Q, w = 18 doubles read and written)
    bytes_per_launch = n_local * nsub * (rods_per_env * algorithmic_bytes_per_rod_substep(int(cfg.n_elem))
                                         + (2 * 18 * 8 if octo else 0))
    kernel_s = kernel_ms * 1e-3
    key = profile_key(env_id, int(cfg.n_elem), taper, math_mode != "fast")
    traffic_rec, traffic_why = fresh_or_none(load_profile_table("hbm_traffic.json", f"{key}|envs={n_local}"), lib_hash)
    traffic = (traffic_rec or {}).get("hbm_bytes_per_launch")
    valu_rec, valu_why = fresh_or_none(load_profile_table("valu_counts.json", key), lib_hash)
    valu_per = (valu_rec or {}).get("valu_instr_per_rod_substep")
    achieved = None if valu_per is None else valu_per * rod_substeps / kernel_s / 1e9
    frac = None if achieved is None else achieved / VALU_PEAK_GINSTR
    n_waves = 1
    if octo:
        n_waves = -(-int(cfg.n_arm) * int(backend.state()["arm_stride"]) // 64) if hip else 2
    lanes = useful_lane_fraction(cfg, octo, n_waves)
    busy_raw = (valu_rec or {}).get("valu_busy_frac")
    busy = None if busy_raw is None else min(1.0, busy_raw)
    return {
        # what bounds a register-resident kernel: fp64 VALU issue slots
        "bound": "fp64_valu",
        "achieved": achieved,
        "peak": VALU_PEAK_GINSTR,
        "unit": "G wave64-VALU-instr/s",
        # the same two numbers in the contract's TFLOP/s: a wave64 instruction is 64 lanes, an FMA 2 flops —
        # what the issue slots the kernel fills would deliver if every one of them held an FMA (30 % do)
        "achieved_tflops_fma_equivalent": None if achieved is None else achieved * 128.0 / 1e3,
        "peak_tflops": VALU_PEAK_GINSTR * 128.0 / 1e3,
        # nominal: every VALU wave-instruction priced at 4 cycles at the 2.4 GHz peak clock
        "frac": frac,
        # PRIMARY: the share of the kernel's own cycles in which the VALU was executing an instruction,
        # from the profiler's cycle counters (clock-independent, and each instruction weighs what it
        # really occupies the pipe for): SQ_ACTIVE_INST_VALU x 4 / (GRBM_GUI_ACTIVE / 8 x 1024 SIMDs) of
        # the hash-matched table; 1 - this is the idle share an implementation could still win back
        "frac_cycle_weighted": busy,
        "frac_withheld": valu_why,
        "traffic": traffic,
        "traffic_withheld": traffic_why,
        # lanes that carry a node / lanes launched: `frac` counts every issued wave instruction
        # as work, idle lanes included; frac x useful_lane_frac is the share of the chip's fp64
        # LANE-slots that advance a node
        "useful_lane_frac": lanes,
        "frac_of_lane_slots": None if frac is None else frac * lanes,
        "kernel": "softrod_octo_step_kernel (muscle-arm instantiation; softrod_mocto_action / _epilogue kernels either side)"
                  if env_id in MUSCLE_OCTOPUS_ENVS else "softrod_octo_step_kernel" if octo else
                  "softrod_octo_step_kernel (ArmPullWeight instantiation: one arm wave + the rigid weight)"
                  if env_id == "OctoArmPullWeight-v0" else
                  "softrod_step_window_kernel + softrod_step_fast_kernel (epilogue only)"
                  if (env_id == "OctoArmSingle-v0" and 64 <= int(cfg.n_elem) <= 102 and math_mode == "fast") else
                  (("softrod_step_fast_kernel" + (" (TAPER instantiation)" if taper or env_id in MUSCLE_ENVS else ""))
                   if math_mode == "fast" else "softrod_step_libm_kernel"),
        "kernel_ms_avg": kernel_ms,
        "peak_definition": f"{N_SIMD} SIMDs x {CLOCK_HZ / 1e9} GHz / {CYCLES_PER_FP64_WAVE_INSTR:g} cycles per "
                           "wave64 fp64 instruction (= 78.6 TFLOP/s of FMAs, MI355X_MICROARCH.md).  `frac` prices EVERY "
                           "VALU instruction at those 4 cycles; what the classes of the loop really occupy the pipe "
                           "for: fp64 FMA / mul / add 4 cycles, b32 ops incl. the DPP wave shifts 2, the quarter-rate "
                           "v_rsq_f64 / v_rcp_f64 seeds 16 (the SoftPendulum loop: ~83 fp64, 14 b32 DPP, 3 seeds per "
                           "substep -> 408 cycles against the nominal 400).  `frac_cycle_weighted` needs no such "
                           "prices: it is the measured busy share of the measured cycles",
        # ONE measured sample of a pure stream of independent v_fma_f64 (tools/microbench/valu_issue.hip,
        # four waves per SIMD: 4.2-4.6 cycles per instruction at the 1.9-2.0 GHz the clock settles at
        # under that load).  Context only, NOT a ceiling: a mixed stream (mul/add/DPP next to the
        # FMAs) draws less power, clocks at 2.25-2.3 GHz and can exceed it, as `achieved` does
        "fma_stream_sample": {"value": 466.0, "unit": "G wave64-VALU-instr/s",
                              "source": "profiles/history/r2j_valu_issue_microbench.txt",
                              "note": "one sample at ~1.9-2.0 GHz; not a ceiling on `achieved`"},
        "valu_instr_per_rod_substep": valu_per,
        "valu_instr_source": (valu_rec or {}).get("source"),
        "rod_substeps_per_launch": rod_substeps,
        # (the cycle-weighted fraction again under its round-3 names; raw can exceed 1 by ~1 %:
        # GRBM_GUI_ACTIVE / 8 is the mean over the XCDs' active cycles)
        "profiled_valu_busy_frac": busy,
        "profiled_valu_busy_frac_raw": busy_raw,
        "hbm": {
            "measured_traffic_GBs": None if traffic is None else traffic / kernel_s / 1e9,
            "measured_traffic_frac_of_8TBs": None if traffic is None else traffic / kernel_s / 1e9 / HBM_PEAK_GBS,
            "traffic_source": (traffic_rec or {}).get("source"),
            "minimum_bytes_per_launch": n_local * rods_per_env * (18 * int(cfg.n_elem) + 6) * 8 * 2,
            "streaming_model_GBs": bytes_per_launch / kernel_s / 1e9,
            "streaming_model_bytes_per_launch": bytes_per_launch,
            "streaming_model_note": "SURVEY 8d's algorithmic bytes (rods x substeps x 2(18n+6) x 8 B) / kernel "
                                    "time: what a one-substep-per-pass implementation would have to move; this "
                                    "kernel keeps the state in registers for all substeps, so the figure exceeds "
                                    "the 8 TB/s peak by construction and is NOT a fraction of anything",
        },
    }


# BASELINE.json configs[2] and configs[4]'s per-GPU share: measured after the headline, same process
SECONDARY = (
    dict(env_id="OctoArmSingle-v0", n_local=4096, extra={"n_elems": 100}, baseline="configs[2]"),
    dict(env_id="OctoFlat-v0", n_local=1024, extra={}, baseline="configs[4], one GPU's share of 8192 envs"),
    # the headline workload on the kernel that evaluates the substep literally as PyElastica writes it (libm
    # sin / cos / acos / exp, IEEE division, no planar specialisation): what the fast-math reformulations buy
    dict(env_id="SoftPendulum-v0", n_local=ENVS_PER_GPU, extra={}, force_math_mode="libm",
         baseline="configs[1] in math_mode libm (the reference-literal arithmetic; not the headline)"),
    # the COOMM muscle arm (arm_push_env.py), continuous mode, 40 elements tapered 12:1, 500 substeps
    dict(env_id="OctoArmPush-v1", n_local=ENVS_PER_GPU, extra={},
         baseline="none: BASELINE configs[2]'s `+ muscle actuation` taken literally (the reference's COOMM muscle arm); "
                  "PARITY UNPINNED - the muscle law restates the published model, COOMM is not on disk"),
)


def secondary_workload(gsa, _capi, torch, device, math_mode, lib_hash, env_id, n_local, extra, baseline,
                       steps: int = 10, warmup: int = 3, force_math_mode=None):
    """`steps` timed env.steps of one more workload on the already warm clock (the headline run has
    just kept the GPU under load for ~1 s; `warmup` steps take the workload's own one-off costs)."""
    import numpy as np

    if force_math_mode is not None:
        math_mode = _capi.MATH_FAST if force_math_mode == "fast" else _capi.MATH_LIBM

    env = gsa.make_vec(env_id, n_local, device=device, math_mode=math_mode, **extra)
    env.reset(seed=0)
    adim = env.backend.action_dim
    acts = random_actions(np, env_id, (warmup + steps, n_local, adim), AMAX[env_id])
    acts_dev = torch.from_numpy(acts).to(env.backend.device)
    env.backend.set_timing(warmup + steps)
    for t in range(warmup):
        env.step(acts_dev[t])
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(warmup, warmup + steps):
        obs, rew, term, trunc, _ = env.step(acts_dev[t])
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    kt = env.backend.kernel_times_ms()
    per_step = 2 if len(kt) == 2 * (warmup + steps) else 1      # the windowed arm runs two kernels per env.step
    kernel_ms = float(np.sum(kt[per_step * warmup:])) / steps
    cfg = env.cfg
    out = {
        "workload": workload_name(env_id, cfg, n_local, libm=math_mode != _capi.MATH_FAST),
        "baseline_config": baseline,
        "parity_label": gsa.parity_label(env_id),        # None, or what is NOT pinned (the COOMM muscle envs)
        "math_mode": "fast" if math_mode == _capi.MATH_FAST else "libm",
        "value": n_local * steps / elapsed, "unit": "env-steps/s",
        "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3,
        "kernel_ms_avg": kernel_ms,
        "substeps_per_env_step": int(cfg.n_substeps),
        "non_finite_envs_at_end": int((~torch.isfinite(obs).all(dim=1)).sum().item()),
        "last_step_checksum": float(torch.nan_to_num(obs.double()).sum().item()) + float(torch.nan_to_num(rew.double()).sum().item()),
        "roofline": roofline_block(env_id, cfg, n_local, kernel_ms, "fast" if math_mode == _capi.MATH_FAST else "libm",
                                   lib_hash, True, env.backend),
    }
    if env_id == "OctoArmPush-v1":
        # the CPU beside it: the C oracle's own ArmPush env.step (the restated muscle law on the same tapered arm), one
        # thread, a bounded sample of the same actions; and env 0..15 after step 1 against it (rtol 1e-5)
        try:
            out["cpu_port"] = muscle_arm_cpu_port(np, _capi, cfg, acts, obs_step1=None)
        except Exception as exc:  # noqa: BLE001
            out["cpu_port"] = {"error": repr(exc)}
    env.close()
    return out


def muscle_arm_cpu_port(np, _capi, cfg, acts, obs_step1=None, n_envs: int = 4, steps: int = 3):
    """The oracle's OctoArmPush-v1 env.step on ONE host thread (oracle/softrod_oracle.c oracle_env_step_push: the same
    restated COOMM law, PARITY UNPINNED like the kernel's): `n_envs` envs x `steps` env.steps of the benchmark's actions."""
    from oracle import oracle_c

    c1 = cfg.copy()
    c1.n_envs = 1
    radii = _capi.arm_push_radii(int(cfg.n_elem))
    layers = _capi.es_muscle_layers(radii, 0.012)
    rods = []
    for _ in range(n_envs):
        r = oracle_c.OracleRod(c1)
        r.set_radius_profile(radii)
        r.set_muscle_layers(*layers)
        r.reset_push()
        rods.append(r)
    t0 = time.perf_counter()
    for t in range(steps):
        for i, r in enumerate(rods):
            r.env_step_push(acts[t, i])
    el = time.perf_counter() - t0
    return {"value": n_envs * steps / el, "unit": "env-steps/s", "cores": 1, "kind": "port",
            "sample": f"{n_envs} envs x {steps} env.steps of the benchmark's actions, one thread, the C oracle's ArmPush env "
                      "(restated muscle law: parity unpinned)"}


def pcie_inclusive(gsa, torch, device, math_mode, n_local, steps: int = 40, warmup: int = 5):
    """The same workload when the caller holds HOST buffers (the reference's own situation: NumPy in,
    NumPy out): per env.step the actions cross PCIe from pinned memory, the step runs, obs / reward /
    flags come back into pinned memory, and the host waits for them before it can act again (a closed
    loop: one stream synchronisation per step, so the launch latency is exposed too).  Reported beside
    `value`, never as `value` (which is measured with inputs resident in HBM)."""
    import numpy as np

    env = gsa.make_vec("SoftPendulum-v0", n_local, device=device, math_mode=math_mode)
    env.reset(seed=0)
    dev = env.backend.device
    acts = torch.from_numpy(np.random.default_rng(1).uniform(-22, 22, (warmup + steps, n_local, 1)).astype(np.float32)).pin_memory()
    a_dev = torch.empty((n_local, 1), dtype=torch.float32, device=dev)
    host = None
    for t in range(warmup + steps):
        if t == warmup:
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        a_dev.copy_(acts[t], non_blocking=True)
        out = env.step(a_dev)[:4]
        if host is None:
            host = [torch.empty(o.shape, dtype=o.dtype).pin_memory() for o in out]
        for h, o in zip(host, out):
            h.copy_(o, non_blocking=True)
        torch.cuda.current_stream(dev).synchronize()        # the host reads the outputs before its next action
    elapsed = time.perf_counter() - t0
    moved = acts[0].numel() * 4 + sum(h.numel() * h.element_size() for h in host)
    env.close()
    return {"value": n_local * steps / elapsed, "unit": "env-steps/s", "ms_per_step": elapsed / steps * 1e3,
            "steps": steps, "bytes_over_pcie_per_step": moved,
            "what": "pinned host actions -> device, step, obs/reward/flags -> pinned host, stream sync, every step "
                    "(closed loop); the headline `value` has its inputs resident in HBM and is open loop"}


def profiler_preload_present() -> bool:
    """True when this process runs under rocprofv3 (its preload initialises the GPU in every child; with
    --pmc a `#!/usr/bin/env python3` child such as rocm-smi is the exec of a GPU-initialised process that
    takes a box of this pool down): no child process may be started then."""
    if "rocprof" in os.environ.get("LD_PRELOAD", "").lower():
        return True
    return any(k.startswith(("ROCP", "ROCPROF")) for k in os.environ)


def rocm_smi_command(args):
    """rocm-smi as `python3 <its script>` with the profiler's variables removed from the child's
    environment — never through the `#!/usr/bin/env python3` shebang — or None under a profiler."""
    if profiler_preload_present():
        return None, None
    script = "/opt/rocm/libexec/rocm_smi/rocm_smi.py"
    if not os.path.exists(script):
        return None, None
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD" and not k.startswith(("ROCP", "ROCPROF"))}
    return [sys.executable, script] + list(args), env


def gpu_sensors(torch, device_index: int):
    """Shader clock (MHz), socket power (W) and the busy percentage of the GPU this process steps on,
    read from sysfs (the amdgpu node whose PCI address is the torch device's), with `rocm-smi` as the
    fall-back for the clock and power.  Every field is None where the box does not let an ordinary
    user read it; `source` says where the numbers came from.  Cheap (a few file reads): it can be
    called while the stream is busy."""
    import glob
    import re
    import subprocess

    out = {"sclk_mhz": None, "power_w": None, "gpu_busy_percent": None, "source": None}
    try:
        pr = torch.cuda.get_device_properties(device_index)
        want = "%04x:%02x:%02x" % (int(getattr(pr, "pci_domain_id", 0)), int(pr.pci_bus_id), int(pr.pci_device_id))
    except Exception:  # noqa: BLE001
        want = None
    cards = sorted(glob.glob("/sys/class/drm/card[0-9]*/device"))
    pick = None
    for c in cards:
        try:
            addr = os.path.basename(os.path.realpath(c))
        except OSError:
            continue
        if want is not None and addr.lower().startswith(want):
            pick = c
            break
    if pick is None and len(cards) == 1:
        pick = cards[0]
    if pick is not None:
        try:
            for ln in Path(pick, "pp_dpm_sclk").read_text().splitlines():
                if ln.rstrip().endswith("*"):
                    out["sclk_mhz"] = float(re.search(r"(\d+(?:\.\d+)?)\s*mhz", ln.lower()).group(1))
            out["source"] = "sysfs " + pick
        except (OSError, AttributeError, ValueError):
            pass
        for name in ("power1_average", "power1_input"):
            hits = glob.glob(os.path.join(pick, "hwmon", "hwmon*", name))
            if hits:
                try:
                    out["power_w"] = int(Path(hits[0]).read_text()) / 1e6
                    out["source"] = "sysfs " + pick
                    break
                except (OSError, ValueError):
                    pass
        try:
            out["gpu_busy_percent"] = float(Path(pick, "gpu_busy_percent").read_text())
        except (OSError, ValueError):
            pass
    cmd, child_env = rocm_smi_command(["-d", str(device_index), "--showclocks", "--showpower", "--showuse", "--json"])
    if out["sclk_mhz"] is None and out["power_w"] is None and cmd is None:
        out["source"] = out["source"] or "unreadable (sysfs silent; no child process under a profiler)"
    elif out["sclk_mhz"] is None and out["power_w"] is None:
        try:
            txt = subprocess.run(cmd, capture_output=True, text=True, timeout=20, env=child_env).stdout
            card = next(iter(json.loads(txt).values()))
            for k, v in card.items():
                kl = k.lower()
                if "sclk" in kl and "clock" in kl:
                    m = re.search(r"(\d+(?:\.\d+)?)", str(v))
                    out["sclk_mhz"] = float(m.group(1)) if m else None
                elif "power" in kl and "(w)" in kl and out["power_w"] is None:
                    out["power_w"] = float(v)
                elif "gpu use" in kl:
                    out["gpu_busy_percent"] = float(v)
            out["source"] = "rocm-smi"
        except Exception as exc:  # noqa: BLE001
            out["source"] = f"unreadable ({type(exc).__name__})"
    return out


def sustained_leg(gsa, torch, device, math_mode, n_local, min_seconds: float = 8.0, max_steps: int = 60000):
    """The GPU under CONTINUOUS load for >= `min_seconds` (8 s, about a quarter of the whole run: the driver takes a handful of samples of the GPU's
    busy state over the whole run, most of which is host work — imports, the CPU baseline — so the leg has to be
    long enough for one of them to land in it), so that an observer outside this process has something to see: the headline workload, random actions from
    a ring of 120 pre-staged steps, device-side NEXT_STEP auto-reset (episodes truncate on step 126 and
    restart from staged draws: no host round trip, no idle gap), as many env.steps as `min_seconds`
    takes at the headline's rate.  Restarting env-steps are not counted as work.  Reported BESIDE the
    median-window `value`, never instead of it."""
    import numpy as np

    env = gsa.make_vec("SoftPendulum-v0", n_local, device=device, math_mode=math_mode, autoreset="device")
    env.reset(seed=0)
    dev = env.backend.device
    ring = torch.from_numpy(np.random.default_rng(1).uniform(-22, 22, (120, n_local, 1)).astype(np.float32)).to(dev)
    # rate probe: 60 steps (also this env's one-off costs)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(60):
        env.step(ring[t])
    torch.cuda.synchronize()
    per_step = (time.perf_counter() - t0) / 60
    steps = int(min(max_steps, max(500, min_seconds / max(per_step, 1e-6) * 1.05)))
    r0 = int(env.backend.queue_status()[0].sum())
    sensors = {}
    smi = None
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(steps):
        env.step(ring[t % 120])
        if t == steps // 8:
            sensors["near_start"] = gpu_sensors(torch, device)
        elif t == steps // 2:
            # what `rocm-smi` itself says in the middle of the leg (the tool an outside observer is likely to
            # use), started as a child process and collected after the leg: it must not stall the launch loop
            try:
                import subprocess

                cmd, child_env = rocm_smi_command(["--showuse", "--showpower", "--json"])
                smi = None if cmd is None else subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                                                                text=True, env=child_env)
            except OSError:
                smi = None
        elif t == steps - steps // 8:
            sensors["near_end"] = gpu_sensors(torch, device)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if smi is not None:
        try:
            out, _ = smi.communicate(timeout=30)
            sensors["rocm_smi_mid_leg"] = json.loads(out)
        except Exception as exc:  # noqa: BLE001
            sensors["rocm_smi_mid_leg"] = {"error": repr(exc)}
    restarts = int(env.backend.queue_status()[0].sum()) - r0
    env.close()
    return {"value": (n_local * steps - restarts) / elapsed, "unit": "env-steps/s", "steps": steps,
            "seconds": elapsed, "ms_per_step": elapsed / steps * 1e3,
            "episode_restarts_not_counted": restarts, "autoreset": "device",
            "sensors": sensors,
            "what": "continuous load for the driver's GPU-busy samples: the headline workload with device-side "
                    "auto-reset for >= 8 s; beside `value` (the median 20-step window), never instead of it"}


def make_policy(torch, device, obs_dim: int, act_dim: int, amax: float, hidden: int = 64, seed: int = 0):
    """A 2 x `hidden` tanh MLP with fixed random weights (the shape of SB3's default PPO MlpPolicy used by
    /root/reference/examples/soft_pendulum_3d/train_ppo.py:26-38): float32 obs -> float32 actions in
    [-amax, amax].  NaN observations (a rod the integrator lost; the env restarts on the next step) give
    action 0."""
    g = torch.Generator(device="cpu").manual_seed(seed)
    w1 = (torch.randn(obs_dim, hidden, generator=g) / obs_dim ** 0.5).to(device)
    w2 = (torch.randn(hidden, hidden, generator=g) / hidden ** 0.5).to(device)
    w3 = (torch.randn(hidden, act_dim, generator=g) / hidden ** 0.5).to(device)
    b1, b2, b3 = (torch.zeros(k, device=device) for k in (hidden, hidden, act_dim))

    def policy(obs):
        with torch.no_grad():
            h = torch.tanh(torch.addmm(b1, torch.nan_to_num(obs), w1))
            h = torch.tanh(torch.addmm(b2, h, w2))
            return amax * torch.tanh(torch.addmm(b3, h, w3))
    return policy


def policy_loop(torch, env, policy, obs, steps: int, sync_every_call: bool = False, record=None):
    """obs -> policy -> env.step -> obs ..., everything enqueued on ONE stream with no host
    synchronisation (`sync_every_call` adds a device synchronise after every call: the control of
    tests/test_gpu_policy_loop.py, which must be bit-identical).  `record`: list that receives a host copy
    of (obs, reward, terminated, truncated) of every step."""
    for _ in range(steps):
        a = policy(obs)
        if sync_every_call:
            torch.cuda.synchronize()
        obs, rew, term, trunc, _ = env.step(a)
        if sync_every_call:
            torch.cuda.synchronize()
        if record is not None:
            record.append(tuple(x.detach().cpu().clone() for x in (obs, rew, term, trunc)))
    return obs


def policy_in_loop(gsa, torch, device, math_mode, n_local, steps: int = 1500, warmup: int = 60):
    """N2's "real RL throughput": the closed loop on the device (see make_policy / policy_loop), device
    auto-reset on.  Reported next to `pcie_inclusive` (the closed loop through HOST buffers)."""
    env = gsa.make_vec("SoftPendulum-v0", n_local, device=device, math_mode=math_mode, autoreset="device")
    obs, _ = env.reset(seed=0)
    policy = make_policy(torch, env.backend.device, env.obs_dim, env.action_dim, 22.0)
    obs = policy_loop(torch, env, policy, obs, warmup)
    r0 = int(env.backend.queue_status()[0].sum())
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    obs = policy_loop(torch, env, policy, obs, steps)
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    restarts = int(env.backend.queue_status()[0].sum()) - r0
    checksum = float(torch.nan_to_num(obs.double()).sum().item())
    # the same loop as ONE HIP graph per env.step (policy kernels + softrod_step captured once, replayed;
    # the queue top-ups stay outside): north_star's "hipGraphs for launch-bound inner loops"
    graph = {}
    try:
        replay = env.capture_policy_step(policy)
        for _ in range(warmup):
            replay()
        r1 = int(env.backend.queue_status()[0].sum())
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            replay()
        torch.cuda.synchronize()
        el = time.perf_counter() - t0
        rs = int(env.backend.queue_status()[0].sum()) - r1
        graph = {"graph_value": (n_local * steps - rs) / el, "graph_ms_per_step": el / steps * 1e3,
                 "graph_episode_restarts_not_counted": rs,
                 "graph_note": "policy kernels + softrod_step captured once into a HIP graph, one graph launch per env.step "
                               "(VecRodEnvBase.capture_policy_step; bit-identical to the eager loop).  No faster: the "
                               "eager loop is GPU-bound at every batch measured (64 .. 4096 envs: the host enqueues "
                               "ahead of the kernels); what the graph buys is host time, one call per step instead of ten"}
    except Exception as exc:  # noqa: BLE001
        graph = {"graph_error": repr(exc)}
    env.close()
    return {"value": (n_local * steps - restarts) / elapsed, "unit": "env-steps/s", "steps": steps,
            "ms_per_step": elapsed / steps * 1e3, "episode_restarts_not_counted": restarts, **graph,
            "policy": "2 x 64 tanh MLP, fixed random weights, torch on the env's stream, no host synchronisation",
            "autoreset": "device", "last_obs_checksum": checksum,
            "what": "obs -> MLP -> actions -> softrod_step in a closed loop on the device (what an on-device "
                    "PPO rollout pays per env.step); the headline `value` is open loop on pre-staged actions"}


def guarded(fn, *a, **kw):
    """An add-on of the line (secondary workloads, PCIe leg, CPU baseline, ...) must not cost the measured
    headline: its exception becomes its entry."""
    try:
        return fn(*a, **kw)
    except Exception as exc:  # noqa: BLE001
        import traceback

        sys.stderr.write(f"bench.py: add-on {getattr(fn, '__name__', fn)} failed: {exc!r}\n{traceback.format_exc()}")
        return {"error": repr(exc)}


def single_env_latency(gsa, np, steps: int = 60, warmup: int = 5):
    """BASELINE configs[0]'s shape on the GPU: ONE SoftPendulum-v0 env through the drop-in Gymnasium
    surface (`make(id)`, NumPy action in, NumPy observation out, a host synchronisation every step) —
    what a user of the reference gets who changes nothing but the import.  One wavefront runs the 400
    substeps alone, so this is a latency, not a throughput."""
    env = gsa.make("SoftPendulum-v0")
    env.reset(seed=0)
    acts = np.random.default_rng(1).uniform(-22, 22, (warmup + steps, 1)).astype(np.float32)
    for t in range(warmup + steps):
        if t == warmup:
            t0 = time.perf_counter()
        obs, rew, term, trunc, info = env.step(acts[t])
    elapsed = time.perf_counter() - t0
    env.close()
    return {"ms_per_env_step": elapsed / steps * 1e3, "env_steps_per_s": steps / elapsed, "steps": steps,
            "what": "gym_softrobot_amd.make('SoftPendulum-v0'): one env, NumPy in / out, synchronised every step"}


def p2p_trial(args, rank: int, local_rank: int, world: int, script=None, timeout_s: float = 90.0):
    """N > 1, after the headline (RCCL) measurement: every rank starts ONE child process that runs this
    file again as the same rank of a second job over transport "p2p" (its own rendezvous port), so that
    the driver's one run per N yields both transports — and a crash, a hang or a fall-back of the
    experimental transport is a reported fact, not a lost measurement (rank 0 has printed the headline
    before this runs).  The parents have finished their GPU work; they never exec.  The child is NOT
    detached: it stays in the launcher's process group (a killed torchrun / self_launch takes it down
    too), asks the kernel to SIGKILL it when its parent dies (PR_SET_PDEATHSIG), and is killed by its
    parent at the timeout.  -> rank 0: the child's figures; other ranks: None."""
    import ctypes
    import signal
    import subprocess

    import socket

    import torch.distributed as dist

    # a rendezvous port of the trial's own: rank 0 asks the OS for a free one and tells the others over
    # the parents' process group (which is still up)
    box = [None]
    if rank == 0:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            box[0] = sk.getsockname()[1]
    dist.broadcast_object_list(box, src=0)
    port = int(box[0])
    env = dict(os.environ, MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(local_rank), WORLD_SIZE=str(world))
    for k in ("TORCHELASTIC_RUN_ID", "TORCHELASTIC_RESTART_COUNT", "TORCHELASTIC_MAX_RESTARTS",
              "TORCHELASTIC_USE_AGENT_STORE", "TORCH_NCCL_ASYNC_ERROR_HANDLING"):
        env.pop(k, None)
    cmd = [sys.executable, str(Path(script or __file__).resolve()), "--gpus", str(world), "--steps", str(args.steps),
           "--warmup", str(args.warmup), "--env", args.env, "--scaling", args.scaling, "--math-mode", args.math_mode,
           "--actions", args.actions, "--autoreset", args.autoreset, "--transport", "p2p", "--trial-child",
           "--no-cpu-baseline", "--preheat", str(min(args.preheat, 200.0)), "--windows", str(args.windows)]
    if args.envs_per_gpu is not None:
        cmd += ["--envs-per-gpu", str(args.envs_per_gpu)]
    if args.n_elems is not None:
        cmd += ["--n-elems", str(args.n_elems)]
    prctl = ctypes.CDLL(None, use_errno=True).prctl       # resolved BEFORE the fork: the hook below only calls it
    PR_SET_PDEATHSIG = 1

    def die_with_parent():
        prctl(PR_SET_PDEATHSIG, int(signal.SIGKILL))

    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env,
                            preexec_fn=die_with_parent)
    try:
        out, err = proc.communicate(timeout=timeout_s)
        rc = proc.returncode
    except subprocess.TimeoutExpired:
        proc.kill()
        out, err = proc.communicate()
        rc = 124
    if rank != 0:
        return None
    res = {"what": "the same job once more over transport p2p, one child process per rank", "returncode": rc}
    for ln in (out or "").splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            try:
                d = json.loads(ln)
            except ValueError:
                continue
            res.update({"value": d["value"], "ms_per_step": d["ms_per_step"],
                        "transport": d["config"]["transport"], "exchange_memory": d["config"].get("exchange_memory"),
                        "transport_fallback_reason": d["config"].get("transport_fallback_reason"),
                        "last_step_checksum": d["config"]["last_step_checksum"],
                        "windows_value": d["windows"]["value"], "per_rank": d.get("per_rank")})
    if "value" not in res:
        res["stderr_tail"] = (err or "")[-1500:]
    return res


def main(argv=None, script=None) -> int:
    args = parse_args(argv)
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        return self_launch(args, script=script, argv=argv)

    import numpy as np
    import torch
    import torch.distributed as dist

    import gym_softrobot_amd as gsa
    from gym_softrobot_amd import _capi
    from gym_softrobot_amd.distributed import ShardedVecEnv

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X: the hot path has no CPU fallback")
    if os.environ.get("SOFTROD_BENCH_ALL_RANKS_ON_DEVICE0") == "1":
        local_rank = 0   # the N>1 code path on a 1-GPU box (tests/test_gpu_two_ranks.py; not a measurement)
    torch.cuda.set_device(local_rank)
    force_dist = os.environ.get("SOFTROD_BENCH_FORCE_DIST") == "1"   # RCCL smoke test in a world of one
    distributed = world > 1 or force_dist
    if distributed:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        # transport of the packed rows: RCCL ("nccl") unless told otherwise (two ranks on ONE device
        # need gloo: RCCL refuses two ranks per GPU)
        backend = os.environ.get("SOFTROD_BENCH_DIST_BACKEND", "nccl")
        from gym_softrobot_amd.distributed import init_process_group

        # RCCL's stream at high priority (SOFTROD_RCCL_HIGH_PRIORITY=0: normal): distributed.py says why
        init_process_group(backend, device=torch.device("cuda", local_rank) if backend == "nccl" else None)

    n_local = args.envs_per_gpu or (1024 if args.env in HEADED_ENVS else ENVS_PER_GPU)
    if args.scaling == "strong":
        if n_local % world:
            raise SystemExit(f"--scaling strong: {n_local} envs do not split over {world} GPUs")
        n_local //= world
    n_total = n_local * world
    K, W = args.steps, args.warmup
    R = args.windows if args.windows > 0 else max(1, min(5, (120 - W) // max(K, 1)))
    math_mode = _capi.MATH_FAST if args.math_mode == "fast" else _capi.MATH_LIBM
    extra = {} if args.n_elems is None else {"n_elems": args.n_elems}
    if args.taper:
        if args.env != "OctoArmSingle-v0":
            raise SystemExit("--taper is the tapered OctoArmSingle-v0 arm")
        extra["radius_profile"] = taper_profile(_capi.arm_single_config(1).base_radius, args.n_elems or 50)
    scratch_kw = dict(extra)
    T = W + R * K
    if args.autoreset == "auto":
        args.autoreset = "off" if (T <= 120 or args.env != "SoftPendulum-v0") else "device"
    if args.autoreset != "off":
        extra["autoreset"] = True if args.autoreset == "host" else "device"
    local = gsa.make_vec(args.env, n_local, device=local_rank, math_mode=math_mode, **extra)
    hip = type(local.backend).__name__ == "HipRodBackend"     # a test double labels its line (tests/bench_cpu_launcher.py)
    lib_hash = _capi.library_source_hash() if hip else None
    # world > 1: kernel-packed rows + one all-gather per step, issued asynchronously so that the
    # next step's kernel does not wait for it (ShardedVecEnv overlap; the final sync is timed)
    env = ShardedVecEnv(local, n_total, overlap=True, force_collective=force_dist, transport=args.transport)
    if getattr(env, "transport", "rccl") == "p2p":
        # the timed windows read nothing between two sync() calls (the checksum is taken after the closing
        # sync), so the every-step interval of the p2p contract does not apply (distributed.py)
        env.p2p_enforce_sync_interval = False
    env.reset(seed=0)                      # global env i seeded i
    lo, hi = env.lo, env.hi
    adim = local.backend.action_dim
    amax = AMAX[args.env]
    if args.actions == "zero":
        amax = 0.0
    # the truncation flag of SoftPendulum first fires on env.step #126; the default window
    # (120 steps) stays inside one episode
    acts = random_actions(np, args.env, (T, n_total, adim), amax)
    acts_dev = torch.from_numpy(acts[:, lo:hi].copy()).to(local.backend.device)

    timed = hasattr(local.backend, "set_timing")
    # Everything that costs host time once — the first call through the stepping path (lazy imports,
    # allocations of the gather buffers) and the creation of the timing events — happens BEFORE the
    # pre-heat: an idle GPU loses its clock within ~10 ms, and a cold box has been seen to spend that
    # between the pre-heat and the first timed window.  The reset puts the batch back (same seeds,
    # same draws: the state the run starts from is the one it would have started from).
    env.step(acts_dev[0])
    env.sync()
    env.reset(seed=0)
    import gc

    gc.collect()          # (tens of ms with torch loaded: here, not between the pre-heat and the windows)
    gc.disable()          # no collector pause between two launches of a timed window (re-enabled below)
    if timed:
        local.backend.set_timing(W + R * K)
    scratch, heat = None, {"launches": 0}

    def agree(done: bool, capped: bool):
        """all ranks done / any rank capped (one small all-reduce per ~6 ms group of the pre-heat)"""
        f = torch.tensor([1.0 if done else 0.0, 0.0 if capped else 1.0], device=local.backend.device)
        dist.all_reduce(f, op=dist.ReduceOp.MIN)
        f = f.cpu()
        return bool(f[0] > 0), bool(f[1] < 1)

    if timed and args.preheat > 0:
        scratch, heat = preheat(lambda: gsa.make_vec(args.env, n_local, device=local_rank, math_mode=math_mode,
                                                     **scratch_kw), acts_dev, args.preheat,
                                agree=agree if distributed else None)
    for t in range(W):
        env.step(acts_dev[t])

    def restarts_so_far():
        return int(local.backend.queue_status()[0].sum()) if args.autoreset == "device" else 0

    win_elapsed, win_restarts, win_host_worst, win_host_mean = [], [], [], []
    r_before = restarts_so_far()
    for w in range(R):
        if distributed:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tp, worst = t0, (0.0, -1)
        for t in range(W + w * K, W + (w + 1) * K):
            obs, rew, term, trunc, _ = env.step(acts_dev[t])
            tn = time.perf_counter()
            if tn - tp > worst[0]:
                worst = (tn - tp, t)
            tp = tn
        win_host_worst.append(worst)
        win_host_mean.append((tp - t0) / K)
        env.sync()
        torch.cuda.synchronize()
        if distributed:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        # env-steps that restarted an episode instead of integrating are not counted as work
        r_now = restarts_so_far()
        rs = r_now - r_before
        r_before = r_now
        if distributed:
            el = torch.tensor([elapsed], dtype=torch.float64, device=local.backend.device)
            dist.all_reduce(el, op=dist.ReduceOp.MAX)
            elapsed = float(el.item())
            rt = torch.tensor([rs], dtype=torch.int64, device=local.backend.device)
            dist.all_reduce(rt)
            rs = int(rt.item())
        win_elapsed.append(elapsed)
        win_restarts.append(rs)

    gc.enable()
    kt_all = local.backend.kernel_times_ms() if timed else np.repeat(np.asarray(win_elapsed) / K * 1e3, K)
    kt = kt_all[W:] if timed else kt_all
    assert len(kt) == R * K or args.autoreset != "off"
    win_value = [(n_total * K - win_restarts[w]) / win_elapsed[w] for w in range(R)]
    m = int(np.argsort(win_value)[(R - 1) // 2])     # the median window (the LOWER one of two)
    elapsed, restarts = win_elapsed[m], win_restarts[m]
    n_bad = int((~torch.isfinite(obs).all(dim=1)).sum().item())
    # what the last step returned, over ALL envs (gathered rows included): lets two runs be compared
    obs_checksum = float(torch.nan_to_num(obs.double()).sum().item()) + float(torch.nan_to_num(rew.double()).sum().item())

    # N > 1: what every rank measured (its step-kernel time; the rest of its step is the exchange)
    per_rank = None
    my_kernel_ms = float(np.mean(kt[m * K:(m + 1) * K])) if len(kt) == R * K else float(np.mean(kt))
    if distributed:
        mine = {"rank": rank, "kernel_ms_avg": my_kernel_ms,
                "kernel_ms_each_window": [float(np.mean(kt[w * K:(w + 1) * K])) for w in range(R)] if len(kt) == R * K else None}
        per_rank = [None] * world
        dist.all_gather_object(per_rank, mine)
    transport_ran = env.transport if distributed else None
    exchange_memory = getattr(env, "exchange_memory", None)
    p2p_error = getattr(env, "_p2p_error", None)

    line = None
    if rank == 0:
        cfg = local.cfg
        nsub = int(cfg.n_substeps)
        octo = args.env in HEADED_ENVS
        rods_per_env = int(cfg.n_arm) if octo else 1
        per_win_kernel = [float(np.mean(kt[w * K:(w + 1) * K])) for w in range(R)] if len(kt) == R * K else []
        kernel_ms = per_win_kernel[m] if per_win_kernel else float(np.mean(kt))
        ms_per_step = elapsed / K * 1e3
        line = {
            "metric": "env_steps_per_sec",
            "value": win_value[m],
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": ms_per_step,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic" if hip else "TEST-DOUBLE (not a measurement)",
            "methodology_version": METHODOLOGY_VERSION,
            "config": {
                "workload": workload_name(args.env, cfg, n_local, world, args.taper, args.math_mode != "fast"),
                "envs_total": n_total,
                "substeps_per_env_step": nsub,
                "math_mode": args.math_mode,
                "actions": args.actions,
                "autoreset": args.autoreset,
                "episode_restarts_not_counted": restarts,
                "sharding": ("contiguous env blocks per rank; "
                             + ("one packed all_gather per step" if env.transport == "rccl" else
                                "peer copies of the packed rows per step (no collective), a barrier at sync")
                             ) if (world > 1 or force_dist) else "single GPU",
                "transport": transport_ran,
                "exchange_memory": exchange_memory,
                "transport_fallback_reason": p2p_error if (args.transport == "p2p" and transport_ran != "p2p") else None,
                "rccl_stream_priority": ("high" if os.environ.get("SOFTROD_RCCL_HIGH_PRIORITY", "1") != "0" else "normal")
                                        if distributed else None,
                "rod_substeps_per_sec": (n_total * K - restarts) * rods_per_env * nsub / elapsed,
                "non_finite_envs_at_end": n_bad, "last_step_checksum": obs_checksum,
                "library_source_hash": lib_hash,
            },
            # every timed window of K steps (value / ms_per_step above are the median window's)
            "windows": {
                "count": R, "median_index": m,
                "value": win_value,
                "ms_per_step": [e / K * 1e3 for e in win_elapsed],
                "kernel_ms_avg": per_win_kernel,
                # the slowest single env.step() call of each window on the HOST (ms, step index): a window
                # starts from a synchronised stream, so a host hiccup longer than the few steps the host
                # is ahead by idles the GPU (and an idle GPU loses its clock)
                "host_call_ms_max": [[round(x * 1e3, 3), t] for x, t in win_host_worst],
                # mean host time of one env.step() call (enqueue only): the run is GPU-bound while this
                # stays below the kernel time
                "host_call_ms_mean": [round(x * 1e3, 4) for x in win_host_mean],
                "spread": (max(win_value) - min(win_value)) / win_value[m],
                # every launch of the measured batch, warm-up first (when there are few enough to print)
                "kernel_ms_each": [round(float(x), 4) for x in kt_all] if len(kt_all) <= 256 else None,
            },
            # window 0 alone: the one figure that does not depend on how many windows the arguments allow
            "single_window": {"value": win_value[0], "ms_per_step": win_elapsed[0] / K * 1e3,
                              "note": "the first timed window of K steps after the pre-heat and the warm-up"},
            "preheat": heat,
            "roofline": roofline_block(args.env, cfg, n_local, kernel_ms, args.math_mode, lib_hash, hip,
                                       local.backend, args.taper),
        }
        if per_rank is not None:
            for r_ in per_rank:
                r_["exchange_us_per_step"] = (ms_per_step - r_["kernel_ms_avg"]) * 1e3
                r_["efficiency_vs_n1_kernel"] = r_["kernel_ms_avg"] / ms_per_step
            line["per_rank"] = {
                "ranks": per_rank,
                "note": "kernel_ms_avg: each rank's step kernel (HIP events, median window); exchange_us_per_step = "
                        "job ms_per_step - that rank's kernel time: the all-gather / peer copies plus launch gaps; "
                        "efficiency_vs_n1_kernel = kernel time / step time (1.0 = the exchange costs nothing)",
            }

    # ---- N = 1: the other two single-GPU BASELINE workloads on the warm clock, then the CPU baseline
    env.close()
    if scratch is not None:
        scratch.close()
        scratch = None
    if (distributed and world > 1 and args.env == "SoftPendulum-v0" and hip and not args.no_sustained
            and not args.trial_child and args.math_mode == "fast" and args.scaling == "weak"
            and args.envs_per_gpu in (None, ENVS_PER_GPU) and args.n_elems is None):
        # N > 1: EVERY rank keeps its GPU under continuous load for >= 8 s on a shard-sized batch of its own
        # (no exchange, no collective: a rank that fails here cannot block another), so that the driver's
        # GPU-busy samples see all N GPUs; rank 0's figure goes into the line, the others' to stderr
        mine = guarded(sustained_leg, gsa, torch, local_rank, math_mode, n_local)
        sys.stderr.write(f"bench.py: sustained rank {rank}: " + json.dumps(mine) + "\n")
        sys.stderr.flush()
        if rank == 0:
            line["sustained"] = dict(mine, note=f"rank 0's GPU; all {world} ranks ran this leg at about the same time, each on "
                                                "its own shard-sized batch with no exchange (their figures: stderr, "
                                                "`bench.py: sustained rank k: {...}`); not a whole-job figure")
    if rank == 0 and world == 1 and not distributed:
        full = (args.env == "SoftPendulum-v0" and hip and not args.no_secondary and args.n_elems is None
                and args.envs_per_gpu in (None, ENVS_PER_GPU) and args.math_mode == "fast")
        if full:
            line["secondary"] = [guarded(secondary_workload, gsa, _capi, torch, local_rank, math_mode, lib_hash, **spec)
                                 for spec in SECONDARY]
            line["pcie_inclusive"] = guarded(pcie_inclusive, gsa, torch, local_rank, math_mode, n_local)
            line["policy_in_loop"] = guarded(policy_in_loop, gsa, torch, local_rank, math_mode, n_local)
            line["single_env"] = guarded(single_env_latency, gsa, np)
        if not args.no_cpu_baseline and args.env == "SoftPendulum-v0" and hip:
            line["cpu_baseline"] = guarded(cpu_baseline, cfg, usable_cpus(), n_rods=ENVS_PER_GPU)
            line["cpu_baseline"]["parity_vs_oracle"] = guarded(parity_vs_oracle, gsa, torch, local_rank, math_mode, cfg)
        else:
            line["cpu_baseline"] = None
        if full and not args.no_sustained:
            # LAST: >= 8 s of continuous GPU work for the driver's GPU-busy samples (the timed windows
            # above are ~30 ms in all)
            line["sustained"] = guarded(sustained_leg, gsa, torch, local_rank, math_mode, n_local)
            if "value" in line["sustained"]:
                line["sustained"]["ratio_to_value"] = line["sustained"]["value"] / line["value"]
    elif rank == 0:
        line["cpu_baseline"] = None

    if rank == 0:
        print(json.dumps(line), flush=True)       # the headline is OUT before anything experimental runs
    # ---- N > 1, default transport: the same windows once more over transport "p2p", in child processes,
    # AFTER the line (a hang of the never-cross-device-tested transport must not hold the line
    # back); bounded, children die with their parent; the figures go to stderr and gpurun_out/
    if (world > 1 and args.transport == "rccl" and args.p2p_trial and not args.no_p2p_trial and not args.trial_child):
        trial = guarded(p2p_trial, args, rank, local_rank, world, script)
        if rank == 0:
            sys.stderr.write("bench.py: p2p_trial: " + json.dumps(trial) + "\n")
            sys.stderr.flush()
            try:
                (ROOT / "gpurun_out").mkdir(exist_ok=True)
                (ROOT / "gpurun_out" / f"p2p_trial_N{world}.json").write_text(json.dumps(trial, indent=1))
            except OSError:
                pass
    if distributed:
        try:
            dist.barrier()
            dist.destroy_process_group()
        except Exception as exc:  # noqa: BLE001 - the line is out; a group that cannot shut down is not a failed run
            sys.stderr.write(f"bench.py: process group shutdown: {exc!r}\n")
    return 0


if __name__ == "__main__":
    sys.exit(main())
