#!/usr/bin/env python3
"""bench.py — env-steps/s of N parallel SoftPendulum-v0 on MI355X (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one env.step() of every env of the job: 4096 envs x 50 elements per GPU
(BASELINE.json configs[1]; with N GPUs the batch is 4096*N envs sharded contiguously,
configs[3] at N=8 — weak scaling), i.e. per GPU 4096 x 400 PositionVerlet substeps in
one kernel launch, plus (N>1) one packed RCCL all-gather of the per-env outputs.
Inputs (state, pre-staged float32 actions) are resident in HBM when the timed region
starts.  Rank 0 prints ONE JSON line.

roofline: algorithmic bytes per launch = rods x substeps x 2*(18n+6)*8 B (SURVEY.md
§8(d): every substep reads+writes x, v, Q, omega once) divided by the step kernel's
average duration, measured with HIP events recorded on the launch stream around every
timed launch (softrod_set_timing / softrod_kernel_times_ms).  The kernel is
register-resident (HBM is touched once per env.step), so measured `traffic` is far below
the algorithmic figure — see DESIGN.md "roofline".

cpu_baseline: the repo's fp64 C oracle (a port/restatement, NOT PyElastica — see
oracle/softrod_oracle.c) timed on this box's host cores with OpenMP over rods, rank 0,
N=1 only, on a bounded sample of the same workload.
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

# SQ_INSTS_VALU per rod-substep measured with rocprofv3 --pmc (profiles/README.md), by (env, n_elem)
VALU_PER_ROD_SUBSTEP = {("SoftPendulum-v0", 50): 99.5, ("SoftPendulum3D-v0", 50): 596.0,
                        ("OctoArmSingle-v0", 50): 597.0, ("SoftArmTracking-v0", 40): 349.0}
ENVS_PER_GPU = 4096
N_ELEM = 50
HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8 TB/s


def algorithmic_bytes_per_rod_substep(n_elem: int, sizeof_real: int = 8) -> int:
    return 2 * (18 * n_elem + 6) * sizeof_real  # 14 496 B for n = 50, fp64


def cpu_baseline(cfg, cores: int, budget_s: float = 12.0):
    """Time the C oracle (OpenMP over rods) on a bounded sample: 16 rods per core,
    env.steps until ~budget_s of wall time (at least 2 steps)."""
    import numpy as np

    os.environ["OMP_NUM_THREADS"] = str(cores)
    from gym_softrobot_amd.seeding import initial_angle, np_random
    from oracle import oracle_c

    oracle_c.build()
    n_rods = 16 * cores
    batch = oracle_c.OracleBatch(cfg, n_rods, omp=True)
    batch.reset([initial_angle(np_random(i)[0]) for i in range(n_rods)])
    acts = np.random.default_rng(1).uniform(-22, 22, (64, n_rods)).astype(np.float32)
    batch.env_step(acts[0])  # warm-up
    t0 = time.perf_counter()
    steps = 0
    while steps < 2 or (time.perf_counter() - t0 < budget_s and steps < 60):
        batch.env_step(acts[1 + steps])
        steps += 1
    dt = time.perf_counter() - t0
    # one rod on one thread, for reading the threaded figure (cgroup quotas and SMT siblings make
    # "cores" an upper bound of what the box really gives the process)
    one = oracle_c.OracleRod(cfg)
    one.reset_pendulum(initial_angle(np_random(0)[0]))
    one.env_step(float(acts[0, 0]))
    t1 = time.perf_counter()
    k = 0
    while time.perf_counter() - t1 < 1.0:
        one.env_step(float(acts[1 + k % 60, 0]))
        k += 1
    dt1 = time.perf_counter() - t1
    return {
        "value": n_rods * steps / dt,
        "unit": "env-steps/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{n_rods} rods x {steps} env.steps (400 substeps, 50 elements, fp64 C oracle, "
                  f"OpenMP {cores} threads, {dt:.1f} s)",
        "single_thread_value": k / dt1,
    }


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: 20 + 100 env.steps stay inside one SoftPendulum episode (truncation fires on step
    # 126), and the warm-up covers the ~20 launches the GPU clock takes to settle after the reset
    # (profiles/README.md r1f: 0.38 ms per launch at first, 0.34 ms from then on)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs-per-gpu", type=int, default=None,
                    help=f"default {ENVS_PER_GPU}; OctoFlat-v0: 1024 (BASELINE configs[4]: 8192 envs on 8 GPUs)")
    ap.add_argument("--math-mode", choices=["fast", "libm"], default="fast")
    ap.add_argument("--env", default="SoftPendulum-v0",
                    choices=["SoftPendulum-v0", "SoftPendulum3D-v0", "OctoArmSingle-v0", "OctoFlat-v0", "SoftArmTracking-v0"],
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
    ap.add_argument("--n-elems", type=int, default=None, help="elements per rod (env default if omitted)")
    args = ap.parse_args()

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
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch N>1 with torch.distributed.run --nproc-per-node N (one rank per GPU)")
        raise SystemExit(f"--gpus {args.gpus} != WORLD_SIZE {world}")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a MI355X: the hot path has no CPU fallback")
    if os.environ.get("SOFTROD_BENCH_ALL_RANKS_ON_DEVICE0") == "1":
        local_rank = 0   # smoke-testing the N>1 code path on a 1-GPU box (not a measurement)
    torch.cuda.set_device(local_rank)
    force_dist = os.environ.get("SOFTROD_BENCH_FORCE_DIST") == "1"   # RCCL smoke test in a world of one
    if world > 1 or force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29511")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        backend = os.environ.get("SOFTROD_BENCH_DIST_BACKEND", "nccl")   # "gloo": 1-GPU smoke test only
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    n_local = args.envs_per_gpu or (1024 if args.env == "OctoFlat-v0" else ENVS_PER_GPU)
    if args.scaling == "strong":
        if n_local % world:
            raise SystemExit(f"--scaling strong: {n_local} envs do not split over {world} GPUs")
        n_local //= world
    n_total = n_local * world
    K, W = args.steps, args.warmup
    math_mode = _capi.MATH_FAST if args.math_mode == "fast" else _capi.MATH_LIBM
    extra = {} if args.n_elems is None else {"n_elems": args.n_elems}
    if args.autoreset == "auto":
        args.autoreset = "off" if (args.steps + args.warmup <= 120 or args.env != "SoftPendulum-v0") else "device"
    if args.autoreset != "off":
        extra["autoreset"] = True if args.autoreset == "host" else "device"
    local = gsa.make_vec(args.env, n_local, device=local_rank, math_mode=math_mode, **extra)
    # world > 1: kernel-packed rows + one all-gather per step, issued asynchronously so that the
    # next step's kernel does not wait for it (ShardedVecEnv overlap; the final sync is timed)
    env = ShardedVecEnv(local, n_total, overlap=True, force_collective=force_dist)
    env.reset(seed=0)                      # global env i seeded i
    lo, hi = env.lo, env.hi
    adim = local.backend.action_dim
    amax = {"SoftPendulum-v0": 22.0, "SoftPendulum3D-v0": 1.0, "OctoArmSingle-v0": 6.0,
            "OctoFlat-v0": 22.0, "SoftArmTracking-v0": 1.0}[args.env]
    # the truncation flag of SoftPendulum first fires on env.step #126; the default window
    # (120 steps) stays inside one episode
    T = W + K
    acts = np.random.default_rng(1).uniform(-amax, amax, (T, n_total, adim)).astype(np.float32)
    acts_dev = torch.from_numpy(acts[:, lo:hi].copy()).to(local.backend.device)

    for t in range(W):
        env.step(acts_dev[t])
    local.backend.set_timing(K)
    restarts_before = int(local.backend.queue_status()[0].sum()) if args.autoreset == "device" else 0
    if world > 1 or force_dist:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in range(W, T):
        obs, rew, term, trunc, _ = env.step(acts_dev[t])
    env.sync()
    torch.cuda.synchronize()
    if world > 1 or force_dist:
        dist.barrier()
    elapsed = time.perf_counter() - t0
    if world > 1 or force_dist:
        el = torch.tensor([elapsed], dtype=torch.float64, device=local.backend.device)
        dist.all_reduce(el, op=dist.ReduceOp.MAX)
        elapsed = float(el.item())

    kt = local.backend.kernel_times_ms()
    assert len(kt) == K or args.autoreset != "off"
    # env-steps that restarted an episode instead of integrating are not counted as work
    restarts = 0
    if args.autoreset == "device":
        restarts = int(local.backend.queue_status()[0].sum()) - restarts_before
    if world > 1 or force_dist:
        rs = torch.tensor([restarts], dtype=torch.int64, device=local.backend.device)
        dist.all_reduce(rs)
        restarts = int(rs.item())
    n_bad = int((~torch.isfinite(obs).all(dim=1)).sum().item())
    # what the last step returned, over ALL envs (gathered rows included): lets two runs be compared
    obs_checksum = float(torch.nan_to_num(obs.double()).sum().item()) + float(torch.nan_to_num(rew.double()).sum().item())

    if rank == 0:
        cfg = local.cfg
        nsub = int(cfg.n_substeps)
        octo = args.env == "OctoFlat-v0"
        rods_per_env = int(cfg.n_arm) if octo else 1
        # OctoFlat: n_arm rods + the rigid head (x, v, Q, w = 18 doubles read and written)
        bytes_per_launch = n_local * nsub * (rods_per_env * algorithmic_bytes_per_rod_substep(int(cfg.n_elem))
                                             + (2 * 18 * 8 if octo else 0))
        kernel_ms = float(np.mean(kt))
        achieved = bytes_per_launch / (kernel_ms * 1e-3) / 1e9
        traffic = None
        tf = ROOT / "profiles" / "hbm_traffic.json"
        if tf.exists():  # measured with rocprofv3 --pmc (separate passes), see profiles/README.md
            try:   # keyed by workload: only a measurement of THIS env / size / batch is reported
                key = f"{args.env}|n_elem={int(cfg.n_elem)}|envs={n_local}"
                traffic = json.loads(tf.read_text()).get(key, {}).get("hbm_bytes_per_launch")
            except Exception:  # noqa: BLE001
                traffic = None
        line = {
            "metric": "env_steps_per_sec",
            "value": (n_total * K - restarts) / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": K,
            "warmup": W,
            "ms_per_step": elapsed / K * 1e3,
            "higher_is_better": True,
            "scaling": args.scaling,
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": f"{args.env}, {n_local} envs x "
                            + (f"{rods_per_env} arms x " if octo else "") + f"{int(cfg.n_elem)} elements per GPU "
                            + (f"(BASELINE configs[1]; x{world} GPUs)" if args.env == "SoftPendulum-v0"
                               else "(widened row of SURVEY §8; not the headline metric)"),
                "envs_total": n_total,
                "substeps_per_env_step": nsub,
                "math_mode": args.math_mode,
                "autoreset": args.autoreset,
                "episode_restarts_not_counted": restarts,
                "sharding": "contiguous env blocks per rank; one packed all_gather per step" if world > 1 else "single GPU",
                "rod_substeps_per_sec": (n_total * K - restarts) * rods_per_env * nsub / elapsed,
                "non_finite_envs_at_end": n_bad, "last_step_checksum": obs_checksum,
            },
            "roofline": {
                "bound": "hbm",
                "achieved": achieved,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                "kernel": "softrod_octo_step_kernel" if octo else
                          "softrod_step_window_kernel + softrod_step_fast_kernel (epilogue only)"
                          if (args.env == "OctoArmSingle-v0" and 64 <= int(cfg.n_elem) <= 102
                              and args.math_mode == "fast") else
                          ("softrod_step_fast_kernel" if args.math_mode == "fast" else "softrod_step_libm_kernel"),
                "fp64_valu": {
                    "note": "the binding unit: wave64 fp64 VALU ops issue in 4 cycles (78.6 TFLOP/s); "
                            "SIMD-cycles per rod-substep below vs ~4 x fp64 instruction count (profiles/README.md)",
                    "simd_cycles_per_rod_substep_at_2.4GHz":
                        kernel_ms * 1e-3 * 2.4e9 * 1024 / (n_local * rods_per_env * nsub),
                    # measured SQ_INSTS_VALU per rod-substep of this workload's kernel (profiles/README.md;
                    # tools/pmc_valu_per_substep.sh), and the share of the VALU issue slots they fill
                    # at the nominal 2.4 GHz — the roofline of the unit that actually bounds the kernel
                    "valu_instructions_per_rod_substep": VALU_PER_ROD_SUBSTEP.get((args.env, int(cfg.n_elem))),
                    "valu_issue_frac_at_2.4GHz":
                        None if (args.env, int(cfg.n_elem)) not in VALU_PER_ROD_SUBSTEP else
                        4.0 * VALU_PER_ROD_SUBSTEP[(args.env, int(cfg.n_elem))]
                        / (kernel_ms * 1e-3 * 2.4e9 * 1024 / (n_local * rods_per_env * nsub)),
                },
                "kernel_ms_avg": kernel_ms,
                "algorithmic_bytes_per_launch": bytes_per_launch,
                "note": "algorithmic bytes = rods x substeps x 2(18n+6) x 8 B (SURVEY 8d); the kernel keeps "
                        "the state in registers for all substeps, so real HBM traffic is ~1/400 of that",
            },
        }
        if world == 1 and not args.no_cpu_baseline and args.env == "SoftPendulum-v0":
            cores = len(os.sched_getaffinity(0))
            line["cpu_baseline"] = cpu_baseline(cfg, cores)
        else:
            line["cpu_baseline"] = None
        print(json.dumps(line), flush=True)

    env.close()
    if world > 1 or force_dist:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
