"""One rank of tests/test_gpu_two_ranks.py::test_sharded_env_two_hip_ranks_* — TEST INFRASTRUCTURE.

Launched by `python -m torch.distributed.run --nproc-per-node 2` on the 1-GPU box: BOTH ranks open
cuda:0 (RCCL refuses two ranks on one device, so the packed rows travel over gloo), step their shard
with the real HIP kernels through ShardedVecEnv(overlap=True), take a masked reset in the middle of
the rollout, and rank 0 compares every gathered step bit for bit with ONE process stepping the whole
batch on the same GPU.  argv: env-id envs-per-rank steps-before steps-after [autoreset [transport]]"""
import os
import sys
from pathlib import Path

import numpy as np
import torch
import torch.distributed as dist

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

import gym_softrobot_amd as gsa  # noqa: E402
from gym_softrobot_amd.distributed import ShardedVecEnv  # noqa: E402

AMAX = {"SoftPendulum-v0": 22.0, "SoftPendulum3D-v0": 1.0, "OctoArmSingle-v0": 6.0, "OctoFlat-v0": 22.0}


def rollout(env, total, adim, amax, t1, t2, sharded):
    acts = np.random.default_rng(5).uniform(-amax, amax, (t1 + t2, total, adim)).astype(np.float32)
    mask = np.random.default_rng(6).random(total) < 0.4
    mask[[0, total // 2 - 1, total // 2, total - 1]] = [True, False, True, True]    # both sides of the shard boundary
    out = []
    obs0, _ = env.reset(seed=7)
    out.append(torch.as_tensor(obs0).cpu().clone().numpy())
    for t in range(t1 + t2):
        if t == t1:
            o, _ = env.reset(mask=mask)          # masked reset mid-rollout: the envs' NEXT draws, no re-seed
            out.append(torch.as_tensor(o).cpu().clone().numpy())
        o, r, te, tr, _ = env.step(acts[t])
        if sharded:
            env.sync()
        out.append(tuple(torch.as_tensor(x).cpu().clone().numpy() for x in (o, r, te, tr)))
    return out


def main():
    env_id, per, t1, t2 = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
    autoreset = sys.argv[5] if len(sys.argv) > 5 else "off"
    transport = sys.argv[6] if len(sys.argv) > 6 else "rccl"
    kw = {} if autoreset == "off" else {"autoreset": autoreset}
    short = {}
    if autoreset != "off":      # 3-step episodes, so restarts cross the shard boundary many times
        short = dict(final_time=3 * 400e-4 - 1e-9) if env_id == "SoftPendulum-v0" else {}
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)                                   # every rank on the one GPU
    dist.init_process_group("gloo")
    total = per * world
    local = gsa.make_vec(env_id, per, device=0, **kw, **short)
    env = ShardedVecEnv(local, total, overlap=True, transport=transport)
    assert env.transport == transport, getattr(env, "_p2p_error", "transport fell back")
    got = rollout(env, total, local.action_dim, AMAX[env_id], t1, t2, True)
    ok = True
    if rank == 0:
        ref_kw = {} if autoreset == "off" else {"autoreset": True}    # host-driven NEXT_STEP: the same draws
        one = gsa.make_vec(env_id, total, device=0, **ref_kw, **short)
        ref = rollout(one, total, one.action_dim, AMAX[env_id], t1, t2, False)
        assert len(ref) == len(got)
        for k, (g, r) in enumerate(zip(got, ref)):
            g, r = (g if isinstance(g, tuple) else (g,)), (r if isinstance(r, tuple) else (r,))
            for a, b in zip(g, r):
                if not np.array_equal(np.asarray(a), np.asarray(b), equal_nan=True):
                    ok = False
                    print(f"MISMATCH at record {k}: max |d| = {np.nanmax(np.abs(np.asarray(a, float) - np.asarray(b, float)))}")
        restarted = sum(int(np.asarray(x[3]).sum() + np.asarray(x[2]).sum()) for x in ref if isinstance(x, tuple))
        print(f"TWO-RANK-{'OK' if ok else 'FAIL'} env={env_id} total={total} records={len(got)} flagged={restarted}")
        one.close()
    dist.barrier()
    env.close()
    dist.destroy_process_group()
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
