"""Multi-GPU sharding of the env batch: one process per GPU, no data-path collective
inside the 400-substep kernel, ONE packed all-gather of per-env outputs per env.step.

The reference has no counterpart (single env, single process); envs never interact
(each `reset` builds a private simulator, soft_pendulum.py:115), so the batch shards
trivially: env i lives on rank i // (N / world).  After each step every rank holds the
outputs of all N envs (what a centralised policy needs).  The collective is
latency-bound (32 B/env), so it is a single `all_gather_into_tensor` on one packed
buffer rather than one collective per output:

    packed[e] = [obs0, obs1, obs2, obs3, reward_lo, reward_hi, terminated, truncated]
                 (8 x 32-bit words; the float64 reward travels bit-exactly as two words)

Backend `nccl` is RCCL over xGMI on the MI355X node; `gloo` drives the same code in
the CPU tests (tests/test_distributed_gloo.py).  `gather=False` leaves outputs sharded
for a data-parallel consumer.
"""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist

PACK_WORDS = 8


def shard_bounds(total_envs: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block of envs owned by `rank` (env i -> rank i*world // total)."""
    if total_envs % world:
        raise ValueError(f"total_envs={total_envs} must be divisible by world_size={world}")
    per = total_envs // world
    return rank * per, (rank + 1) * per


def pack_outputs(obs, reward, terminated, truncated, out=None) -> torch.Tensor:
    """(n,4) f32, (n,) f64, (n,) u8/bool, (n,) u8/bool -> (n, 8) f32 (bit-exact)."""
    n = obs.shape[0]
    if out is None:
        out = torch.empty((n, PACK_WORDS), dtype=torch.float32, device=obs.device)
    out[:, 0:4] = obs
    out[:, 4:6] = reward.contiguous().view(torch.float32).view(n, 2)
    out[:, 6] = terminated.to(torch.float32)
    out[:, 7] = truncated.to(torch.float32)
    return out


def unpack_outputs(packed: torch.Tensor):
    n = packed.shape[0]
    obs = packed[:, 0:4]
    reward = packed[:, 4:6].contiguous().view(torch.float64).view(n)
    terminated = packed[:, 6] != 0
    truncated = packed[:, 7] != 0
    return obs, reward, terminated, truncated


class ShardedVecEnv:
    """Wraps this rank's local vec env (N/world envs) and presents the global batch.

    local_env: object with reset(seed=[...], mask=...) / step(actions) returning torch
    tensors (VecSoftPendulumEnv); it must have been built with num_envs = N / world.
    """

    def __init__(self, local_env, total_envs: int, group: Optional[dist.ProcessGroup] = None,
                 gather: bool = True):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.total_envs = int(total_envs)
        self.lo, self.hi = shard_bounds(self.total_envs, self.world, self.rank)
        self.local = local_env
        if local_env.num_envs != self.hi - self.lo:
            raise ValueError("local env size does not match this rank's shard")
        self.gather = gather
        dev = local_env.backend.device
        self._packed = torch.empty((self.hi - self.lo, PACK_WORDS), dtype=torch.float32, device=dev)
        self._global = torch.empty((self.total_envs, PACK_WORDS), dtype=torch.float32, device=dev)

    def _all_gather(self, packed: torch.Tensor) -> torch.Tensor:
        if self.world == 1:
            return packed
        dist.all_gather_into_tensor(self._global, packed, group=self.group)
        return self._global

    def reset(self, *, seed: Optional[int] = None, mask=None):
        """Global env i is seeded seed + i regardless of the sharding."""
        seeds = None if seed is None else [int(seed) + i for i in range(self.lo, self.hi)]
        m = None if mask is None else np.asarray(mask)[self.lo : self.hi]
        obs, info = self.local.reset(seed=seeds, mask=m)
        if not self.gather or self.world == 1:
            return obs, info
        n = obs.shape[0]
        zeros64 = torch.zeros(n, dtype=torch.float64, device=obs.device)
        zeros8 = torch.zeros(n, dtype=torch.uint8, device=obs.device)
        g = self._all_gather(pack_outputs(obs, zeros64, zeros8, zeros8, self._packed))
        return unpack_outputs(g)[0], info

    def step(self, actions):
        """actions: global (N,) / (N,1) tensor or array, or this rank's shard."""
        a = torch.as_tensor(actions, dtype=torch.float32).reshape(-1)
        if a.numel() == self.total_envs:
            a = a[self.lo : self.hi]
        obs, rew, term, trunc, info = self.local.step(a)
        if not self.gather or self.world == 1:
            return obs, rew, term, trunc, info
        g = self._all_gather(pack_outputs(obs, rew, term, trunc, self._packed))
        o, r, te, tr = unpack_outputs(g)
        return o, r, te, tr, info

    def close(self):
        self.local.close()
