"""Multi-GPU sharding of the env batch: one process per GPU, no data-path collective
inside the 400-substep kernel, ONE packed all-gather of per-env outputs per env.step.

The reference has no counterpart (single env, single process); envs never interact
(each `reset` builds a private simulator, soft_pendulum.py:115), so the batch shards
trivially: env i lives on rank i // (N / world).  After each step every rank holds the
outputs of all N envs (what a centralised policy needs).  The collective is
latency-bound (32 B/env), so it is a single `all_gather_into_tensor` on one packed
buffer rather than one collective per output.  The step kernel writes the packed rows
itself (`softrod_step_packed`), and the receiver unpacks with views only, so a step of
the sharded env is exactly two device operations per rank: the kernel and the all-gather.

    row = [obs (obs_dim float32) | pad to even | reward (float64 as 2 words, 8-byte aligned)
           | terminated, truncated (bytes 0, 1 of one word) | 0]

Backend `nccl` is RCCL over xGMI on the MI355X node; `gloo` drives the same code in
the CPU tests (tests/test_distributed_gloo.py).  `gather=False` leaves outputs sharded
for a data-parallel consumer.

`overlap=True` (what `bench.py --gpus N` uses) takes the collective off the critical path:
the all-gather of step t is issued asynchronously (it starts when step t's kernel has
finished) and step t+1's kernel is launched without waiting for it, on alternating buffers.
The tensors returned by `step` are then complete once `sync()` — or the step after next —
has been called; a policy that needs them immediately calls `sync()` and loses nothing
compared with `overlap=False`.
"""
from __future__ import annotations

import os
from typing import Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


def packed_width(obs_dim: int) -> int:
    """32-bit words per env in the packed row (include/softrod.h, softrod_step_packed)."""
    return obs_dim + (obs_dim & 1) + 4


def shard_bounds(total_envs: int, world: int, rank: int) -> Tuple[int, int]:
    """Contiguous block of envs owned by `rank` (env i -> rank i*world // total)."""
    if total_envs % world:
        raise ValueError(f"total_envs={total_envs} must be divisible by world_size={world}")
    per = total_envs // world
    return rank * per, (rank + 1) * per


def pack_outputs(obs, reward, terminated, truncated, out=None) -> torch.Tensor:
    """Host-side twin of the kernel's packed epilogue (reset observations, test doubles):
    (n,od) f32, (n,) f64, (n,) u8/bool, (n,) u8/bool -> (n, packed_width(od)) f32, bit-exact."""
    n, od = obs.shape
    ro = od + (od & 1)
    if out is None:
        out = torch.zeros((n, ro + 4), dtype=torch.float32, device=obs.device)
    else:
        out.zero_()
    out[:, 0:od] = obs
    out[:, ro : ro + 2] = reward.contiguous().view(torch.float32).view(n, 2)
    flags = out.view(torch.uint8)
    flags[:, 4 * (ro + 2)] = terminated.to(torch.uint8)
    flags[:, 4 * (ro + 2) + 1] = truncated.to(torch.uint8)
    return out


def unpack_outputs(packed: torch.Tensor, obs_dim: int):
    """Views only: no device work."""
    ro = obs_dim + (obs_dim & 1)
    obs = packed[:, 0:obs_dim]
    reward = packed[:, ro : ro + 2].view(torch.float64)[:, 0]
    flags = packed.view(torch.uint8)
    terminated = flags[:, 4 * (ro + 2)].view(torch.bool)
    truncated = flags[:, 4 * (ro + 2) + 1].view(torch.bool)
    return obs, reward, terminated, truncated


class ShardedVecEnv:
    """Wraps this rank's local vec env (N/world envs) and presents the global batch.

    local_env: a VecRodEnvBase built with num_envs = N / world.
    """

    def __init__(self, local_env, total_envs: int, group: Optional[dist.ProcessGroup] = None,
                 gather: bool = True, overlap: bool = False, force_collective: bool = False,
                 overlap_depth: int = 2):
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.total_envs = int(total_envs)
        self.lo, self.hi = shard_bounds(self.total_envs, self.world, self.rank)
        self.local = local_env
        if local_env.num_envs != self.hi - self.lo:
            raise ValueError("local env size does not match this rank's shard")
        self.gather = gather
        # force_collective: run the all-gather also in a world of one (exercises the RCCL path
        # on a single-GPU box: `SOFTROD_BENCH_FORCE_DIST=1 torchrun --nproc-per-node 1 bench.py`)
        self._collective = gather and (self.world > 1 or (force_collective and dist.is_initialized()))
        self.obs_dim = local_env.obs_dim
        dev = local_env.backend.device
        w = packed_width(self.obs_dim)
        self._global = torch.empty((self.total_envs, w), dtype=torch.float32, device=dev)
        self.overlap = bool(overlap) and self._collective
        # The overlapped path launches its step kernels on a stream of its own, created AFTER the
        # collective backend's stream exists (the gather of reset() creates it): ROCm multiplexes
        # HIP streams onto a few hardware queues round-robin, and when the step kernels and the
        # collective share one, the queue executes them strictly in order — kernel, wait, gather,
        # wait, kernel — with ~20 us of dependency latency on either side of the gather instead of
        # overlapping it with the next kernel (measured in a world of one: 0.324 ms per step against
        # 0.287 for the kernel alone).
        self._compute_stream = None
        if self.overlap:
            n_loc = self.hi - self.lo
            d = max(2, int(overlap_depth))
            self._packed2 = [torch.empty((n_loc, w), dtype=torch.float32, device=dev) for _ in range(d)]
            self._global2 = [self._global] + [torch.empty_like(self._global) for _ in range(d - 1)]
            self._works = [None] * d
            self._k = 0

    def _all_gather(self, packed: torch.Tensor) -> torch.Tensor:
        dist.all_gather_into_tensor(self._global, packed, group=self.group)
        return self._global

    def reset(self, *, seed: Optional[int] = None, mask=None):
        """Global env i is seeded seed + i regardless of the sharding."""
        if self.overlap:      # no all-gather of an earlier step may still be writing the buffers
            self.sync()
            self._works = [None] * len(self._works)
            self._k = 0
        seeds = None if seed is None else [int(seed) + i for i in range(self.lo, self.hi)]
        m = None if mask is None else np.asarray(mask)[self.lo : self.hi]
        obs, info = self.local.reset(seed=seeds, mask=m)
        if not self._collective:
            return obs, info
        n = obs.shape[0]
        zeros64 = torch.zeros(n, dtype=torch.float64, device=obs.device)
        zeros8 = torch.zeros(n, dtype=torch.uint8, device=obs.device)
        g = self._all_gather(pack_outputs(obs, zeros64, zeros8, zeros8))
        if self._compute_stream is not None:       # the next steps run after this reset
            self._compute_stream.wait_stream(torch.cuda.current_stream(obs.device))
        return unpack_outputs(g, self.obs_dim)[0], info

    def step(self, actions):
        """actions: global (N, action_dim) tensor or array, or this rank's shard."""
        adim = self.local.action_dim
        a = torch.as_tensor(actions, dtype=torch.float32).reshape(-1, adim)
        if a.shape[0] == self.total_envs and self.world > 1:
            a = a[self.lo : self.hi]
        if not self._collective:
            return self.local.step(a)
        if self.overlap:
            return self._step_overlapped(a)
        packed, info = self.local.step_packed(a)
        o, r, te, tr = unpack_outputs(self._all_gather(packed), self.obs_dim)
        return o, r, te, tr, info

    def _own_stream(self):
        dev = self.local.backend.device
        if dev.type != "cuda" or os.environ.get("SOFTROD_SHARDED_OWN_STREAM", "1") == "0":
            return None
        if self._compute_stream is None:
            self._compute_stream = torch.cuda.Stream(device=dev)
            self._compute_stream.wait_stream(torch.cuda.current_stream(dev))     # reset / earlier steps first
        return self._compute_stream

    def _step_overlapped(self, a):
        st = self._own_stream()
        if st is None:
            return self._step_overlapped_on_current_stream(a)
        dev = self.local.backend.device
        a = a.to(dev)
        st.wait_stream(torch.cuda.current_stream(dev))     # whatever produced the actions
        a.record_stream(st)
        with torch.cuda.stream(st):
            return self._step_overlapped_on_current_stream(a)

    def _step_overlapped_on_current_stream(self, a):
        k = self._k
        if self._works[k] is not None:
            # buffers k were last used two steps ago: their gather has long finished; this only
            # orders the kernel below after it (a stream-level wait, not a host block on RCCL)
            self._works[k].wait()
        packed, info = self.local.step_packed(a, self._packed2[k])
        self._works[k] = dist.all_gather_into_tensor(self._global2[k], packed, group=self.group, async_op=True)
        self._k = (k + 1) % len(self._works)
        o, r, te, tr = unpack_outputs(self._global2[k], self.obs_dim)
        return o, r, te, tr, info

    def sync(self) -> None:
        """Make the outputs of the latest step() complete (overlap=True)."""
        if self.overlap:
            if self._compute_stream is not None:
                # the caller's stream continues after everything launched on the private one ...
                torch.cuda.current_stream(self.local.backend.device).wait_stream(self._compute_stream)
            for w in self._works:
                if w is not None:
                    w.wait()       # ... and after the gathers

    def close(self):
        self.sync()
        self.local.close()
