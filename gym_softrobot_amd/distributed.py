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

`transport="p2p"` (opt-in; `overlap=True` only) replaces the collective by what it amounts to for
rows this small: right behind its step kernel, ON THE SAME STREAM, every rank launches one small
kernel (`softrod_scatter_rows`) that stores its packed rows into its block of every rank's output
buffer (IPC-mapped device memory; between GPUs the stores travel point to point over xGMI) — no
collective call per step, no second stream, no event.  `sync()` waits for this rank's stream and
meets the other ranks at a barrier, after which everybody's rows have landed.  Why: 4096 rods are
exactly four resident waves on every SIMD, so NOTHING overlaps with a step kernel for free — a
collective (or a copy) on a second stream either waits its turn at ~37 us of cross-queue dependency
latency per step, or runs alongside and stretches the step kernel by as much (kernel
traces) — while a 4 us kernel in order behind the step kernel costs its 4 us.
Ranks are not kept in lockstep by it (a rank may run ahead in an open-loop rollout; a consumer that
reads every step's rows calls sync() every step and is in lockstep through the barrier).  RCCL
stays the default, as BASELINE's north_star asks; `bench.py --transport p2p` measures the other.
OPEN for p2p between different devices (not testable on a 1-GPU box, where both ranks share one L2):
the reader's L2 does not snoop a peer's stores into its HBM, so the output buffers may need to be
fine-grained allocations, or the reader an invalidating read, before this is correct there.
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
                 overlap_depth: Optional[int] = None, transport: str = "rccl"):
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
        if self.overlap:
            n_loc = self.hi - self.lo
            # Two (or `overlap_depth`) alternating sets of output buffers.  Measured on gfx950 (DESIGN.md
            # §4): nothing overlaps with a step kernel of this workload for free — 4096 rods are exactly
            # four resident waves on every SIMD — so a collective on a second stream either follows the
            # kernel it depends on at ~37 us of cross-queue dependency latency per step (what the wait
            # below amounts to: 0.324 against 0.287 ms per step in a world of one) or, left to run
            # alongside the next kernel (a deeper ring, completion queried from the host instead of
            # waited for on the stream), stretches that kernel by more (0.339 ms).  The wait stays.
            if overlap_depth is None:
                overlap_depth = int(os.environ.get("SOFTROD_SHARDED_DEPTH", "2"))
            d = max(2, int(overlap_depth))
            self._packed2 = [torch.empty((n_loc, w), dtype=torch.float32, device=dev) for _ in range(d)]
            self._global2 = [self._global] + [torch.empty_like(self._global) for _ in range(d - 1)]
            self._works = [None] * d
            self._k = 0
        if transport not in ("rccl", "p2p"):
            raise ValueError("transport must be 'rccl' (the group's all-gather) or 'p2p' (peer copies)")
        self.transport = "rccl"
        if transport == "p2p" and self.overlap:
            self._setup_p2p()

    # -- transport="p2p": every rank copies its rows into its block of every peer's buffer --------
    def _setup_p2p(self) -> None:
        """Exchange IPC handles of the output buffers and verify, with a round of test copies, that
        every rank can write every peer's buffer; on any failure ALL ranks stay with the collective."""
        from torch.multiprocessing.reductions import reduce_tensor

        dev = self.local.backend.device
        ok, peers = 1.0, None
        try:
            mine = [reduce_tensor(g) for g in self._global2]
            everyone = [None] * self.world
            dist.all_gather_object(everyone, mine, group=self.group)
            peers = []                    # peers[k][p]: rank p's buffer k as a tensor in THIS process
            for k in range(len(self._global2)):
                row = []
                for p in range(self.world):
                    if p == self.rank:
                        row.append(self._global2[k])
                    else:
                        fn, args = everyone[p][k]
                        row.append(fn(*args))
                peers.append(row)
            # self-test: rank r writes r + 1 into its block of every rank's buffer 0
            probe = torch.full((self.hi - self.lo, self._global.shape[1]), float(self.rank + 1), device=dev)
            self.local.backend.scatter_rows(probe, [t.data_ptr() for t in peers[0]], self.lo)
            torch.cuda.current_stream(dev).synchronize()
        except Exception as exc:  # noqa: BLE001 - any failure means: use the collective
            ok = 0.0
            self._p2p_error = repr(exc)
        flag = torch.tensor([ok], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)     # (also the barrier before the check)
        if float(flag.item()) > 0:
            per = self.hi - self.lo
            expect = torch.arange(1, self.world + 1, device=dev, dtype=torch.float32).repeat_interleave(per)
            good = bool((self._global2[0][:, 0] == expect).all().item())
            flag = torch.tensor([1.0 if good else 0.0], device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        if float(flag.item()) > 0:
            self._peer_global, self.transport = peers, "p2p"
            self._peer_ptrs = [[t.data_ptr() for t in row] for row in peers]

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

    def _step_overlapped(self, a):
        k = self._k
        if self._works[k] is not None:
            # buffers k were last used `depth` steps ago: this orders the kernel below after their
            # gather (a stream-level wait, not a host block on RCCL)
            self._works[k].wait()
        packed, info = self.local.step_packed(a, self._packed2[k])
        if self.transport == "p2p":
            # in order behind the step kernel, on its stream: a few microseconds, no dependency to resolve
            self.local.backend.scatter_rows(packed, self._peer_ptrs[k], self.lo)
        else:
            self._works[k] = dist.all_gather_into_tensor(self._global2[k], packed, group=self.group, async_op=True)
        self._k = (k + 1) % len(self._works)
        o, r, te, tr = unpack_outputs(self._global2[k], self.obs_dim)
        return o, r, te, tr, info

    def sync(self) -> None:
        """Make the outputs of the latest step() complete (overlap=True)."""
        if self.overlap:
            for w in self._works:
                if w is not None:
                    w.wait()
            if self.transport == "p2p":
                torch.cuda.current_stream(self.local.backend.device).synchronize()
                dist.barrier(group=self.group)     # this rank's rows have landed everywhere; now everybody's have

    def close(self):
        self.sync()
        self.local.close()
