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
for a data-parallel consumer.  `init_process_group()` below creates the RCCL group with a
HIGH-PRIORITY collective stream: 4096 rods are exactly four resident waves on every SIMD, so an
all-gather kernel on a normal-priority stream queues behind the next step kernel's waves.

`overlap=True` (what `bench.py --gpus N` uses) takes the collective off the critical path:
the all-gather of step t is issued asynchronously (it starts when step t's kernel has
finished) and step t+1's kernel is launched without waiting for it, on alternating buffers.
With transport "rccl" the tensors returned by `step` are complete once `sync()` — or the step
after next — has been called; a policy that needs them immediately calls `sync()` and loses
nothing compared with `overlap=False`.

`transport="p2p"` (opt-in, EXPERIMENTAL until it has run between different devices; needs
`overlap=True`) replaces the collective by what it amounts to for rows this small: right behind its
step kernel, ON THE SAME STREAM, every rank launches one small kernel (`softrod_scatter_rows`) that
stores its packed rows into its block of every rank's exchange buffer (IPC-mapped device memory;
between GPUs the stores travel point to point over xGMI) — no collective call per step, no second
stream, no event.  Why: nothing overlaps with a step kernel of this workload for free — a collective
(or a copy) on a second stream either waits its turn at ~37 us of cross-queue dependency latency per
step, or runs alongside and stretches the step kernel by as much (kernel traces) —
while a 4 us kernel in order behind the step kernel costs its 4 us.  Correct by construction:
  * the exchange buffers are UNCACHED device allocations (`softrod_exchange_alloc`:
    hipDeviceMallocUncached, fine-grained where that is refused), never ordinary torch tensors — a
    GPU's L2 does not snoop a peer's stores into its HBM, and an uncached buffer has no line there to
    go stale; peer access is enabled explicitly before a peer's handle is opened;
  * the rows are stored system-scope write-through, and every call ends by storing a GENERATION word
    per source rank behind the rows (system-scope release) once all of its rows have been acknowledged;
  * `sync()` waits for this rank's stream, meets the other ranks at a barrier and then CHECKS the
    generation words of the latest step from every rank (raises on a mismatch);
  * the set-up self-test runs several rounds over the SAME buffers with a CHANGING pattern, so a
    stale read of an earlier round fails it, and on any failure — on any rank, at any point of the
    set-up, which every rank walks through to the end — ALL ranks stay with the collective.
Completion contract of "p2p" (differs from "rccl"): the rows `step` returns are complete only after
`sync()`, which all ranks must call at the same step indices (SPMD).  Buffer k = t mod `depth` is written
again by every rank at step t + depth, and a peer that has passed the barrier of a `sync()` runs ahead up to
its next `sync()`.  With a `sync()` every s steps, after the one at step T the peers may already be writing
steps T+1 .. T+s while this rank reads:
  * the rows of the LATEST step only (step T) stay intact if s <= depth - 1;
  * the rows of EVERY step since the previous sync (T-s+1 .. T) stay intact only if 2 s <= depth
    (`p2p_sync_interval()`; the default depth of 2 means: sync after every step).
`step` counts the steps since the last `sync()` and raises `P2PError` beyond floor(depth / 2) unless
`p2p_enforce_sync_interval` is set to False by a caller that reads nothing between syncs (`bench.py`'s
timed windows read the last step's rows once, after the closing sync).  `sync()` checks the generation
words of EVERY step launched since the previous sync whose buffer has not been reused since.  Read the rows
on the env's stream (or finish reading) before the next `sync()`.  RCCL stays the default, as BASELINE's
north_star asks; `bench.py --transport p2p` measures the other.
"""
from __future__ import annotations

import os
import warnings
from typing import Optional, Tuple

import numpy as np
import torch
import torch.distributed as dist


def init_process_group(backend: str = "nccl", device: Optional[torch.device] = None,
                       high_priority: Optional[bool] = None, **kw) -> None:
    """`dist.init_process_group` with, for RCCL ("nccl"), the collective stream created at HIGH
    priority (ProcessGroupNCCL.Options.is_high_priority_stream) unless SOFTROD_RCCL_HIGH_PRIORITY=0:
    the step kernels keep every SIMD's wave slots full, and the all-gather's few workgroups should be
    dispatched ahead of the next step kernel's, not behind them."""
    if high_priority is None:
        high_priority = os.environ.get("SOFTROD_RCCL_HIGH_PRIORITY", "1") != "0"
    if backend == "nccl":
        if device is not None:
            kw["device_id"] = device
        if high_priority and hasattr(dist, "ProcessGroupNCCL"):
            try:
                opts = dist.ProcessGroupNCCL.Options()
                opts.is_high_priority_stream = True
                dist.init_process_group(backend, pg_options=opts, **kw)
                return
            except Exception as exc:  # noqa: BLE001 - ANY failure of the options path (a torch build without
                # pg_options, an Options object the backend refuses, ...) must not abort the job: the
                # priority is an optimisation (+15 us per step instead of +30), the group is not
                warnings.warn(f"RCCL group at normal stream priority: the high-priority path failed with {exc!r}",
                              RuntimeWarning, stacklevel=2)
                if dist.is_initialized():      # a half-made default group would make the retry fail
                    try:
                        dist.destroy_process_group()
                    except Exception:  # noqa: BLE001
                        pass
    dist.init_process_group(backend, **kw)


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


class P2PError(RuntimeError):
    """transport="p2p": a generation word did not hold the step every rank should have written."""


class ShardedVecEnv:
    """Wraps this rank's local vec env (N/world envs) and presents the global batch.

    local_env: a VecRodEnvBase built with num_envs = N / world.
    """

    SELF_TEST_ROUNDS = 5          # p2p set-up: rounds over the same buffers, a new pattern each

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
        self._p2p_error: Optional[str] = None
        self._exchange = None          # p2p: this rank's exchange buffers and the peers' mappings
        if transport == "p2p":
            if self.overlap:
                self._setup_p2p()
            else:
                self._p2p_error = "transport='p2p' needs overlap=True and a process group of more than one rank"
                warnings.warn(self._p2p_error + ": using the collective", RuntimeWarning, stacklevel=2)

    # -- transport="p2p": every rank copies its rows into its block of every peer's buffer --------
    def _agree(self, ok: bool) -> bool:
        """True only when `ok` on EVERY rank (one small all-reduce; also a barrier)."""
        dev = self.local.backend.device
        flag = torch.tensor([1.0 if ok else 0.0], device=dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=self.group)
        return bool(flag.item() > 0)

    def _setup_p2p(self) -> None:
        """Allocate the exchange buffers, exchange their IPC handles, map the peers' and verify with
        SELF_TEST_ROUNDS rounds of test copies under a changing pattern that every rank reads what
        every peer wrote.  Every rank walks through EVERY collective call of this function whatever
        happened to it locally (a failure becomes a flag, never an early exit), so an asymmetric
        failure cannot leave the others blocked in a collective; on any failure anywhere ALL ranks
        stay with the all-gather and release what they had set up."""
        be = self.local.backend
        depth, w = len(self._global2), self._global.shape[1]
        rows_words = self.total_envs * w
        n_words = rows_words + self.world + (-(rows_words + self.world)) % 4     # + one generation word per rank
        ex = {"tensors": [], "ptrs": [], "handles": [], "kind": None, "peer_ptrs": None, "opened": []}
        ok, err = True, None
        try:                                   # (1) local allocations
            for _ in range(depth):
                t, ptr, handle, kind = be.exchange_alloc(n_words)
                ex["tensors"].append(t)
                ex["ptrs"].append(ptr)
                ex["handles"].append(handle)
                ex["kind"] = kind
        except Exception as exc:  # noqa: BLE001 - any failure means: use the collective
            ok, err = False, f"exchange_alloc: {exc!r}"
        mine = {"ok": ok, "handles": ex["handles"] if ok else None, "device": getattr(be, "device_index", -1),
                "pid": os.getpid()}
        everyone = [None] * self.world         # (2) always entered
        dist.all_gather_object(everyone, mine, group=self.group)
        ok = all(e is not None and e["ok"] for e in everyone)
        if ok:
            try:                               # (3) map the peers' buffers
                peer_ptrs = []
                for k in range(depth):
                    row = []
                    for p in range(self.world):
                        if p == self.rank:
                            row.append(ex["ptrs"][k])
                        else:
                            q = be.exchange_open(everyone[p]["handles"][k], everyone[p]["device"])
                            ex["opened"].append(q)
                            row.append(q)
                    peer_ptrs.append(row)
                ex["peer_ptrs"] = peer_ptrs
            except Exception as exc:  # noqa: BLE001
                ok, err = False, f"exchange_open: {exc!r}"
        ok = self._agree(ok)
        # (4) self-test: round j writes the pattern of round j into buffer j % depth; a buffer is
        # written in several rounds, so a reader served a stale copy of an earlier round fails
        per = self.hi - self.lo
        dev = be.device
        for j in range(self.SELF_TEST_ROUNDS if ok else 0):
            good = True
            k = j % depth
            try:                               # local work only inside the try blocks: the barrier and the
                probe = torch.full((per, w), float(1000 * j + self.rank + 1), device=dev)      # agreement below
                be.scatter_rows(probe, ex["peer_ptrs"][k], self.lo, rows_words + self.rank, 0x5E1F0000 + j)       # are reached
                torch.cuda.current_stream(dev).synchronize()                                                     # by every rank
            except Exception as exc:  # noqa: BLE001
                good, err = False, f"self-test round {j}: {exc!r}"
            dist.barrier(group=self.group)     # everybody's stores of this round have been issued and waited for
            if good:
                try:
                    rows = ex["tensors"][k][:rows_words].view(self.total_envs, w)
                    expect = (1000.0 * j + torch.arange(1, self.world + 1, device=dev, dtype=torch.float32)
                              ).repeat_interleave(per)
                    tags = ex["tensors"][k][rows_words:rows_words + self.world].view(torch.int32)
                    good = bool((rows == expect[:, None]).all().item()) and bool((tags == 0x5E1F0000 + j).all().item())
                    if not good:
                        err = f"self-test round {j}: a peer's rows or generation word did not arrive"
                except Exception as exc:  # noqa: BLE001
                    good, err = False, f"self-test round {j}: {exc!r}"
            ok = self._agree(good)
            if not ok:
                break
        if ok:
            self._exchange = ex
            self._rows_words = rows_words
            self._global2 = [t[:rows_words].view(self.total_envs, w) for t in ex["tensors"]]
            self._tags = [t[rows_words:rows_words + self.world].view(torch.int32) for t in ex["tensors"]]
            self._global = self._global2[0]
            self._gen = 0                      # generation of the next step
            self._gen_of = [None] * depth      # generation last written into buffer k
            self._last_k = None
            self._since_sync = []              # buffers written since the last sync(), oldest first
            self.p2p_enforce_sync_interval = True
            self._verify = os.environ.get("SOFTROD_P2P_VERIFY", "1") != "0"
            self.transport = "p2p"
            self.exchange_memory = ex["kind"]
        else:
            self._p2p_error = err or "a peer could not set up its exchange buffers"
            self._release_exchange(ex)

    def _release_exchange(self, ex) -> None:
        """Three phases: every rank unmaps the peers' buffers; all ranks meet; only then does anybody free
        its own (freeing exported memory a peer still has mapped — or is still storing self-test rows
        into — is undefined behaviour for HIP IPC).  Every rank calls this at the same point of the
        set-up / of close(), so the meeting is a collective every rank reaches."""
        be = self.local.backend
        for q in ex.get("opened", []):
            try:
                be.exchange_close(q)
            except Exception:  # noqa: BLE001
                pass
        ex["opened"] = []
        try:
            torch.cuda.current_stream(be.device).synchronize()       # this rank's own stores into peers have drained
        except Exception:  # noqa: BLE001
            pass
        if dist.is_initialized() and self.world > 1:
            try:
                dist.barrier(group=self.group)
            except Exception as exc:  # noqa: BLE001 - a dead group: the processes are going down anyway
                warnings.warn(f"p2p release: barrier failed ({exc!r}); freeing the exchange buffers regardless",
                              RuntimeWarning, stacklevel=2)
        for ptr in ex.get("ptrs", []):
            try:
                be.exchange_free(ptr)
            except Exception:  # noqa: BLE001
                pass
        ex["tensors"], ex["ptrs"] = [], []

    def _all_gather(self, packed: torch.Tensor) -> torch.Tensor:
        if self.transport == "p2p":            # resets: through the group, into ordinary memory
            if getattr(self, "_reset_global", None) is None:
                self._reset_global = torch.empty(tuple(self._global.shape), dtype=torch.float32, device=packed.device)
            dist.all_gather_into_tensor(self._reset_global, packed, group=self.group)
            return self._reset_global
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
        if (self.transport == "p2p" and self.p2p_enforce_sync_interval
                and len(self._since_sync) >= self.p2p_sync_interval()):       # refused BEFORE anything is stepped
            raise P2PError(f"transport p2p with {len(self._works)} buffer sets: sync() at least every "
                           f"{self.p2p_sync_interval()} step(s), or a peer overwrites rows this rank has not "
                           "read yet (set p2p_enforce_sync_interval = False only if nothing is read between syncs)")
        packed, info = self.local.step_packed(a, self._packed2[k])
        if self.transport == "p2p":
            # in order behind the step kernel, on its stream: a few microseconds, no dependency to
            # resolve; the generation word of this rank follows the rows into every peer's buffer
            self._gen = (self._gen + 1) & 0x7FFFFFFF
            self.local.backend.scatter_rows(packed, self._exchange["peer_ptrs"][k], self.lo,
                                            self._rows_words + self.rank, self._gen)
            self._gen_of[k] = self._gen
            self._last_k = k
            self._since_sync = [b for b in self._since_sync if b != k] + [k]     # a reused buffer holds the newer step only
        else:
            self._works[k] = dist.all_gather_into_tensor(self._global2[k], packed, group=self.group, async_op=True)
        self._k = (k + 1) % len(self._works)
        o, r, te, tr = unpack_outputs(self._global2[k], self.obs_dim)
        return o, r, te, tr, info

    def p2p_sync_interval(self, every_step: bool = True) -> int:
        """Largest number of steps between two sync() calls that keeps rows intact under transport
        "p2p": those of every step since the previous sync (2 s <= depth), or of the latest step only
        (s <= depth - 1).  See the module docstring."""
        depth = len(self._works) if self.overlap else 1
        return max(1, depth // 2) if every_step else max(1, depth - 1)

    def sync(self) -> None:
        """Make the outputs of the latest step() complete (overlap=True).  transport "p2p": a stream
        synchronise, a barrier of the group (all ranks call sync() at the same step indices) and a
        check of the generation words, from every rank, of every step launched since the previous
        sync() whose buffer has not been reused since, and a second barrier behind that check (no rank
        starts overwriting buffers while another still reads their tags)."""
        if self.overlap:
            for w in self._works:
                if w is not None:
                    w.wait()
            if self.transport == "p2p":
                torch.cuda.current_stream(self.local.backend.device).synchronize()
                dist.barrier(group=self.group)     # this rank's rows have landed everywhere; now everybody's have
                written, self._since_sync = self._since_sync, []
                if not written and self._last_k is not None:
                    written = [self._last_k]       # nothing new: the latest step's rows are checked again
                bad = None
                if self._verify:
                    for k in written:
                        tags = self._tags[k].cpu()
                        want = self._gen_of[k]
                        if bad is None and not bool((tags == want).all()):
                            bad = (tags.tolist(), k, want)
                    # Nobody leaves sync() before EVERY rank has read its generation words: a peer that
                    # ran ahead after the first barrier would overwrite an older buffer (step t + depth
                    # lands in buffer t mod depth) with a newer generation before this rank had read its
                    # tag — a spurious P2PError in the latest-rows-only mode (s <= depth - 1 with
                    # p2p_enforce_sync_interval = False).  All ranks reach this barrier whatever they found.
                    dist.barrier(group=self.group)
                if bad is not None:
                    raise P2PError(f"rank {self.rank}: generation words {bad[0]} in buffer {bad[1]}, "
                                   f"expected {bad[2]} from every rank (ranks out of step, or a stale read)")

    def close(self):
        try:
            self.sync()
        finally:       # a P2PError out of the last sync must not leak the exchange buffers or the local env
            try:
                if self._exchange is not None:
                    ex, self._exchange = self._exchange, None
                    self._global2 = self._tags = None
                    self._global = None
                    self._release_exchange(ex)     # unmap, meet, free
            finally:
                self.local.close()
