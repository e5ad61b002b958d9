"""N>1 path on CPU: world_size-2 `gloo` processes run the sharded env (oracle-backed test
double as the local stepper) and must reproduce the single-process batch bit-for-bit
after the packed all-gather."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = Path(__file__).resolve().parents[1]


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _octo_worker(rank, world, port, total, q):
    """OctoFlat-v0 through the sharded env: odd obs_dim (461) -> padded packed rows."""
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import gym_softrobot_amd as gsa
    from gym_softrobot_amd import _capi
    from gym_softrobot_amd.distributed import ShardedVecEnv, shard_bounds
    from tests.oracle_backend import OracleBackend

    lo, hi = shard_bounds(total, world, rank)
    cfg = _capi.octo_flat_config(hi - lo)
    cfg.n_substeps = 12
    local = gsa.VecOctoFlatEnv(hi - lo, backend=OracleBackend(cfg))
    local.cfg.n_substeps = 12
    env = ShardedVecEnv(local, total)
    obs0, _ = env.reset(seed=3)
    obs0 = obs0.clone().numpy()      # the gathered buffer is reused by the next call
    acts = np.random.default_rng(8).uniform(-22, 22, (total, 24)).astype(np.float32)
    o, r, te, tr, _ = env.step(acts)
    if rank == 0:
        q.put((obs0, o.clone().numpy(), r.clone().numpy(), te.clone().numpy(), tr.clone().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def _autoreset_worker(rank, world, port, total, T, q):
    """Sharded env over device-side auto-reset (staged reset records), overlapped gather."""
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import gym_softrobot_amd as gsa
    from gym_softrobot_amd import _capi
    from gym_softrobot_amd.distributed import ShardedVecEnv, shard_bounds
    from tests.oracle_backend import OracleBackend

    kw = dict(final_time=5e-4, time_step=1e-4, recording_fps=5000, n_elems=6)   # 3-step episodes
    lo, hi = shard_bounds(total, world, rank)
    cfg = _capi.softpendulum_config(hi - lo, **kw)
    local = gsa.VecSoftPendulumEnv(hi - lo, backend=OracleBackend(cfg), autoreset="device", **kw)
    env = ShardedVecEnv(local, total, overlap=True)
    env.reset(seed=11)
    acts = np.random.default_rng(2).uniform(-5, 5, (T, total)).astype(np.float32)
    out = []
    for t in range(T):
        o, r, te, tr, _ = env.step(acts[t])
        env.sync()
        out.append((o.clone().numpy(), r.clone().numpy(), te.clone().numpy(), tr.clone().numpy()))
    if rank == 0:
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def _worker(rank, world, port, total, T, q, overlap=False, transport="rccl"):
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import gym_softrobot_amd as gsa
    from gym_softrobot_amd import _capi
    from gym_softrobot_amd.distributed import ShardedVecEnv, shard_bounds
    from tests.oracle_backend import OracleBackend

    lo, hi = shard_bounds(total, world, rank)
    cfg = _capi.softpendulum_config(hi - lo)
    cfg.n_substeps = 40  # keep the CPU suite quick; the sharding logic does not depend on it
    local = gsa.VecSoftPendulumEnv(hi - lo, backend=OracleBackend(cfg))
    local.cfg.n_substeps = 40
    env = ShardedVecEnv(local, total, overlap=overlap, transport=transport)
    if transport == "p2p":      # no device to map on a CPU box: every rank must have fallen back, together
        assert env.transport == "rccl" and env._p2p_error
    obs0, _ = env.reset(seed=7)
    acts = np.random.default_rng(5).uniform(-22, 22, (T, total)).astype(np.float32)
    out = [obs0.clone().numpy()]
    for t in range(T):
        o, r, te, tr, _ = env.step(acts[t])
        env.sync()     # overlap=True: outputs are complete after sync()
        out.append((o.clone().numpy(), r.clone().numpy(), te.clone().numpy(), tr.clone().numpy()))
    if rank == 0:
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def _single(total, T):
    import gym_softrobot_amd as gsa
    from gym_softrobot_amd import _capi
    from tests.oracle_backend import OracleBackend

    cfg = _capi.softpendulum_config(total)
    cfg.n_substeps = 40
    env = gsa.VecSoftPendulumEnv(total, backend=OracleBackend(cfg), numpy_output=True)
    obs0, _ = env.reset(seed=7)
    acts = np.random.default_rng(5).uniform(-22, 22, (T, total)).astype(np.float32)
    out = [obs0.copy()]
    for t in range(T):
        o, r, te, tr, _ = env.step(acts[t])
        out.append((o.copy(), r.copy(), te.copy(), tr.copy()))
    return out


def test_pack_unpack_roundtrip_is_bit_exact():
    from gym_softrobot_amd.distributed import pack_outputs, unpack_outputs

    n = 9
    g = torch.Generator().manual_seed(0)
    obs = torch.randn((n, 4), generator=g)
    rew = torch.randn(n, dtype=torch.float64, generator=g) * 1e3
    rew[3] = -50.0
    term = (torch.arange(n) % 3 == 0).to(torch.uint8)
    trunc = (torch.arange(n) % 2 == 0)
    o, r, te, tr = unpack_outputs(pack_outputs(obs, rew, term, trunc), 4)
    assert torch.equal(o, obs) and torch.equal(r, rew)
    assert torch.equal(te, term.bool()) and torch.equal(tr, trunc)
    # odd observation widths are padded so that the float64 reward stays 8-byte aligned
    for od in (9, 25):
        obs = torch.randn((n, od), generator=g)
        p = pack_outputs(obs, rew, term, trunc)
        assert p.shape == (n, od + 1 + 4)
        o, r, te, tr = unpack_outputs(p, od)
        assert torch.equal(o, obs) and torch.equal(r, rew) and torch.equal(te, term.bool()) and torch.equal(tr, trunc)


def test_shard_bounds():
    from gym_softrobot_amd.distributed import shard_bounds

    assert [shard_bounds(32768, 8, r) for r in (0, 7)] == [(0, 4096), (28672, 32768)]
    with pytest.raises(ValueError):
        shard_bounds(10, 4, 0)


@pytest.mark.parametrize("overlap", [False, True], ids=["blocking", "overlapped"])
def test_world2_gloo_matches_single_process(oracle_built, overlap):
    total, T, world = 6, 4, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, T, q, overlap)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _single(total, T)
    np.testing.assert_array_equal(got[0], ref[0])
    for g, r in zip(got[1:], ref[1:]):
        for a, b in zip(g, r):
            np.testing.assert_array_equal(a, b)


def test_p2p_transport_falls_back_to_the_collective_together(oracle_built):
    """transport="p2p" needs IPC-mapped device buffers; where its set-up fails (here: no GPU) ALL ranks
    agree to stay with the collective, and the run is the ordinary one."""
    total, T, world = 6, 3, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, total, T, q, True, "p2p")) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    ref = _single(total, T)
    np.testing.assert_array_equal(got[0], ref[0])
    for g, r in zip(got[1:], ref[1:]):
        for a, b in zip(g, r):
            np.testing.assert_array_equal(a, b)


def test_world2_gloo_octoflat_matches_single_process(oracle_built):
    import gym_softrobot_amd as gsa
    from gym_softrobot_amd import _capi
    from tests.oracle_backend import OracleBackend

    total, world = 4, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_octo_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = _capi.octo_flat_config(total)
    cfg.n_substeps = 12
    env = gsa.VecOctoFlatEnv(total, backend=OracleBackend(cfg), numpy_output=True)
    env.cfg.n_substeps = 12
    obs0, _ = env.reset(seed=3)
    acts = np.random.default_rng(8).uniform(-22, 22, (total, 24)).astype(np.float32)
    ref = (obs0.copy(),) + tuple(np.asarray(x).copy() for x in env.step(acts)[:4])
    assert got[0].shape == (total, 461)
    for a, b in zip(got, ref):
        np.testing.assert_array_equal(a, b)


def test_world2_gloo_device_autoreset_matches_single_process(oracle_built):
    import gym_softrobot_amd as gsa
    from gym_softrobot_amd import _capi
    from tests.oracle_backend import OracleBackend

    total, T, world = 4, 9, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_autoreset_worker, args=(r, world, port, total, T, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    kw = dict(final_time=5e-4, time_step=1e-4, recording_fps=5000, n_elems=6)
    cfg = _capi.softpendulum_config(total, **kw)
    env = gsa.VecSoftPendulumEnv(total, backend=OracleBackend(cfg), numpy_output=True, autoreset=True, **kw)
    env.reset(seed=11)
    acts = np.random.default_rng(2).uniform(-5, 5, (T, total)).astype(np.float32)
    restarted = 0
    for t in range(T):
        o, r, te, tr, _ = env.step(acts[t])
        restarted += int((env._steps == 0).sum())
        for a, b in zip(got[t], (o, r, te, tr)):
            np.testing.assert_array_equal(a, b)
    assert restarted >= total       # episodes really ended and restarted on the way


def _world8_worker(rank, world, port, total, q):
    """8 ranks, 2 envs each: steps, a masked reset whose mask straddles several shard boundaries,
    more steps; overlapped all-gather."""
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import gym_softrobot_amd as gsa
    from gym_softrobot_amd import _capi
    from gym_softrobot_amd.distributed import ShardedVecEnv, shard_bounds
    from tests.oracle_backend import OracleBackend

    lo, hi = shard_bounds(total, world, rank)
    cfg = _capi.softpendulum_config(hi - lo, n_elems=8)
    cfg.n_substeps = 10
    local = gsa.VecSoftPendulumEnv(hi - lo, n_elems=8, backend=OracleBackend(cfg))
    local.cfg.n_substeps = 10
    env = ShardedVecEnv(local, total, overlap=True)
    out = _world8_rollout(env, total, True)
    if rank == 0:
        q.put(out)
    dist.barrier()
    dist.destroy_process_group()


def _world8_rollout(env, total, sharded):
    acts = np.random.default_rng(9).uniform(-22, 22, (6, total)).astype(np.float32)
    # envs 1..2 (ranks 0|1), 5..8 (ranks 2|3|4) and 13, 15 (ranks 6, 7); ranks 5 resets nothing
    mask = np.zeros(total, bool)
    mask[[1, 2, 5, 6, 7, 8, 13, 15]] = True
    obs0, _ = env.reset(seed=21)
    out = [torch.as_tensor(obs0).clone().numpy()]
    for t in range(6):
        if t == 3:
            o, _ = env.reset(mask=mask)
            out.append(torch.as_tensor(o).clone().numpy())
        o, r, te, tr, _ = env.step(acts[t])
        if sharded:
            env.sync()
        out.append(tuple(torch.as_tensor(x).clone().numpy() for x in (o, r, te, tr)))
    return out


def test_world8_gloo_masked_reset_across_shard_boundaries(oracle_built):
    """The world size of BASELINE configs[3] / configs[4]: 8 ranks.  A masked reset in the middle of
    the rollout touches envs on both sides of three shard boundaries and skips one rank entirely;
    every gathered record equals one process stepping all 16 envs."""
    import gym_softrobot_amd as gsa
    from gym_softrobot_amd import _capi
    from tests.oracle_backend import OracleBackend

    total, world = 16, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_world8_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=400)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    cfg = _capi.softpendulum_config(total, n_elems=8)
    cfg.n_substeps = 10
    env = gsa.VecSoftPendulumEnv(total, n_elems=8, backend=OracleBackend(cfg), numpy_output=True)
    env.cfg.n_substeps = 10
    ref = _world8_rollout(env, total, False)
    assert len(got) == len(ref) == 8
    for g, r in zip(got, ref):
        g, r = (g if isinstance(g, tuple) else (g,)), (r if isinstance(r, tuple) else (r,))
        for a, b in zip(g, r):
            np.testing.assert_array_equal(a, b)
    # the masked envs really restarted (fresh draws: their reset observation differs from the seeded one)
    assert not np.array_equal(got[4][[1, 5, 15]], got[0][[1, 5, 15]])


def _p2p_failure_worker(rank, world, port, total, mode, q):
    """transport="p2p" set-up with a backend double whose exchange calls fail at a chosen point on
    a chosen rank: every rank must come out with the collective, nobody may hang."""
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import ctypes

    import gym_softrobot_amd as gsa
    from gym_softrobot_amd import _capi
    from gym_softrobot_amd.distributed import ShardedVecEnv, shard_bounds
    from tests.oracle_backend import OracleBackend

    class Double(OracleBackend):
        device_index = 0
        freed, closed = 0, 0

        def exchange_alloc(self, n_words):
            if mode == "alloc_fails_on_rank_1" and rank == 1:
                raise RuntimeError("no uncached memory here")
            t = torch.zeros(n_words)
            self._keep = getattr(self, "_keep", []) + [t]
            return t, t.data_ptr(), bytes(64), "uncached"

        def exchange_open(self, handle, owner_device):
            if mode == "open_fails_on_rank_0" and rank == 0:
                raise RuntimeError("peer access refused")
            return 0xDEAD0000 + owner_device

        def exchange_close(self, ptr):
            Double.closed += 1

        def exchange_free(self, ptr):
            Double.freed += 1

        def scatter_rows(self, packed, peer_ptrs, first_row, tag_word=-1, tag=0):
            # a transport that only ever reaches this rank's own buffer: the self-test must notice
            mine = [t for t in self._keep if t.data_ptr() in peer_ptrs][0]
            w = packed.shape[1]
            mine[first_row * w:(first_row + packed.shape[0]) * w] = packed.reshape(-1)
            if tag_word >= 0:
                mine[tag_word:tag_word + 1].view(torch.int32)[0] = ctypes.c_int32(tag).value

    lo, hi = shard_bounds(total, world, rank)
    cfg = _capi.softpendulum_config(hi - lo, n_elems=8)
    cfg.n_substeps = 5
    local = gsa.VecSoftPendulumEnv(hi - lo, n_elems=8, backend=Double(cfg))
    local.cfg.n_substeps = 5
    env = ShardedVecEnv(local, total, overlap=True, transport="p2p")
    assert env.transport == "rccl" and env._p2p_error, (rank, env.transport)
    obs0, _ = env.reset(seed=3)
    o, r, te, tr, _ = env.step(np.linspace(-5, 5, total).astype(np.float32))
    env.sync()
    q.put((rank, env._p2p_error, Double.freed, Double.closed, o.clone().numpy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("mode,who_says", [("alloc_fails_on_rank_1", "exchange_alloc"),
                                           ("open_fails_on_rank_0", "exchange_open"),
                                           ("self_test_fails", "self-test round 0")])
def test_p2p_setup_failures_leave_every_rank_on_the_collective(oracle_built, mode, who_says):
    """an asymmetric failure in the p2p set-up must not leave the other ranks blocked in a
    collective.  A failing allocation on one rank, a failing mapping on another, and a transport whose
    stores never reach the peers (the self-test's job): all ranks fall back together, release what
    they had allocated or mapped, and the run is the ordinary one."""
    total, world = 6, 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_p2p_failure_worker, args=(r, world, port, total, mode, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = sorted((q.get(timeout=240) for _ in range(world)), key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    errs = [g[1] for g in got]
    assert any(who_says in e for e in errs), errs         # the rank it happened on names it; the others know a peer failed
    for rank, err, freed, closed, obs in got:
        allocated = 0 if (mode == "alloc_fails_on_rank_1" and rank == 1) else 2
        assert freed == allocated, (mode, rank, freed)
        if mode == "self_test_fails":
            assert closed == 2 * (world - 1)               # every peer mapping of both buffers was unmapped again
    for g in got[1:]:
        np.testing.assert_array_equal(g[4], got[0][4])     # and the gathered rows are the same everywhere


def test_p2p_without_overlap_warns_and_uses_the_collective(oracle_built):
    """transport="p2p" needs overlap=True and a process group; asked for without them it says so
    (a RuntimeWarning, `_p2p_error`) instead of silently running the collective."""
    import gym_softrobot_amd as gsa
    from gym_softrobot_amd import _capi
    from gym_softrobot_amd.distributed import ShardedVecEnv
    from tests.oracle_backend import OracleBackend

    cfg = _capi.softpendulum_config(2, n_elems=8)
    local = gsa.VecSoftPendulumEnv(2, n_elems=8, backend=OracleBackend(cfg))
    with pytest.warns(RuntimeWarning, match="overlap=True"):
        env = ShardedVecEnv(local, 2, transport="p2p")
    assert env.transport == "rccl" and "overlap=True" in env._p2p_error
    with pytest.raises(ValueError):
        ShardedVecEnv(local, 2, transport="carrier pigeon")


def _muscle_worker(rank, world, port, total, q):
    """OctoReach-v0 through the sharded env: per-env random targets (env i seeded seed + i across the shard boundary),
    even obs_dim 1512, 480 actions per env."""
    sys.path.insert(0, str(ROOT))
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import gym_softrobot_amd as gsa
    from gym_softrobot_amd import _capi
    from gym_softrobot_amd.distributed import ShardedVecEnv, shard_bounds
    from tests.oracle_backend import OracleBackend

    lo, hi = shard_bounds(total, world, rank)
    cfg = _capi.muscle_octopus_config(_capi.ENV_REACH, hi - lo)
    cfg.n_substeps = 10
    local = gsa.VecReachEnv(hi - lo, backend=OracleBackend(cfg))
    local.cfg.n_substeps = 10
    env = ShardedVecEnv(local, total)
    obs0, _ = env.reset(seed=3)
    obs0 = obs0.clone().numpy()
    acts = np.random.default_rng(8).uniform(0, 0.6, (total, 480)).astype(np.float32)
    o, r, te, tr, _ = env.step(acts)
    if rank == 0:
        q.put((obs0, o.clone().numpy(), r.clone().numpy(), te.clone().numpy(), tr.clone().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_world2_gloo_muscle_octopus_matches_single_process(oracle_built):
    """The N > 1 path for the round-6 envs: two gloo ranks, OctoReach-v0 sharded 2 + 2, against one process stepping all 4."""
    import gym_softrobot_amd as gsa
    from gym_softrobot_amd import _capi
    from tests.oracle_backend import OracleBackend

    total, world = 4, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_muscle_worker, args=(r, world, port, total, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = q.get(timeout=240)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    cfg = _capi.muscle_octopus_config(_capi.ENV_REACH, total)
    cfg.n_substeps = 10
    env = gsa.VecReachEnv(total, backend=OracleBackend(cfg), numpy_output=True)
    env.cfg.n_substeps = 10
    obs0, _ = env.reset(seed=3)
    acts = np.random.default_rng(8).uniform(0, 0.6, (total, 480)).astype(np.float32)
    ref = (obs0.copy(),) + tuple(np.asarray(x).copy() for x in env.step(acts)[:4])
    assert got[0].shape == (total, 1512)
    assert len({tuple(np.round(row[-18:-15], 6)) for row in got[0]}) == total      # four different targets
    for a, b in zip(got, ref):
        np.testing.assert_array_equal(a, b)
