"""Shared host logic of the batched envs: seeding, masked/auto reset, step bookkeeping.

The reference has a single-env API only (no vector env, no auto-reset: a finished env is
reset by the caller, gym_softrobot/debug/make.py:16-23).  The batched form adds, per
SURVEY.md §8(f) N2:
  * `reset(mask=...)`  partial reset of a subset of the resident envs,
  * `autoreset=True`   Gymnasium-1.0 VectorEnv NEXT_STEP semantics: an env that returned
                       terminated/truncated at step t is reset (instead of stepped) by the
                       call at t+1, which returns its reset observation, reward 0 and both
                       flags False.  Costs one small device->host read of the flags per step.
  * `autoreset="device"`  the same semantics with no host round trip: the next
                       `queue_depth` resets of every env are drawn ahead of time from the
                       env's own NumPy stream (so the streams are the ones `autoreset=True`
                       would consume, draw for draw) and staged on the device
                       (softrod_queue_push*); the step applies them itself.  The host tops
                       the queue up within every `queue_depth` steps (one small non-blocking read).  `infos`
                       then carry device tensors.
"""
from __future__ import annotations

from collections import deque
from typing import Any, Dict, List, Optional, Sequence, Union

import numpy as np

from .. import _capi
from ..seeding import np_random
from ..spaces import Box

try:  # pragma: no cover
    from gymnasium import Env as GymEnv  # type: ignore
except Exception:  # noqa: BLE001
    class GymEnv:  # minimal stand-in for gymnasium.Env
        metadata: Dict[str, Any] = {}
        render_mode = None
        _np_random = None

        def reset(self, *, seed=None, options=None):
            if seed is not None:
                self._np_random, self._np_random_seed = np_random(seed)

        @property
        def np_random(self):
            if self._np_random is None:
                self._np_random, self._np_random_seed = np_random()
            return self._np_random

        @property
        def unwrapped(self):
            return self

        def close(self):
            pass


def time_table(cfg: _capi.SoftrodConfig, n_steps: int) -> np.ndarray:
    """float64 simulated time after k env.steps, accumulated exactly as
    `self.time = self.do_step(self.simulator, self.time, self.time_step)` does
    (soft_pendulum.py:183-184): PositionVerlet adds dt/2 twice per substep.

    `np.add.accumulate` performs the same additions in the same order as the Python loop it
    replaces (t = t + h, one after the other), so the table is bit-identical to it
    (tests/test_host_logic.py) — and 300 times faster: growing the table in the middle of a
    rollout used to stall the launch stream for 10-20 ms on step 64, 128, 256, ..."""
    per = int(cfg.n_substeps) * (2 if cfg.time_two_half_adds else 1)
    inc = np.float64(0.5) * np.float64(cfg.dt) if cfg.time_two_half_adds else np.float64(cfg.dt)
    out = np.empty(n_steps + 1, np.float64)
    out[0] = 0.0
    if n_steps > 0 and per > 0:
        acc = np.add.accumulate(np.full(n_steps * per, inc, np.float64))
        out[1:] = acc[per - 1 :: per]
    elif n_steps > 0:
        out[1:] = 0.0
    return out


class VecRodEnvBase:
    """N parallel envs resident on one GPU.

    reset(seed=None|int|sequence, options=None, mask=None) -> (obs[N,obs_dim] float32, infos)
    step(actions[N,action_dim])  -> (obs, reward[N] float64, terminated[N] bool,
                                     truncated[N] bool, infos)
    Outputs are torch tensors on the device (zero-copy views of the backend's buffers,
    overwritten by the next call) unless `numpy_output=True`.
    """

    metadata: Dict[str, Any] = {"render_modes": ["rgb_array"], "render_fps": 25}
    action_low: float = -1.0
    action_high: float = 1.0

    def __init__(self, num_envs: int, cfg: _capi.SoftrodConfig, *, render_mode, config_generate_video,
                 device: int, numpy_output: bool, autoreset: bool, backend):
        if render_mode not in {None, *self.metadata["render_modes"]}:
            raise ValueError(f"Unsupported render mode: {render_mode}")  # soft_pendulum.py:69-70
        self.config_generate_video = bool(config_generate_video)
        self.render_mode = render_mode
        self.num_envs = int(num_envs)
        self.cfg = cfg
        self.numpy_output = numpy_output
        if autoreset not in (False, True, "host", "device"):
            raise ValueError("autoreset must be False, True/'host' or 'device'")
        self.device_autoreset = autoreset == "device"
        self.autoreset = bool(autoreset) and not self.device_autoreset
        self._queue_depth = self._ring_depth = 32      # (queue_depth / top_up_every properties)
        self.action_dim = _capi.config_action_dim(cfg)
        self.obs_dim = _capi.config_obs_dim(cfg)
        self.n_action = self.action_dim
        lo, hi = self.action_low, self.action_high
        self.single_action_space = Box(lo, hi, shape=(self.action_dim,), dtype=np.float32)
        self.single_observation_space = Box(-np.inf, np.inf, shape=(self.obs_dim,), dtype=np.float32)
        self.action_space = Box(lo, hi, shape=(self.num_envs, self.action_dim), dtype=np.float32)
        self.observation_space = Box(-np.inf, np.inf, shape=(self.num_envs, self.obs_dim), dtype=np.float32)
        if backend is None:
            from ..backend import HipRodBackend

            backend = HipRodBackend(cfg, device=device)
        self.backend = backend
        self._rngs: List[Optional[np.random.Generator]] = [None] * self.num_envs
        self._steps = np.zeros(self.num_envs, np.int64)  # env.steps since each env's reset
        self._time_tab = time_table(cfg, 128)     # (one SoftPendulum episode; grows by doubling)
        self._needs_reset = np.zeros(self.num_envs, bool)
        self._produced = np.zeros(self.num_envs, np.int64)   # reset records staged per env
        self._staged = [deque() for _ in range(self.num_envs)]   # their draws, oldest first
        self._popped = np.zeros(self.num_envs, np.int64)     # draws removed from _staged so far
        self._since_top_up = 0
        self._status_in_flight = False
        self.top_up_paused = False     # True: the caller reads the queue counters and stages records itself (tests)
        if self.device_autoreset:
            self.backend.autoreset_enable(self.queue_depth)
        # soft_pendulum.py:117-126: RodCallBack -> rod_parameters_dict, one sample per env.step.
        # Video/plot generation from it stays out of scope; the data tap is here (env 0, or
        # `record_envs` set before reset).
        self.record_envs = (0,)
        self.recorder = None

    spec = None      # set by gymnasium.make_vec (`env.unwrapped.spec = ...`)

    @property
    def unwrapped(self):
        return self

    @property
    def queue_depth(self) -> int:
        """Unconsumed reset records the host keeps staged per env (device auto-reset).  May be
        lowered at any time; it cannot exceed the device ring allocated when the env was built."""
        return self._queue_depth

    @queue_depth.setter
    def queue_depth(self, depth: int) -> None:
        depth = int(depth)
        if depth < 3:
            raise ValueError("queue_depth must be at least 3")
        if self.device_autoreset and depth > self._ring_depth:
            raise ValueError(f"queue_depth {depth} exceeds the device ring ({self._ring_depth} records per env)")
        self._queue_depth = depth

    @property
    def top_up_every(self) -> int:
        """Deadline of the non-blocking top-up in steps (see _top_up_tick), DERIVED from
        queue_depth so that the two cannot drift apart: an env uses at most one record per two
        steps, so between a reading of the counters and the end of the NEXT top-up (fewer than
        2 * top_up_every steps) it uses at most top_up_every records — fewer than the
        queue_depth - 1 that are certainly staged at the reading."""
        every = max(1, self._queue_depth - 2)
        assert every <= self._queue_depth - 1
        return every

    # -- hooks ---------------------------------------------------------------------
    def _reset_backend(self, mask: np.ndarray, use_mask: bool, draws: Optional[dict] = None) -> None:
        """Reset the masked rods on the backend from what the env's build function draws from
        self._rngs[i] (`self._draw(i, draws)`: a draw taken earlier for env i, if given)."""
        raise NotImplementedError

    def _draw(self, i: int, draws: Optional[dict]):
        return draws[i] if draws is not None and i in draws else self._draw_reset(i)

    def _draw_reset(self, i: int):
        """What one reset of env i draws from self._rngs[i], as the tuple of per-env arguments
        `_reset_backend` / `_queue_from_draws` understand."""
        raise NotImplementedError

    def _queue_from_draws(self, draws, counts) -> None:
        """backend.queue_push*(...) from draws[i] = list of counts[i] draws."""
        raise NotImplementedError

    def _top_up(self, status=None) -> None:
        """Stage resets until every env has `queue_depth` unconsumed records (device mode).

        `status` = (consumed, underflow) from a finished non-blocking read; without it the
        counters are read now, which waits for the stream."""
        consumed = self._sync_staged(status)
        counts = (self.queue_depth - (self._produced - consumed)).astype(np.int32)
        need = np.nonzero(counts > 0)[0]
        if need.size:
            draws = [()] * self.num_envs
            for i in need:
                d = [self._draw_reset(i) for _ in range(int(counts[i]))]
                draws[i] = d
                self._staged[i].extend(d)
            self._queue_from_draws(draws, counts)
            self._produced += counts
        self._since_top_up = 0
        self._status_in_flight = False

    def _top_up_tick(self) -> None:
        """Called once per step in device mode.  Right after a top-up the queue counters start
        their way to the host without stalling the stream (softrod_queue_status_begin); from
        half way to the deadline (`top_up_every` steps since the last top-up) each step looks
        whether they have arrived and tops up if so; at the deadline it waits for that read
        only — the steps enqueued after it keep the GPU busy while the host draws and stages
        (a host that runs ahead of the GPU, as a rollout loop without a policy does, always ends
        up there; one that reads every step's outputs tops up at the half-way mark).

        Why an old reading is enough: the counters only grow, so records computed from it always
        fit; which draw an env's k-th reset uses does not depend on when it was staged; and an
        env uses at most one record per two steps (the step that resets does not also end an
        episode), so between a reading and the end of the NEXT top-up — fewer than
        2 * top_up_every steps — it uses at most top_up_every < queue_depth records."""
        if self.top_up_paused:
            return
        self._since_top_up += 1
        due = self._since_top_up >= self.top_up_every
        if self._status_in_flight:
            if 2 * self._since_top_up >= self.top_up_every:
                st = self.backend.queue_status_poll(due)
                if st is not None:
                    self._top_up(st)
        elif due:
            self._top_up()
        elif hasattr(self.backend, "queue_status_begin"):
            self.backend.queue_status_begin()
            self._status_in_flight = True

    def _sync_staged(self, status=None) -> np.ndarray:
        """Read how many staged records the device has used and forget their draws."""
        from .. import _capi as capi

        consumed, underflow = self.backend.queue_status() if status is None else status
        if underflow:
            raise capi.SoftrodError(
                f"{underflow} auto-resets found no staged record: top up more often or raise queue_depth")
        for i in np.nonzero(consumed > self._popped)[0]:
            for _ in range(int(consumed[i] - self._popped[i])):
                self._staged[i].popleft()
            self._popped[i] = consumed[i]
        return consumed

    def _infos(self, times: np.ndarray) -> Dict[str, Any]:
        return {"time": times, "TimeLimit.truncated": times > self.cfg.final_time}

    def _validate_actions(self, actions) -> None:
        pass

    # -- helpers -------------------------------------------------------------------
    def _times(self) -> np.ndarray:
        kmax = int(self._steps.max()) if self.num_envs else 0
        if kmax >= len(self._time_tab):
            self._time_tab = time_table(self.cfg, max(2 * kmax, 16))
        return self._time_tab[self._steps]

    def _out(self, t):
        return t.cpu().numpy() if self.numpy_output else t

    def _seed_rngs(self, seed, mask: np.ndarray) -> None:
        n = self.num_envs
        if seed is None or isinstance(seed, (int, np.integer)):
            seeds = [None if seed is None else int(seed) + i for i in range(n)]
        else:
            seeds = list(seed)
            if len(seeds) != n:
                raise ValueError("need one seed per env")
        reseeded = np.zeros(n, bool)
        for i in range(n):
            if mask[i] and (seeds[i] is not None or self._rngs[i] is None):
                self._rngs[i], _ = np_random(seeds[i])
                reseeded[i] = True
        return reseeded

    # -- API -----------------------------------------------------------------------
    def reset(
        self,
        *,
        seed: Optional[Union[int, Sequence[Optional[int]]]] = None,
        options: Optional[dict] = None,
        mask: Optional[np.ndarray] = None,
    ):
        n = self.num_envs
        m = np.ones(n, bool) if mask is None else np.asarray(mask, bool).reshape(n)
        reseeded = self._seed_rngs(seed, m)
        draws = None
        if self.device_autoreset and self._produced.any():
            # The next draws of these envs' streams are already staged on the device.  A manual
            # reset takes the env's NEXT draw, i.e. the oldest staged one (and marks it used);
            # a re-seeded env starts a new stream, so everything staged for it is dropped.
            self._sync_staged()
            by = np.zeros(n, np.int32)
            draws = {}
            for i in np.nonzero(m)[0]:
                if reseeded[i]:
                    by[i] = -1
                    self._popped[i] += len(self._staged[i])
                    self._staged[i].clear()
                elif self._staged[i]:
                    by[i] = 1
                    draws[i] = self._staged[i].popleft()
                    self._popped[i] += 1
            self.backend.queue_advance(by)
        self._reset_backend(m, mask is not None, draws)
        self._steps[m] = 0
        self._needs_reset[m] = False
        # _prev_action lives with the resident state (softrod_state_view.prev_action): it
        # survives reset except where the reference clears it (soft_pendulum_3d.py:68)
        obs = self.backend.observe(None)
        if self.device_autoreset:
            self._top_up()
        if self.config_generate_video and not self.is_octo:
            from ..diagnostics import RodRecorder

            self.recorder = RodRecorder(self.backend, self.record_envs)   # fresh dict per reset (:118)
        elif self.is_octo and (self.config_generate_video or getattr(self, "config_save_head_data", False)):
            from ..diagnostics import OctoRecorder

            # flat_env.py:188-206: per-arm RodCallBack dicts and the head's dict, fresh per reset
            self.recorder = OctoRecorder(self.backend, self.record_envs[0], rods=self.config_generate_video,
                                         head=self.config_save_head_data)
        return self._out(obs), {}

    @property
    def is_octo(self) -> bool:
        return int(self.cfg.env_kind) == _capi.ENV_OCTO_FLAT

    @property
    def rod_parameters_dict(self):
        """The reference's `rod_parameters_dict` for the first recorded env."""
        return None if self.recorder is None or self.is_octo else self.recorder.params[0]

    @property
    def rod_parameters_dict_list(self):
        """FlatEnv.rod_parameters_dict_list (octopus/flat_env.py:190-198): one dict per arm."""
        return getattr(self.recorder, "rod_parameters_dict_list", None)

    @property
    def head_dict(self):
        """FlatEnv.head_dict (octopus/flat_env.py:199-206)."""
        return getattr(self.recorder, "head_dict", None)

    def _step_device_autoreset(self, a):
        import torch

        obs, reward, term, trunc = self.backend.step(a)    # auto-reset pass + step kernel
        self._top_up_tick()
        if getattr(self, "_dev_time", None) is None:
            self._dev_time = self.backend.state()["time"]
        infos = {"time": self._out(self._dev_time), "TimeLimit.truncated": self._out(trunc.view(torch.bool))}
        return (self._out(obs), self._out(reward), self._out(term.view(torch.bool)),
                self._out(trunc.view(torch.bool)), infos)

    def step(self, actions):
        import torch

        self._validate_actions(actions)
        a = torch.as_tensor(actions, dtype=torch.float32, device=self.backend.device)
        a = a.reshape(self.num_envs, self.action_dim)
        if self.device_autoreset:
            return self._step_device_autoreset(a)
        pending = self._needs_reset.copy() if self.autoreset else None
        saved_prev = None
        if pending is not None and pending.any():
            # the action of a restarting env is ignored (NEXT_STEP): its _prev_action must stay
            # the one of its last real step, as env.reset() would find it
            saved_prev = self.backend.prev_action_rows()[torch.from_numpy(pending)].clone()
        obs, reward, term, trunc = self.backend.step(a)   # also records _prev_action[:] = action
        self._steps += 1
        if pending is not None and pending.any():
            # NEXT_STEP auto-reset: these envs finished on the previous call; their step above
            # is discarded, they restart and report their reset observation
            keep = obs.clone()
            self.backend.prev_action_rows()[torch.from_numpy(pending)] = saved_prev
            self._seed_rngs(None, pending)
            self._reset_backend(pending, True)
            self._steps[pending] = 0
            robs = self.backend.observe(None)
            pm = torch.from_numpy(pending).to(robs.device)
            obs = torch.where(pm[:, None], robs, keep)
            reward = torch.where(pm, torch.zeros_like(reward), reward)
            term = torch.where(pm, torch.zeros_like(term), term)
            trunc = torch.where(pm, torch.zeros_like(trunc), trunc)
        if self.autoreset:
            self._needs_reset = (term | trunc).cpu().numpy().astype(bool)
        if self.recorder is not None:
            self.recorder.record()
        times = self._times()
        infos = self._infos(times)
        return (
            self._out(obs),
            self._out(reward),
            self._out(term.view(torch.bool)),     # uint8 0/1 -> bool, zero-copy
            self._out(trunc.view(torch.bool)),
            infos,
        )

    def capture_policy_step(self, policy):
        """One HIP graph for `actions = policy(obs); step(actions)` — the launch-bound tail of an on-device
        rollout (a small policy is half a dozen tiny kernels per step) becomes ONE graph launch per env.step.
        `softrod_step` is capturable: it enqueues kernels on the caller's stream and does nothing else (no
        allocation, no synchronisation, no host read) while timing is off.

        policy: obs (N, obs_dim) float32 device tensor -> actions (N, action_dim) float32, pure device work.
        Returns replay() -> (obs, reward, terminated, truncated): the backend's resident output tensors, overwritten
        by every replay (as `step` does).  The observation the policy reads is the one the previous step wrote —
        call `reset` BEFORE capturing.  Needs autoreset off or "device" (the host-driven NEXT_STEP mode reads flags
        back every step and cannot live in a graph); the queue top-ups of the device mode stay outside the graph
        and run between replays.  Results are bit-identical to the eager loop (tests/test_gpu_policy_loop.py).
        The kernel arguments (RodParams, array pointers) are baked into the graph BY VALUE at capture: after
        anything that changes them (a radius profile, an action basis, enabling auto-reset) capture again.
        Kernel timing (set_timing) is switched off by the capture and stays off."""
        import torch

        if self.autoreset:
            raise NotImplementedError("capture_policy_step needs autoreset=False or autoreset='device'")
        if self.recorder is not None:
            raise NotImplementedError("capture_policy_step with a diagnostics recorder: the per-step host tap "
                                      "cannot live in a graph (config_generate_video=False)")
        be = self.backend
        if hasattr(be, "set_timing"):
            be.set_timing(0)                      # event records around the kernel are not part of the graph:
                                                  # kernel timing is OFF from here on (call set_timing again to re-arm)
        side = torch.cuda.Stream(device=be.device)
        side.wait_stream(torch.cuda.current_stream(be.device))
        with torch.cuda.stream(side):             # warm-up off the capture: the POLICY only (lazy BLAS handles,
            for _ in range(3):                    # allocator pools); it is a pure function of the observation, and
                policy(be.obs)                    # softrod_step has nothing lazy to warm up, so no state moves
        torch.cuda.current_stream(be.device).wait_stream(side)
        torch.cuda.synchronize(be.device)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            be.step(policy(be.obs))

        def replay():
            graph.replay()
            if self.device_autoreset:
                self._top_up_tick()
            else:
                self._steps += 1                  # what step() books: infos["time"] of a later eager step stays right
            return be.obs, be.reward, be.terminated.view(torch.bool), be.truncated.view(torch.bool)

        replay.graph = graph
        return replay

    def step_packed(self, actions, out=None):
        """step() with every per-env output in one (N, packed_width) float32 buffer written
        by the kernel itself (distributed.unpack_outputs gives views); no host auto-reset."""
        import torch

        if self.autoreset:
            raise NotImplementedError("step_packed does not auto-reset on the host; use autoreset='device'")
        self._validate_actions(actions)
        a = torch.as_tensor(actions, dtype=torch.float32, device=self.backend.device)
        a = a.reshape(self.num_envs, self.action_dim)
        packed = self.backend.step_packed(a) if out is None else self.backend.step_packed(a, out)
        if self.device_autoreset:
            self._top_up_tick()
            return packed, {}
        self._steps += 1
        return packed, self._infos(self._times())

    # -- checkpoint / resume ----------------------------------------------------------
    def state_dict(self) -> Dict[str, Any]:
        """Everything needed to continue this batch exactly where it is: the resident state
        (backend.snapshot()) and the host bookkeeping (steps since reset, pending auto-resets,
        every env's RNG state).  The reference never serialises an env; this is
        an addition.  Not available with autoreset="device" (reset records staged on the device
        belong to the RNG streams' future)."""
        if self.device_autoreset:
            raise NotImplementedError("state_dict() with autoreset='device': staged reset records are not captured")
        extra = {k: np.array(getattr(self, k)) for k in ("targets", "_traj") if getattr(self, k, None) is not None}
        return {
            "backend": self.backend.snapshot(),
            "steps": self._steps.copy(),
            "needs_reset": self._needs_reset.copy(),
            "rng": [None if g is None else g.bit_generator.state for g in self._rngs],
            "extra": extra,
        }

    def load_state_dict(self, sd: Dict[str, Any]) -> None:
        self.backend.restore(sd["backend"])
        self._steps[:] = sd["steps"]
        self._needs_reset[:] = sd["needs_reset"]
        for i, st in enumerate(sd["rng"]):
            if st is None:
                self._rngs[i] = None
            else:
                if self._rngs[i] is None:
                    self._rngs[i], _ = np_random(0)
                self._rngs[i].bit_generator.state = st
        for k, v in sd.get("extra", {}).items():
            setattr(self, k, np.array(v))

    def close(self):
        if self.backend is not None and hasattr(self.backend, "close"):
            self.backend.close()
