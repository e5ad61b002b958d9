"""Test double for the device backend: same interface as
gym_softrobot_amd.backend.HipRodBackend, arithmetic by the CPU oracle.  Lives under
tests/ on purpose — it lets the CPU suite exercise the host logic (env classes,
seeding, sharding, packed all-gather) without a GPU.  It is never importable from the
product package."""
from __future__ import annotations

import numpy as np
import torch

from gym_softrobot_amd import _capi
from oracle import oracle_c


class OracleBackend:
    def __init__(self, cfg, omp: bool = False):
        self.cfg = cfg.copy()
        self.n_envs = int(cfg.n_envs)
        self.device = torch.device("cpu")
        self.is3d = cfg.env_kind == _capi.ENV_SOFTPENDULUM3D
        self.isarm = cfg.env_kind == _capi.ENV_ARM_SINGLE
        self.isocto = cfg.env_kind == _capi.ENV_OCTO_FLAT
        self.issoftarm = cfg.env_kind == _capi.ENV_SOFT_ARM
        self.ispush = cfg.env_kind == _capi.ENV_ARM_PUSH
        self.ispull = cfg.env_kind == _capi.ENV_ARM_PULL_WEIGHT
        self.ismocto = int(cfg.env_kind) in _capi.MUSCLE_OCTOPUS_ENVS      # CrawlEnv / ArmTwoEnv / ReachEnv (tests/oracle_mocto.py)
        # softrod_state_view.control: SoftArmTracking keeps tick and the target there
        self._ctrl = torch.zeros((4, int(cfg.n_envs)), dtype=torch.float64)
        self.action_dim = _capi.config_action_dim(cfg)
        self.obs_dim = _capi.config_obs_dim(cfg)
        if self.ismocto:
            try:
                from tests.oracle_mocto import MuscleOctopusOracleEnv
            except ImportError:                     # imported with tests/ itself on the path
                from oracle_mocto import MuscleOctopusOracleEnv

            self.rods = [MuscleOctopusOracleEnv(self.cfg) for _ in range(self.n_envs)]
            self._octo_obs = [None] * self.n_envs
        elif self.isocto or self.ispull:
            self.rods = [oracle_c.OracleOcto(self.cfg) for _ in range(self.n_envs)]
            self._octo_obs = [None] * self.n_envs
        else:
            self.rods = [oracle_c.OracleRod(self.cfg, omp=omp) for _ in range(self.n_envs)]
        self.obs = torch.zeros((self.n_envs, self.obs_dim), dtype=torch.float32)
        self.reward = torch.zeros(self.n_envs, dtype=torch.float64)
        self.terminated = torch.zeros(self.n_envs, dtype=torch.uint8)
        self.truncated = torch.zeros(self.n_envs, dtype=torch.uint8)
        self.aux = torch.zeros((self.n_envs, 1), dtype=torch.float64) if self.is3d else None
        self._prev = np.zeros((self.n_envs, self.action_dim), np.float32)
        self._queue = None          # device-side auto-reset emulation (softrod_queue_*)

    def set_radius_profile(self, radius):
        self._radius = np.asarray(radius, np.float64).copy()
        if self.ispull or self.ismocto:
            return                      # handed over together with the layers (OracleOcto.pull_setup)
        for r in self.rods:
            r.set_radius_profile(radius)

    def set_muscle_layers(self, ratio_position, strength):
        for r in self.rods:
            if self.ismocto:
                r.body.mocto_setup(self._radius, ratio_position, strength)
            elif self.ispull:
                r.pull_setup(self._radius, ratio_position, strength)
            else:
                r.set_muscle_layers(ratio_position, strength)

    def state(self):
        st = {"time": torch.tensor([r.time for r in self.rods], dtype=torch.float64), "control": self._ctrl}
        if not self.isocto and not self.ismocto:       # softrod_state_view.bc_targets: fixed_position[3], fixed_directors[9] per env
            bc = np.stack([np.concatenate([r.get("fixed_pos"), r.get("fixed_dir").ravel()]) for r in self.rods], axis=1)
            st["bc_targets"] = torch.from_numpy(bc)
            if self.is3d:         # MovingBaseController position x, y; velocity x, y
                st["control"] = torch.from_numpy(np.stack([r.get("control") for r in self.rods], axis=1))
        return st

    def rod_snapshot(self, env_indices):
        rods = [self.rods[i] for i in env_indices]
        return {"x": np.stack([r.get("x") for r in rods]), "v": np.stack([r.get("v") for r in rods]),
                "w": np.stack([r.get("w") for r in rods]), "Q": np.stack([r.get("Q") for r in rods]),
                "time": np.array([r.time for r in rods])}

    def reset(self, theta0, mask=None):
        for i, r in enumerate(self.rods):
            if mask is None or mask[i]:
                r.reset_pendulum(float(theta0[i]))
                if self._queue is not None:
                    self._need[i] = False

    def reset_straight(self, start, direction, normal, mask=None):
        for i, r in enumerate(self.rods):
            if mask is None or mask[i]:
                if self.is3d:
                    self._prev[i] = 0.0   # SoftPendulum3DEnv.reset clears _prev_action
                if self.ispull:
                    r.reset_pull()
                elif self.ispush:
                    r.reset_push()
                elif self.isarm:
                    r.reset_arm()   # also re-arms prev_kappa_state / prev_com_state
                elif self.issoftarm:
                    r.reset_soft_arm()
                    self._ctrl[0, i] = 0.0
                    self._ctrl[1:4, i] = torch.tensor(list(self.cfg.arm_target), dtype=torch.float64)
                else:
                    r.reset_straight(start[i], direction[i], normal[i])

    # -- softrod_autoreset_enable / softrod_queue_* emulated on the host ----------------------
    def autoreset_enable(self, depth):
        from collections import deque

        self.queue_depth = int(depth)
        self._queue = [deque() for _ in range(self.n_envs)]
        self._need = np.zeros(self.n_envs, bool)
        self._consumed = np.zeros(self.n_envs, np.int32)
        self._underflow = 0

    def _push(self, records, counts):
        for i in range(self.n_envs):
            if len(self._queue[i]) + int(counts[i]) > self.queue_depth:
                raise _capi.SoftrodError("reset queue overflow: staged + new records exceed depth")
            self._queue[i].extend(records[i][: int(counts[i])])

    def queue_push(self, theta0, counts):
        th = np.asarray(theta0, np.float64).reshape(self.n_envs, -1)
        self._push([[("theta", t) for t in row] for row in th], counts)

    def queue_push_straight(self, start, direction, normal, counts):
        s, d, nrm = (np.asarray(v, np.float64).reshape(self.n_envs, -1, 3) for v in (start, direction, normal))
        self._push([[("straight", s[i, j], d[i, j], nrm[i, j]) for j in range(s.shape[1])]
                    for i in range(self.n_envs)], counts)

    def queue_push_octo(self, targets, counts):
        tg = np.asarray(targets, np.float64).reshape(self.n_envs, -1, 4 if self.ismocto else 2)
        self._push([[("octo", t) for t in row] for row in tg], counts)

    def queue_status(self):
        return self._consumed.copy(), self._underflow

    def queue_status_begin(self):
        # the counters as of now; delivered one poll late, like a copy still in flight
        self._status = [self._consumed.copy(), self._underflow, 1]

    def queue_status_poll(self, wait=False):
        if self._status[2] > 0 and not wait:
            self._status[2] -= 1
            return None
        return self._status[0], self._status[1]

    def queue_advance(self, by):
        for i, b in enumerate(np.asarray(by)):
            k = len(self._queue[i]) if b < 0 else min(int(b), len(self._queue[i]))
            for _ in range(k):
                self._queue[i].popleft()
            self._consumed[i] += k

    def _autoreset_pass(self):
        """-> indices reset instead of stepped (obs/reward/flags already written)."""
        done = []
        for i in np.nonzero(self._need)[0]:
            if not self._queue[i]:
                self._underflow += 1
                continue
            rec = self._queue[i].popleft()
            self._consumed[i] += 1
            m = np.zeros(self.n_envs, bool)
            m[i] = True
            if rec[0] == "theta":
                th = np.zeros(self.n_envs)
                th[i] = rec[1]
                self.reset(th, m)
            elif rec[0] == "straight":
                z = np.zeros((self.n_envs, 3))
                s, d, nrm = z.copy(), z.copy(), z.copy()
                s[i], d[i], nrm[i] = rec[1], rec[2], rec[3]
                self.reset_straight(s, d, nrm, m)
            else:
                tg = np.zeros((self.n_envs, 4 if self.ismocto else 2))
                tg[i] = rec[1]
                self.reset_octo(tg, m)
            self._need[i] = False
            done.append(i)
        if done:
            keep = self.obs.clone()
            fresh = self.observe(None).clone()
            self.obs[:] = keep
            for i in done:
                self.obs[i] = fresh[i]
                self.reward[i] = 0.0
                self.terminated[i] = 0
                self.truncated[i] = 0
        return done

    def reset_octo(self, targets, mask=None):
        for i, r in enumerate(self.rods):
            if mask is None or mask[i]:
                if self.ismocto:
                    t4 = np.asarray(targets[i], np.float64)
                    self._octo_obs[i] = r.reset(t4[:3], final_time=(t4[3] if t4[3] > 0 else float(self.cfg.final_time)))
                    if self._queue is not None:
                        self._need[i] = False
                    continue
                ob = r.reset(targets[i])
                self._octo_obs[i] = np.concatenate([ob["individual"].ravel(), ob["shared"]])

    def observe(self, prev_action=None):
        if self.ismocto:    # get_state() on the current state; like the reference's it moves ArmTwoEnv's _prev_kappa
            if prev_action is not None:
                raise _capi.SoftrodError("the muscle octopus envs observe with their resident prev_action")
            for i, r in enumerate(self.rods):
                self.obs[i] = torch.from_numpy(r.get_state() if self._octo_obs[i] is None else self._octo_obs[i])
                self._octo_obs[i] = None
            return self.obs
        if self.isocto:     # the oracle keeps _prev_action itself (it survives reset)
            for i in range(self.n_envs):
                self.obs[i] = torch.from_numpy(self._octo_obs[i])
            return self.obs
        pa = self._prev     # resident _prev_action, like softrod_state_view.prev_action
        if prev_action is not None:
            pa = torch.as_tensor(prev_action).reshape(self.n_envs, self.action_dim).numpy()
        for i, r in enumerate(self.rods):
            if self.ispush or self.ispull:
                self.obs[i] = torch.from_numpy(r.observe_pull() if self.ispull else r.observe_push())   # the oracle keeps _prev_action itself
                continue
            if self.is3d:
                o = r.observe3d()
                o[6:8] = pa[i]   # the env owns _prev_action; the rod only sees it at step time
                self.obs[i] = torch.from_numpy(o)
            elif self.issoftarm:
                r.set_arm_target(self._ctrl[1:4, i].numpy())
                self.obs[i] = torch.from_numpy(r.observe_soft_arm().astype(np.float32))
            elif self.isarm:
                # get_state at reset: rates are zero, kappa is zero (straight arm)
                o = np.zeros(25, np.float32)
                c = self.cfg
                o[0:7] = np.float32((0.0 - c.kappa_range[0]) / (c.kappa_range[1] - c.kappa_range[0]))
                o[7:14] = np.float32((0.0 - c.kappa_rate_range[0]) / (c.kappa_rate_range[1] - c.kappa_rate_range[0]))
                o[16:23] = pa[i]
                o[23:25] = [c.target[0], c.target[1]]
                self.obs[i] = torch.from_numpy(o)
            else:
                r.set_prev_action(float(pa[i, 0]))
                self.obs[i] = torch.from_numpy(r.observe())
        return self.obs

    def prev_action_rows(self):
        return torch.from_numpy(self._prev)      # shares memory with the resident copy

    def step(self, actions):
        a = torch.as_tensor(actions, dtype=torch.float32).reshape(self.n_envs, self.action_dim).numpy()
        skip = self._autoreset_pass() if self._queue is not None else []
        for i, r in enumerate(self.rods):
            if i in skip:
                continue
            self._prev[i] = a[i]
            if self.ismocto:
                o, rw, te, tr = r.step(a[i])
                self._octo_obs[i] = None
            elif self.isocto:
                ob, rw, te, tr = r.env_step(a[i])
                o = np.concatenate([ob["individual"].ravel(), ob["shared"]])
                self._octo_obs[i] = o
            elif self.is3d:
                o, rw, te, tr, tilt = r.env_step3d(a[i])
                self.aux[i, 0] = tilt
            elif self.isarm:
                o, rw, te, tr = r.env_step_arm(a[i])
            elif self.ispull:
                o, rw, te, tr = r.env_step_pull(a[i])
            elif self.ispush:
                o, rw, te, tr = r.env_step_push(a[i])
            elif self.issoftarm:
                r.set_arm_target(self._ctrl[1:4, i].numpy())
                o, rw, te, tr = r.env_step_soft_arm(a[i])
                o = o.astype(np.float32)
                self._ctrl[0, i] += int(self.cfg.n_substeps)
            else:
                o, rw, te, tr = r.env_step(a[i, 0])
            self.obs[i] = torch.from_numpy(o)
            self.reward[i] = rw
            self.terminated[i] = int(te)
            self.truncated[i] = int(tr)
            if self._queue is not None:
                self._need[i] = bool(te) or bool(tr)
        return self.obs, self.reward, self.terminated, self.truncated

    def step_packed(self, actions, out=None):
        from gym_softrobot_amd.distributed import pack_outputs

        o, r, te, tr = self.step(actions)
        return pack_outputs(o, r, te, tr, out)

    def close(self):
        self.rods = []
