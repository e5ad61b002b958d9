"""Test double for the device backend: same interface as
gym_softrobot_amd.backend.HipRodBackend, arithmetic by the CPU oracle.  Lives under
tests/ on purpose — it lets the CPU suite exercise the host logic (env classes,
seeding, sharding, packed all-gather) without a GPU.  It is never importable from the
product package."""
from __future__ import annotations

import numpy as np
import torch

from oracle import oracle_c


class OracleBackend:
    def __init__(self, cfg, omp: bool = False):
        self.cfg = cfg.copy()
        self.n_envs = int(cfg.n_envs)
        self.device = torch.device("cpu")
        self.rods = [oracle_c.OracleRod(self.cfg, omp=omp) for _ in range(self.n_envs)]
        self.obs = torch.zeros((self.n_envs, 4), dtype=torch.float32)
        self.reward = torch.zeros(self.n_envs, dtype=torch.float64)
        self.terminated = torch.zeros(self.n_envs, dtype=torch.uint8)
        self.truncated = torch.zeros(self.n_envs, dtype=torch.uint8)

    def reset(self, theta0, mask=None):
        for i, r in enumerate(self.rods):
            if mask is None or mask[i]:
                r.reset_pendulum(float(theta0[i]))

    def observe(self, prev_action=None):
        for i, r in enumerate(self.rods):
            r.set_prev_action(0.0 if prev_action is None else float(prev_action[i]))
            self.obs[i] = torch.from_numpy(r.observe())
        return self.obs

    def step(self, actions):
        a = torch.as_tensor(actions, dtype=torch.float32).reshape(self.n_envs).numpy()
        for i, r in enumerate(self.rods):
            o, rw, te, tr = r.env_step(a[i])
            self.obs[i] = torch.from_numpy(o)
            self.reward[i] = rw
            self.terminated[i] = int(te)
            self.truncated[i] = int(tr)
        return self.obs, self.reward, self.terminated, self.truncated

    def close(self):
        self.rods = []
