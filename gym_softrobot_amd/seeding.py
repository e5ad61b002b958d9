"""Seeding identical to Gymnasium's `gymnasium.utils.seeding.np_random`, which
`SoftPendulumEnv.reset` reaches through `super().reset(seed=seed)`
(gym_softrobot/envs/soft_pendulum/soft_pendulum.py:114) and whose first draw sets the
initial angle (gym_softrobot/envs/soft_pendulum/build.py:47-49)."""
from __future__ import annotations

from typing import Optional, Tuple

import numpy as np


def np_random(seed: Optional[int] = None) -> Tuple[np.random.Generator, int]:
    if seed is not None and not (isinstance(seed, (int, np.integer)) and seed >= 0):
        raise ValueError(f"Seed must be a non-negative integer or None, got {seed!r}")
    seed_seq = np.random.SeedSequence(seed)
    np_seed = seed_seq.entropy
    rng = np.random.Generator(np.random.PCG64(seed_seq))
    return rng, np_seed


def initial_angle(rng: np.random.Generator) -> float:
    """theta0 of build_soft_pendulum (build.py:47-49): one draw per reset."""
    return float(np.deg2rad(90 + (rng.random() - 0.5) * 10))
