"""`Box` space: Gymnasium's when it is installed, otherwise a minimal stand-in with the
subset SoftPendulumEnv uses (soft_pendulum.py:84-94: shape/dtype/low/high, sample,
seed, contains)."""
from __future__ import annotations

import numpy as np

try:  # pragma: no cover - gymnasium is not installed in the build image
    from gymnasium.spaces import Box, Dict, Discrete  # type: ignore
    HAVE_GYMNASIUM = True
except Exception:  # noqa: BLE001
    HAVE_GYMNASIUM = False

    class Discrete:  # type: ignore[no-redef]
        """Subset of gymnasium.spaces.Discrete used by ArmPushEnv (octopus/arm_push_env.py:101):
        {0, ..., n - 1}, shape (), dtype int64."""

        def __init__(self, n, seed=None, start=0):
            self.n, self.start = int(n), int(start)
            self.shape, self.dtype = (), np.dtype(np.int64)
            self._np_random = None
            if seed is not None:
                self.seed(seed)

        @property
        def np_random(self):
            if self._np_random is None:
                self.seed()
            return self._np_random

        def seed(self, seed=None):
            from .seeding import np_random

            self._np_random, s = np_random(seed)
            return s

        def sample(self):
            return np.int64(self.start + self.np_random.integers(self.n))

        def contains(self, x) -> bool:
            if isinstance(x, (int, np.integer)):
                v = int(x)
            elif isinstance(x, np.ndarray) and x.shape == () and np.issubdtype(x.dtype, np.integer):
                v = int(x)
            else:
                return False
            return self.start <= v < self.start + self.n

        def __contains__(self, x):
            return self.contains(x)

        def __repr__(self):
            return f"Discrete({self.n})"

    class Dict:  # type: ignore[no-redef]
        """Subset of gymnasium.spaces.Dict used by FlatEnv (octopus/flat_env.py:100-109)."""

        def __init__(self, spaces):
            self.spaces = dict(spaces)

        def __getitem__(self, key):
            return self.spaces[key]

        def keys(self):
            return self.spaces.keys()

        def sample(self):
            return {k: s.sample() for k, s in self.spaces.items()}

        def contains(self, x) -> bool:
            return isinstance(x, dict) and x.keys() == self.spaces.keys() and all(
                self.spaces[k].contains(x[k]) for k in self.spaces)

        def __contains__(self, x):
            return self.contains(x)

        def __repr__(self):
            return "Dict(" + ", ".join(f"{k!r}: {s!r}" for k, s in self.spaces.items()) + ")"

    class Box:  # type: ignore[no-redef]
        def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
            self.dtype = np.dtype(dtype)
            if shape is None:
                shape = np.shape(low)
            self._shape = tuple(shape)
            self.low = np.full(self._shape, low, dtype=self.dtype) if np.isscalar(low) else np.asarray(low, dtype=self.dtype).reshape(self._shape)
            self.high = np.full(self._shape, high, dtype=self.dtype) if np.isscalar(high) else np.asarray(high, dtype=self.dtype).reshape(self._shape)
            self.bounded_below = np.isfinite(self.low)
            self.bounded_above = np.isfinite(self.high)
            self._np_random = None
            if seed is not None:
                self.seed(seed)

        @property
        def shape(self):
            return self._shape

        @property
        def np_random(self):
            if self._np_random is None:
                self.seed()
            return self._np_random

        def seed(self, seed=None):
            from .seeding import np_random

            self._np_random, s = np_random(seed)
            return s

        def sample(self):
            # Gymnasium Box.sample: uniform for bounded, normal for unbounded dims
            rng = self.np_random
            high = self.high if self.dtype.kind == "f" else self.high.astype("int64") + 1
            sample = np.empty(self._shape)
            unbounded = ~self.bounded_below & ~self.bounded_above
            upp = ~self.bounded_below & self.bounded_above
            low = self.bounded_below & ~self.bounded_above
            bounded = self.bounded_below & self.bounded_above
            sample[unbounded] = rng.normal(size=unbounded[unbounded].shape)
            sample[low] = rng.exponential(size=low[low].shape) + self.low[low]
            sample[upp] = -rng.exponential(size=upp[upp].shape) + high[upp]
            sample[bounded] = rng.uniform(low=self.low[bounded], high=high[bounded], size=bounded[bounded].shape)
            return sample.astype(self.dtype)

        def contains(self, x) -> bool:
            if not isinstance(x, np.ndarray):
                try:
                    x = np.asarray(x, dtype=self.dtype)
                except (ValueError, TypeError):
                    return False
            return bool(
                np.can_cast(x.dtype, self.dtype)
                and x.shape == self._shape
                and np.all(x >= self.low)
                and np.all(x <= self.high)
            )

        def __contains__(self, x):
            return self.contains(x)

        def __repr__(self):
            return f"Box({self.low.min()}, {self.high.max()}, {self._shape}, {self.dtype})"
