"""A minimal stand-in for the `gymnasium` package — TEST INFRASTRUCTURE (tests/test_gymnasium_registry.py).

Gymnasium is not installed in the build image, so the branches of gym_softrobot_amd that run only when it
IS (registration into `gymnasium.registry`, `gymnasium.Env` as the env base class, `gymnasium.spaces`)
were never executed.  `install()` puts modules with the slice of Gymnasium's public surface those
branches touch into `sys.modules` — the same way tools/refshim.py stands in for the packages the
reference imports — written from Gymnasium 1.0's documented behaviour:

  gymnasium.registry            dict id -> EnvSpec(id, entry_point, kwargs, max_episode_steps)
  gymnasium.register(...)       adds a spec; a duplicate id is an error
  gymnasium.make(id, **kw)      spec.entry_point(**{**spec.kwargs, **kw}); "module:Class" strings are resolved
  gymnasium.make_vec(id, num_envs, vectorization_mode)   spec.vector_entry_point(num_envs=..., **kwargs)
  gymnasium.Env                 reset(seed=...) seeds `np_random` with Generator(PCG64(SeedSequence(seed)));
                                `unwrapped`, `close`
  gymnasium.spaces.Box / Dict   shape / dtype / low / high, seed, sample, contains
"""
import importlib
import sys
import types

import numpy as np


class EnvSpec:
    def __init__(self, id, entry_point, kwargs=None, max_episode_steps=None, vector_entry_point=None):  # noqa: A002
        self.id, self.entry_point, self.kwargs, self.max_episode_steps = id, entry_point, dict(kwargs or {}), max_episode_steps
        self.vector_entry_point = vector_entry_point

    def make(self, **kw):
        return make(self.id, **kw)


registry = {}


def register(id, entry_point=None, kwargs=None, max_episode_steps=None, vector_entry_point=None, **_):  # noqa: A002
    if id in registry:
        raise ValueError(f"Cannot re-register id: {id}")
    registry[id] = EnvSpec(id, entry_point, kwargs, max_episode_steps, vector_entry_point)


def make_vec(id, num_envs=1, vectorization_mode=None, vector_kwargs=None, **kw):  # noqa: A002
    """Gymnasium 1.0: with a registered `vector_entry_point` (mode None or "vector_entry_point") the batched class
    is instantiated as vector_entry_point(num_envs=num_envs, **spec.kwargs, **kw)."""
    spec = registry[id]
    if spec.vector_entry_point is None or vectorization_mode not in (None, "vector_entry_point"):
        raise NotImplementedError("the stand-in only knows vector_entry_point")
    env = spec.vector_entry_point(num_envs=num_envs, **{**spec.kwargs, **(vector_kwargs or {}), **kw})
    env.unwrapped.spec = spec      # as upstream's make_vec ends (it also reads env.metadata)
    _ = env.metadata
    return env


def make(id, **kw):  # noqa: A002
    spec = registry[id]
    ep = spec.entry_point
    if isinstance(ep, str):
        mod, _, name = ep.partition(":")
        ep = getattr(importlib.import_module(mod), name)
    env = ep(**{**spec.kwargs, **kw})
    env.spec = spec
    return env


class Env:
    metadata = {"render_modes": []}
    render_mode = None
    spec = None
    _np_random = None
    _np_random_seed = None

    def reset(self, *, seed=None, options=None):
        if seed is not None:
            ss = np.random.SeedSequence(seed)
            self._np_random, self._np_random_seed = np.random.Generator(np.random.PCG64(ss)), ss.entropy

    @property
    def np_random(self):
        if self._np_random is None:
            ss = np.random.SeedSequence()
            self._np_random, self._np_random_seed = np.random.Generator(np.random.PCG64(ss)), ss.entropy
        return self._np_random

    @np_random.setter
    def np_random(self, value):
        self._np_random = value

    @property
    def unwrapped(self):
        return self

    def close(self):
        pass


class Space:
    def __init__(self, shape=None, dtype=None, seed=None):
        self._shape = None if shape is None else tuple(shape)
        self.dtype = None if dtype is None else np.dtype(dtype)
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
        ss = np.random.SeedSequence(seed)
        self._np_random = np.random.Generator(np.random.PCG64(ss))
        return ss.entropy

    def __contains__(self, x):
        return self.contains(x)


class Box(Space):
    def __init__(self, low, high, shape=None, dtype=np.float32, seed=None):
        if shape is None:
            shape = np.shape(low)
        super().__init__(shape, dtype, seed)
        full = lambda v: (np.full(self._shape, v, dtype=self.dtype) if np.isscalar(v)   # noqa: E731
                          else np.asarray(v, dtype=self.dtype).reshape(self._shape))
        self.low, self.high = full(low), full(high)
        self.bounded_below, self.bounded_above = np.isfinite(self.low), np.isfinite(self.high)

    def sample(self):
        rng = self.np_random
        out = np.empty(self._shape)
        unb = ~self.bounded_below & ~self.bounded_above
        upp = ~self.bounded_below & self.bounded_above
        low = self.bounded_below & ~self.bounded_above
        bnd = self.bounded_below & self.bounded_above
        out[unb] = rng.normal(size=unb[unb].shape)
        out[low] = rng.exponential(size=low[low].shape) + self.low[low]
        out[upp] = -rng.exponential(size=upp[upp].shape) + self.high[upp]
        out[bnd] = rng.uniform(low=self.low[bnd], high=self.high[bnd], size=bnd[bnd].shape)
        return out.astype(self.dtype)

    def contains(self, x):
        if not isinstance(x, np.ndarray):
            try:
                x = np.asarray(x, dtype=self.dtype)
            except (ValueError, TypeError):
                return False
        return bool(np.can_cast(x.dtype, self.dtype) and x.shape == self._shape
                    and np.all(x >= self.low) and np.all(x <= self.high))


class Discrete(Space):
    """gymnasium.spaces.Discrete: {start, ..., start + n - 1}, shape (), int64."""

    def __init__(self, n, seed=None, start=0):
        super().__init__((), np.int64, seed)
        self.n, self.start = int(n), int(start)

    def sample(self):
        return np.int64(self.start + self.np_random.integers(self.n))

    def contains(self, x):
        if isinstance(x, (int, np.integer)) or (isinstance(x, np.ndarray) and x.shape == () and np.issubdtype(x.dtype, np.integer)):
            return self.start <= int(x) < self.start + self.n
        return False


class Dict(Space):
    def __init__(self, spaces=None, seed=None, **kw):
        super().__init__(None, None, None)
        self.spaces = dict(spaces or {}, **kw)

    def __getitem__(self, k):
        return self.spaces[k]

    def keys(self):
        return self.spaces.keys()

    def seed(self, seed=None):
        return {k: s.seed(None if seed is None else seed + i) for i, (k, s) in enumerate(self.spaces.items())}

    def sample(self):
        return {k: s.sample() for k, s in self.spaces.items()}

    def contains(self, x):
        return isinstance(x, dict) and x.keys() == self.spaces.keys() and all(self.spaces[k].contains(x[k]) for k in self.spaces)


def install():
    """Put the stand-in into sys.modules (before gym_softrobot_amd is imported)."""
    g = types.ModuleType("gymnasium")
    g.__version__ = "1.0.0-standin"
    g.registry, g.register, g.make, g.make_vec, g.Env, g.EnvSpec = registry, register, make, make_vec, Env, EnvSpec
    sp = types.ModuleType("gymnasium.spaces")
    sp.Space, sp.Box, sp.Dict, sp.Discrete = Space, Box, Dict, Discrete
    g.spaces = sp
    envs = types.ModuleType("gymnasium.envs")
    reg = types.ModuleType("gymnasium.envs.registration")
    reg.register, reg.registry, reg.EnvSpec, reg.make = register, registry, EnvSpec, make
    envs.registration = reg
    g.envs = envs
    for name, mod in (("gymnasium", g), ("gymnasium.spaces", sp), ("gymnasium.envs", envs),
                      ("gymnasium.envs.registration", reg)):
        sys.modules[name] = mod
    return g
