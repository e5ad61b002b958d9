"""Env registry mirroring gym_softrobot/__init__.py:74-76 for the path this package
accelerates.  `make("SoftPendulum-v0", **kwargs)` works without Gymnasium; when
Gymnasium is importable the same ids are also registered there under the
`gym_softrobot_amd/` namespace so `gymnasium.make("gym_softrobot_amd/SoftPendulum-v0")`
resolves to the HIP-backed env (tests/test_gymnasium_registry.py exercises that branch with a
stand-in `gymnasium` module, since Gymnasium itself is not installed in the build image)."""
from __future__ import annotations

from typing import Callable, Dict

_REGISTRY: Dict[str, Dict] = {}
NAMESPACE = "gym_softrobot_amd"

try:
    import gymnasium as _gymnasium
except ImportError:
    _gymnasium = None


def register(id: str, entry_point: Callable, kwargs=None, vector_entry_point: Callable = None,  # noqa: A002
             label: str = None) -> None:
    """`vector_entry_point`: the batched (N envs on one GPU) class of the id — what Gymnasium 1.0's
    `gymnasium.make_vec(id, num_envs=N, vectorization_mode="vector_entry_point")` instantiates.
    `label`: a parity caveat that travels with the id (`parity_label(id)`, `python -m gym_softrobot_amd`,
    bench.py): the COOMM muscle envs carry "parity-unpinned (...)"."""
    _REGISTRY[id] = {"entry_point": entry_point, "kwargs": dict(kwargs or {}), "vector_entry_point": vector_entry_point,
                     "label": label}
    if _gymnasium is not None:
        gid = f"{NAMESPACE}/{id}"
        if gid not in _gymnasium.registry:
            # no max_episode_steps: the reference registers none either, truncation is the env's own
            # (gym_softrobot/__init__.py:74-76, soft_pendulum.py:226-229)
            extra = {} if vector_entry_point is None else {"vector_entry_point": vector_entry_point}
            try:
                _gymnasium.register(id=gid, entry_point=entry_point, kwargs=dict(kwargs or {}), **extra)
            except TypeError:          # a Gymnasium older than 1.0: no vector_entry_point
                _gymnasium.register(id=gid, entry_point=entry_point, kwargs=dict(kwargs or {}))


def make(id: str, **kwargs):  # noqa: A002
    if id not in _REGISTRY:
        raise KeyError(f"unknown env id {id!r}; registered: {sorted(_REGISTRY)}")
    spec = _REGISTRY[id]
    kw = dict(spec["kwargs"])
    kw.update(kwargs)
    return spec["entry_point"](**kw)


def registered() -> list:
    return sorted(_REGISTRY)


def parity_label(id: str):  # noqa: A002
    """None for an env whose whole path is restated from PyElastica's published algorithm and the reference's
    on-disk code; otherwise what is NOT pinned (the COOMM muscle envs)."""
    return _REGISTRY[id].get("label")
