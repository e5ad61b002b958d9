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


def register(id: str, entry_point: Callable, kwargs=None) -> None:  # noqa: A002
    _REGISTRY[id] = {"entry_point": entry_point, "kwargs": dict(kwargs or {})}
    if _gymnasium is not None:
        gid = f"{NAMESPACE}/{id}"
        if gid not in _gymnasium.registry:
            # no max_episode_steps: the reference registers none either, truncation is the env's own
            # (gym_softrobot/__init__.py:74-76, soft_pendulum.py:226-229)
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
