"""`render()` for render_mode="rgb_array": a matplotlib (Agg) session over host snapshots of
the resident rods — the counterpart of gym_softrobot/utils/render/matplotlib_renderer.py:177-238
(`Session`: add_rod / add_rigid_body / add_point / render / close; rods drawn as their node
positions sized by the element radius, equal axes).  Drawing happens on the host from a small
device->host copy (HipRodBackend.rod_snapshot); it is not part of the hot path.  POV-Ray and the
pyglet window of render_mode="human" are not provided.
"""
from __future__ import annotations

import enum
from typing import Callable, List, Optional

import numpy as np


class RendererType(enum.Enum):
    """Which renderer `render()` uses (the reference keeps this choice in `RENDERER_CONFIG`,
    gym_softrobot/__init__.py:83); only MATPLOTLIB is provided here."""

    POVRAY = 1
    MATPLOTLIB = 2


class MatplotlibSession:
    def __init__(self, width: int, height: int, dpi: int = 100):
        import matplotlib

        matplotlib.use("Agg", force=False)
        import matplotlib.pyplot as plt

        self._plt = plt
        self.width, self.height, self.dpi = int(width), int(height), int(dpi)
        px = 1.0 / dpi
        self.fig = plt.figure(figsize=(width * px, height * px), frameon=True, dpi=dpi)
        self.ax = self.fig.add_subplot(projection="3d")
        self.ax.set_xlabel("x")
        self.ax.set_ylabel("y")
        self.ax.set_zlabel("z")
        self._rods: List[Callable[[], tuple]] = []
        self._points: List[tuple] = []
        self._artists = []

    def add_rod(self, getter: Callable[[], tuple]) -> None:
        """getter() -> (position (3, n+1), radius (n,)) of the rod as it is now."""
        self._rods.append(getter)

    def add_rigid_body(self, getter: Callable[[], tuple]) -> None:
        """getter() -> (position (3,), radius) of a rigid body drawn as one marker."""
        self._rods.append(lambda: (np.asarray(getter()[0], float).reshape(3, 1), np.array([getter()[1]])))

    def add_point(self, loc, radius: float) -> None:
        self._points.append((np.asarray(loc, float).reshape(3), float(radius)))

    def _equal_axes(self, pts: np.ndarray) -> None:
        lo, hi = pts.min(axis=1), pts.max(axis=1)
        mid, half = 0.5 * (lo + hi), max(0.5 * float((hi - lo).max()), 1e-6)
        self.ax.set_xlim(mid[0] - half, mid[0] + half)
        self.ax.set_ylim(mid[1] - half, mid[1] + half)
        self.ax.set_zlim(mid[2] - half, mid[2] + half)

    def render(self, width: Optional[int] = None, height: Optional[int] = None, **_kw) -> np.ndarray:
        for a in self._artists:
            a.remove()
        self._artists = []
        cloud = []
        for get in self._rods:
            x, r = get()
            x = np.asarray(x, float)
            size = np.full(x.shape[1], float(np.mean(r)))
            size[: len(r)] = r
            scale = 72.0 * self.height / self.dpi          # marker size in points^2 ~ (radius / extent)^2
            self._artists.append(self.ax.plot(x[0], x[1], x[2], color="tab:blue", lw=1.0)[0])
            self._artists.append(self.ax.scatter(x[0], x[1], x[2], s=np.maximum(2.0, (size * scale) ** 2 * 1e-2),
                                                 color="tab:blue", alpha=0.6))
            cloud.append(x)
        for loc, radius in self._points:
            self._artists.append(self.ax.scatter(*loc, s=30.0, color="tab:red"))
            cloud.append(loc.reshape(3, 1))
        if cloud:
            self._equal_axes(np.concatenate(cloud, axis=1))
        self.fig.canvas.draw()
        buf = np.asarray(self.fig.canvas.buffer_rgba())
        return np.ascontiguousarray(buf[..., :3])

    def close(self) -> None:
        self._plt.close(self.fig)
        self._rods.clear()
        self._points.clear()


def _build(env) -> MatplotlibSession:
    """Session for one of the single-env classes (env._vec is its one-env batch)."""
    from . import _capi

    vec = env._vec
    cfg = vec.cfg
    kind = int(cfg.env_kind)
    sess = MatplotlibSession(800, 600)            # maxwidth 800, aspect 3/4 (soft_pendulum.py:261-262)
    r0 = float(cfg.base_radius)
    ne = int(cfg.n_elem)
    if kind == _capi.ENV_OCTO_FLAT or kind in _capi.MUSCLE_OCTOPUS_ENVS:     # arms + head (the muscle octopus: same state layout)
        na = int(cfg.n_arm)
        for a in range(na):
            sess.add_rod(lambda a=a: (vec.backend.octo_state_numpy()["x"][0, a], np.full(ne, r0)))
        sess.add_rigid_body(lambda: (vec.backend.octo_state_numpy()["head_x"][0], float(cfg.head_radius)))
        sess.add_point([vec.targets[0][0], vec.targets[0][1], 0.0], 0.02)
    else:
        sess.add_rod(lambda: (vec.backend.rod_snapshot([0])["x"][0], np.full(ne, r0)))
        if kind == _capi.ENV_ARM_SINGLE:
            sess.add_point([float(cfg.target[0]), float(cfg.target[1]), 0.0], 0.02)
        elif kind == _capi.ENV_SOFT_ARM:
            ctrl = vec.backend.state()["control"]
            sess.add_point([float(ctrl[1, 0]), float(ctrl[2, 0]), float(ctrl[3, 0])], 25.0)
    return sess


def render_env(env):
    """The body of `render()` of the single-env classes: None without a render mode, an
    (H, W, 3) uint8 frame for "rgb_array"; the pyglet window of "human" is not provided."""
    if env.render_mode is None:
        return None
    if env.render_mode == "human":
        raise NotImplementedError('render_mode="human" needs the pyglet viewer, which is not provided; '
                                  'use "rgb_array"')
    if getattr(env, "_render_session", None) is None:
        env._render_session = _build(env)
    return env._render_session.render()


def close_env(env) -> None:
    sess = getattr(env, "_render_session", None)
    if sess is not None:
        sess.close()
        env._render_session = None
