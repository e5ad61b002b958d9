"""Device backend: one `softrod_handle` (include/softrod.h) per GPU, fed with torch-ROCm
tensors.  torch is plumbing only (device memory, streams); all arithmetic of the hot
path happens in libsoftrod_hip.so.  There is no CPU fallback here by design."""
from __future__ import annotations

import ctypes as C
from typing import Dict, Optional, Tuple

import numpy as np
import torch

from . import _capi
from ._capi import LANE_STRIDE, SoftrodConfig, SoftrodStateView, check, load_library


class _DevArray:
    """Borrowed device memory exposed through __cuda_array_interface__ (zero-copy)."""

    def __init__(self, ptr: int, shape, typestr: str, owner):
        self._owner = owner  # keeps the handle alive
        self.__cuda_array_interface__ = {
            "shape": tuple(shape),
            "typestr": typestr,
            "data": (int(ptr), False),
            "version": 2,
            "strides": None,
        }


class HipRodBackend:
    """Resident batch of rods on one MI355X."""

    def __init__(self, cfg: SoftrodConfig, device: int = 0):
        if not torch.cuda.is_available():
            raise _capi.SoftrodError(
                "HipRodBackend needs a ROCm device (torch.cuda.is_available() is False); "
                "the hot path has no CPU fallback"
            )
        self._lib = load_library()
        self.cfg = cfg.copy()
        self.n_envs = int(cfg.n_envs)
        self.action_dim = _capi.config_action_dim(cfg)
        self.obs_dim = _capi.config_obs_dim(cfg)
        # FlatEnv's host API (reset_octo, Dict observations); the muscle arm with a weight (ENV_ARM_PULL_WEIGHT) has a
        # rigid body too but resets and observes like any single rod
        self.is_octo = bool(cfg.features & _capi.FEAT_OCTO_HEAD) and int(cfg.env_kind) == _capi.ENV_OCTO_FLAT
        # the muscle octopus envs (CrawlEnv / ArmTwoEnv / ReachEnv): FlatEnv's reset API with their own arm frames,
        # three numbers per target
        self.is_mocto = int(cfg.env_kind) in _capi.MUSCLE_OCTOPUS_ENVS
        self.aux_dim = _capi.aux_dim(cfg.env_kind)
        self.device_index = int(device)
        self.device = torch.device("cuda", self.device_index)
        self._h = C.c_void_p()
        # per-handle physics tables that live outside softrod_config (action basis, spline table,
        # radius profile): their bytes go into config_fingerprint()
        self._tables: Dict[str, bytes] = {}
        check(self._lib.softrod_create(C.byref(self.cfg), self.device_index, C.byref(self._h)))
        if self.cfg.features & _capi.FEAT_REST_KAPPA_ACTION:
            if self.is_octo:
                basis = _capi.octo_action_basis(int(cfg.n_elem), int(cfg.n_knots))
            else:
                basis = _capi.action_basis(int(cfg.n_elem), self.action_dim)
            check(self._lib.softrod_set_action_basis(self._h, basis.ctypes.data), self._h)
            self._tables["action_basis"] = basis.tobytes()
        if int(cfg.env_kind) == _capi.ENV_ARM_TWO:
            basis, _ = _capi.arm_two_activation_basis(int(cfg.n_elem), 3)
            check(self._lib.softrod_set_action_basis(self._h, basis.ctypes.data), self._h)
            self._tables["action_basis"] = basis.tobytes()
        if self.cfg.features & _capi.FEAT_SPLINE_MUSCLE_TORQUES:
            breaks, coef = _capi.spline_table(float(cfg.base_length), int(cfg.n_ctrl))
            if len(breaks) != int(cfg.n_spline_pieces) + 1:
                raise _capi.SoftrodError("n_spline_pieces does not match the interpolant")
            check(self._lib.softrod_set_spline_table(self._h, breaks.ctypes.data, coef.ctypes.data), self._h)
            self._tables["spline_table"] = breaks.tobytes() + coef.tobytes()
        n = self.n_envs
        with torch.cuda.device(self.device):
            self.obs = torch.empty((n, self.obs_dim), dtype=torch.float32, device=self.device)
            self.reward = torch.empty((n,), dtype=torch.float64, device=self.device)
            self.terminated = torch.empty((n,), dtype=torch.uint8, device=self.device)
            self.truncated = torch.empty((n,), dtype=torch.uint8, device=self.device)
            self.aux = (
                torch.empty((n, self.aux_dim), dtype=torch.float64, device=self.device)
                if self.aux_dim else None
            )

    # -- lifetime -------------------------------------------------------------------
    def close(self) -> None:
        if getattr(self, "_h", None) is not None and self._h:
            self._lib.softrod_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass

    def _stream(self) -> C.c_void_p:
        return C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    def _actions(self, actions) -> torch.Tensor:
        a = torch.as_tensor(actions, dtype=torch.float32, device=self.device).reshape(-1)
        if a.numel() != self.n_envs * self.action_dim:
            raise ValueError(f"expected {self.n_envs}x{self.action_dim} actions, got {a.numel()}")
        return a.contiguous()

    # -- C-ABI calls ----------------------------------------------------------------
    def set_radius_profile(self, radius) -> None:
        """CosseratRod.straight_rod(base_radius=<array>): a tapered rod (octopus/arm_push_env.py:
        160-179).  Before the first reset."""
        r = np.ascontiguousarray(radius, dtype=np.float64).reshape(int(self.cfg.n_elem))
        check(self._lib.softrod_set_radius_profile(self._h, r.ctypes.data), self._h)
        self._tables["radius_profile"] = r.tobytes()

    def set_muscle_layers(self, ratio_position, strength) -> None:
        """The layers handed to COOMM's ApplyMuscles (softrod_set_muscle_layers): ratio_position
        (n_muscles, 3, n_elem), strength = max_muscle_stress * rest_muscle_area (n_muscles, n_elem), signed."""
        m, n = int(self.cfg.n_muscles), int(self.cfg.n_elem)
        rp = np.ascontiguousarray(ratio_position, dtype=np.float64).reshape(m, 3, n)
        st = np.ascontiguousarray(strength, dtype=np.float64).reshape(m, n)
        check(self._lib.softrod_set_muscle_layers(self._h, rp.ctypes.data, st.ctypes.data), self._h)
        self._tables["muscle_layers"] = rp.tobytes() + st.tobytes()

    def reset(self, theta0: np.ndarray, mask: Optional[np.ndarray] = None) -> None:
        th = np.ascontiguousarray(theta0, dtype=np.float64).reshape(self.n_envs)
        m = None
        if mask is not None:
            m = np.ascontiguousarray(mask, dtype=np.uint8).reshape(self.n_envs)
        check(
            self._lib.softrod_reset(
                self._h, th.ctypes.data, m.ctypes.data if m is not None else None, self._stream()
            ),
            self._h,
        )

    def reset_straight(self, start, direction, normal, mask: Optional[np.ndarray] = None) -> None:
        arrs = [
            np.ascontiguousarray(np.broadcast_to(np.asarray(v, np.float64), (self.n_envs, 3)))
            for v in (start, direction, normal)
        ]
        m = None
        if mask is not None:
            m = np.ascontiguousarray(mask, dtype=np.uint8).reshape(self.n_envs)
        check(
            self._lib.softrod_reset_straight(
                self._h, arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data,
                m.ctypes.data if m is not None else None, self._stream(),
            ),
            self._h,
        )

    def _arm_frames(self):
        if self.is_mocto:
            pos, dirs, _ = _capi.muscle_octopus_arm_frames(int(self.cfg.env_kind), float(self.cfg.head_radius))
            return pos, dirs
        return _capi.octo_arm_frames(int(self.cfg.n_arm), float(self.cfg.head_radius))

    def reset_octo(self, targets, mask: Optional[np.ndarray] = None) -> None:
        """FlatEnv.reset: targets (n_envs, 2); the arm frames are build_octopus's.  The muscle octopus envs: targets
        (n_envs, 4) — x, y, z and the episode's final_time (0: the config's) —, the frames of build_octopus_muscles / build_two_arms."""
        na = int(self.cfg.n_arm)
        pos, dirs = self._arm_frames()
        pos = np.ascontiguousarray(np.broadcast_to(pos, (self.n_envs, na, 3)))
        dirs = np.ascontiguousarray(np.broadcast_to(dirs, (self.n_envs, na, 3)))
        tg = np.ascontiguousarray(targets, dtype=np.float64).reshape(self.n_envs, 4 if self.is_mocto else 2)
        m = None
        if mask is not None:
            m = np.ascontiguousarray(mask, dtype=np.uint8).reshape(self.n_envs)
        check(
            self._lib.softrod_reset_octo(
                self._h, pos.ctypes.data, dirs.ctypes.data, tg.ctypes.data,
                m.ctypes.data if m is not None else None, self._stream(),
            ),
            self._h,
        )

    # -- device-side auto-reset (softrod_autoreset_enable / softrod_queue_*) ---------------------
    def autoreset_enable(self, depth: int) -> None:
        check(self._lib.softrod_autoreset_enable(self._h, int(depth)), self._h)
        self.queue_depth = int(depth)

    @staticmethod
    def _counts(counts, n):
        return np.ascontiguousarray(counts, dtype=np.int32).reshape(n)

    def queue_push(self, theta0, counts) -> None:
        th = np.ascontiguousarray(theta0, dtype=np.float64).reshape(self.n_envs, -1)
        c = self._counts(counts, self.n_envs)
        check(self._lib.softrod_queue_push(self._h, th.ctypes.data, c.ctypes.data, th.shape[1], self._stream()),
              self._h)

    def queue_push_straight(self, start, direction, normal, counts) -> None:
        arrs = [np.ascontiguousarray(v, dtype=np.float64).reshape(self.n_envs, -1, 3)
                for v in (start, direction, normal)]
        c = self._counts(counts, self.n_envs)
        check(self._lib.softrod_queue_push_straight(
            self._h, arrs[0].ctypes.data, arrs[1].ctypes.data, arrs[2].ctypes.data, c.ctypes.data,
            arrs[0].shape[1], self._stream()), self._h)

    def queue_push_octo(self, targets, counts) -> None:
        na = int(self.cfg.n_arm)
        tg = np.ascontiguousarray(targets, dtype=np.float64).reshape(self.n_envs, -1, 4 if self.is_mocto else 2)
        m = tg.shape[1]
        pos, dirs = self._arm_frames()
        pos = np.ascontiguousarray(np.broadcast_to(pos, (self.n_envs, m, na, 3)))
        dirs = np.ascontiguousarray(np.broadcast_to(dirs, (self.n_envs, m, na, 3)))
        c = self._counts(counts, self.n_envs)
        check(self._lib.softrod_queue_push_octo(
            self._h, pos.ctypes.data, dirs.ctypes.data, tg.ctypes.data, c.ctypes.data, m, self._stream()),
            self._h)

    def queue_status(self):
        """(consumed[n_envs], underflow) — synchronises the stream."""
        cons = np.zeros(self.n_envs, np.int32)
        uf = C.c_int32(0)
        check(self._lib.softrod_queue_status(self._h, cons.ctypes.data, C.byref(uf), self._stream()), self._h)
        return cons, int(uf.value)

    def queue_status_begin(self) -> None:
        """Start a non-blocking read of the queue counters (softrod_queue_status_begin)."""
        check(self._lib.softrod_queue_status_begin(self._h, self._stream()), self._h)

    def queue_status_poll(self, wait: bool = False):
        """None while the read is in flight, else (consumed[n_envs], underflow) as of when it was
        started; `wait` blocks until that read (not the whole stream) has arrived."""
        cons = np.zeros(self.n_envs, np.int32)
        uf = C.c_int32(0)
        rc = self._lib.softrod_queue_status_poll(self._h, int(bool(wait)), cons.ctypes.data, C.byref(uf))
        if rc < 0:
            check(rc, self._h)
        return None if rc == 0 else (cons, int(uf.value))

    def queue_advance(self, by) -> None:
        """Mark by[e] staged records of env e as used (by[e] < 0: all of them)."""
        b = np.ascontiguousarray(by, dtype=np.int32).reshape(self.n_envs)
        check(self._lib.softrod_queue_advance(self._h, b.ctypes.data, self._stream()), self._h)

    def observe(self, prev_action: Optional[torch.Tensor] = None) -> torch.Tensor:
        pa = None
        if prev_action is not None:
            pa = self._actions(prev_action)
        check(
            self._lib.softrod_observe(
                self._h, pa.data_ptr() if pa is not None else None, self.obs.data_ptr(), self._stream()
            ),
            self._h,
        )
        return self.obs

    def step(self, actions) -> Tuple[torch.Tensor, torch.Tensor, torch.Tensor, torch.Tensor]:
        a = self._actions(actions)
        check(
            self._lib.softrod_step(
                self._h, a.data_ptr(), self.obs.data_ptr(), self.reward.data_ptr(),
                self.terminated.data_ptr(), self.truncated.data_ptr(),
                self.aux.data_ptr() if self.aux is not None else None, self._stream(),
            ),
            self._h,
        )
        return self.obs, self.reward, self.terminated, self.truncated

    def step_packed(self, actions, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        """softrod_step_packed: (n_envs, packed_width(obs_dim)) float32 words per env, into
        `out` if given (the overlapped multi-GPU path alternates two buffers)."""
        from .distributed import packed_width

        a = self._actions(actions)
        if out is None:
            if getattr(self, "packed", None) is None:
                self.packed = torch.empty((self.n_envs, packed_width(self.obs_dim)), dtype=torch.float32,
                                          device=self.device)
            out = self.packed
        check(
            self._lib.softrod_step_packed(
                self._h, a.data_ptr(), out.data_ptr(),
                self.aux.data_ptr() if self.aux is not None else None, self._stream(),
            ),
            self._h,
        )
        return out

    def scatter_rows(self, packed: torch.Tensor, peer_ptrs, first_row: int, tag_word: int = -1, tag: int = 0) -> None:
        """softrod_scatter_rows: this batch's packed rows into rows first_row.. of every buffer in
        `peer_ptrs` (device pointers: the ranks' exchange buffers), one small kernel on the current
        stream; tag_word >= 0: word `tag_word` of every buffer receives `tag` once all rows have landed."""
        tab = np.ascontiguousarray(peer_ptrs, dtype=np.uint64)
        check(
            self._lib.softrod_scatter_rows(
                self._h, packed.data_ptr(), tab.ctypes.data, int(tab.size), int(packed.shape[1]), int(first_row),
                int(tag_word), int(tag) & 0xFFFFFFFF, self._stream(),
            ),
            self._h,
        )

    # -- exchange buffers of ShardedVecEnv(transport="p2p") (softrod_exchange_*) ------------------
    def exchange_alloc(self, n_words: int):
        """Uncached device memory the other GPUs may store into: (tensor float32[n_words] — a
        zero-copy view —, device pointer, 64-byte IPC handle, memory kind)."""
        ptr, kind = C.c_void_p(), C.c_int()
        handle = (C.c_uint8 * 64)()
        check(self._lib.softrod_exchange_alloc(self.device_index, int(n_words) * 4, C.byref(ptr), handle, C.byref(kind)))
        t = torch.as_tensor(_DevArray(ptr.value, (int(n_words),), "<f4", self), device=self.device)
        return t, int(ptr.value), bytes(handle), {1: "uncached", 2: "fine-grained"}[int(kind.value)]

    def exchange_open(self, handle: bytes, owner_device: int) -> int:
        """Map a peer process's exchange buffer; -> the device pointer valid in this process."""
        ptr = C.c_void_p()
        buf = (C.c_uint8 * 64).from_buffer_copy(handle)
        check(self._lib.softrod_exchange_open(self.device_index, buf, int(owner_device), C.byref(ptr)))
        return int(ptr.value)

    def exchange_close(self, ptr: int) -> None:
        check(self._lib.softrod_exchange_close(self.device_index, C.c_void_p(ptr)))

    def exchange_free(self, ptr: int) -> None:
        check(self._lib.softrod_exchange_free(self.device_index, C.c_void_p(ptr)))

    def substeps(self, actions, n: int) -> None:
        a = self._actions(actions) if actions is not None else None
        check(
            self._lib.softrod_substeps(
                self._h, a.data_ptr() if a is not None else None, int(n), self._stream()
            ),
            self._h,
        )

    def set_timing(self, n_launches: int) -> None:
        """Record HIP events around the next `n_launches` kernels (0 = off)."""
        check(self._lib.softrod_set_timing(self._h, int(n_launches)), self._h)

    def kernel_times_ms(self) -> np.ndarray:
        """Durations of the launches timed since set_timing (synchronises)."""
        cap = 1 << 16
        out = np.zeros(cap, np.float32)
        cnt = C.c_int()
        check(self._lib.softrod_kernel_times_ms(self._h, out.ctypes.data, cap, C.byref(cnt)), self._h)
        return out[: cnt.value].copy()

    def last_kernel_ms(self) -> float:
        ms = C.c_float()
        check(self._lib.softrod_last_kernel_ms(self._h, C.byref(ms)), self._h)
        return float(ms.value)

    def prev_action_rows(self) -> torch.Tensor:
        """Writable (n_envs, action_dim) view of the resident `_prev_action`."""
        if getattr(self, "_prev_rows", None) is None:
            self._prev_rows = self.state()["prev_action"]
            if not self.is_octo and not self.is_mocto:
                self._prev_rows = self._prev_rows[:, : self.action_dim]
        return self._prev_rows

    def state(self) -> Dict[str, torch.Tensor]:
        """Zero-copy torch views of the resident SoA state (softrod_state_view)."""
        v = SoftrodStateView()
        check(self._lib.softrod_state_view_get(self._h, C.byref(v)), self._h)
        n, s = self.n_envs, int(v.lane_stride)

        def view(ptr, comps):
            return torch.as_tensor(_DevArray(ptr, (comps, n, s), "<f8", self), device=self.device)

        extra = {}
        ns = n * (int(self.cfg.n_arm) if self.is_mocto else 1)      # SuckerControllers: per env, per arm of the muscle octopus
        if v.env_aux:
            extra["env_aux"] = torch.as_tensor(_DevArray(v.env_aux, (8, n), "<f8", self), device=self.device)
            extra["prev_kappa"] = torch.as_tensor(
                _DevArray(v.prev_kappa, (n, int(self.cfg.n_arm) * (int(self.cfg.n_elem) - 1)), "<f4", self), device=self.device)
        if v.muscle_activation:
            extra["muscle_activation"] = torch.as_tensor(
                _DevArray(v.muscle_activation, (_capi.MAX_MUSCLES, n, s), "<f8", self), device=self.device)
        return {
            **extra,
            "sucker_index": torch.as_tensor(_DevArray(v.sucker_index, (_capi.MAX_SUCKERS, ns), "<i4", self),
                                            device=self.device),
            "position": view(v.position, 3),
            "velocity": view(v.velocity, 3),
            "director": view(v.director, 9),
            "omega": view(v.omega, 3),
            "tangents": view(v.tangents, 3),
            "time": torch.as_tensor(_DevArray(v.time, (n,), "<f8", self), device=self.device),
            "control": torch.as_tensor(_DevArray(v.control, (4, n), "<f8", self), device=self.device),
            "kappa": view(v.kappa, 3),
            "rest_kappa": view(v.rest_kappa, 3),
            "env_memory": torch.as_tensor(_DevArray(v.env_memory, (n, s), "<f8", self), device=self.device),
            "prev_action": torch.as_tensor(_DevArray(v.prev_action, (n, max(7, self.action_dim)), "<f4", self),
                                           device=self.device),
            "head": torch.as_tensor(_DevArray(v.head, (20, n), "<f8", self), device=self.device),
            "bc_targets": torch.as_tensor(_DevArray(v.bc_targets, (12, n), "<f8", self), device=self.device),
            "sucker_ratio": torch.as_tensor(_DevArray(v.sucker_ratio, (_capi.MAX_SUCKERS, ns), "<f8", self),
                                            device=self.device),
            "arm_stride": int(v.arm_stride),
        }

    _SNAPSHOT_KEYS = ("position", "velocity", "director", "omega", "tangents", "time", "control", "kappa",
                      "rest_kappa", "env_memory", "prev_action", "head", "bc_targets", "sucker_ratio", "sucker_index")

    def _snapshot_keys(self):
        return (self._SNAPSHOT_KEYS + (("muscle_activation",) if self.cfg.features & _capi.FEAT_COOMM_MUSCLES else ())
                + (("env_aux", "prev_kappa") if self.is_mocto else ()))

    def config_fingerprint(self) -> bytes:
        """What a snapshot is only valid for: the ABI, every field of softrod_config except the
        batch size (checked through the shapes), and a digest of the per-handle tables that hold
        physics outside the config — the radius profile of a tapered rod (masses, stiffnesses,
        damping per lane), the spline table and the action basis.  A snapshot taken under another
        dt, substep count, feature set, material or taper continues with the wrong physics
        otherwise."""
        import hashlib

        c = self.cfg.copy()
        c.n_envs = 0
        d = hashlib.sha256()
        for k in sorted(self._tables):
            d.update(k.encode() + b"\0" + self._tables[k])
        return bytes([_capi.ABI_VERSION]) + bytes(memoryview(c).cast("B")) + d.digest()

    def snapshot(self) -> Dict[str, torch.Tensor]:
        """Host copy of the whole resident batch (every array of softrod_state_view): what
        `restore` needs to put the batch back exactly — checkpoint / resume, or branching a
        rollout.  The reference has no counterpart (its env state is never serialised)."""
        if getattr(self, "queue_depth", 0):
            raise _capi.SoftrodError("snapshot/restore of a handle with device-side auto-reset is not supported: "
                                     "the pending-reset flags and the staged queue are not part of the state view")
        st = self.state()
        torch.cuda.synchronize(self.device)
        snap = {k: st[k].cpu().clone() for k in self._snapshot_keys()}
        snap["config_fingerprint"] = torch.frombuffer(bytearray(self.config_fingerprint()), dtype=torch.uint8).clone()
        return snap

    def restore(self, snap: Dict[str, torch.Tensor]) -> None:
        if getattr(self, "queue_depth", 0):
            raise _capi.SoftrodError("snapshot/restore of a handle with device-side auto-reset is not supported: "
                                     "stale needs_reset / skip flags would reset or skip the wrong envs")
        fp = snap.get("config_fingerprint")
        if fp is None or bytes(fp.numpy().tobytes()) != self.config_fingerprint():
            raise ValueError("snapshot was taken under a different softrod_config / ABI version / table set "
                             "(dt, n_substeps, features, env_kind, material, radius profile, spline table, "
                             "action basis ...): refusing to load it")
        st = self.state()
        for k in self._snapshot_keys():
            if tuple(snap[k].shape) != tuple(st[k].shape):
                raise ValueError(f"snapshot field {k!r} has shape {tuple(snap[k].shape)}, expected {tuple(st[k].shape)}")
            st[k].copy_(snap[k].to(self.device))
        torch.cuda.synchronize(self.device)

    def rod_snapshot(self, env_indices) -> Dict[str, np.ndarray]:
        """Host copy of a few rods only (diagnostic taps): x, v (k,3,n+1); Q (k,3,3,n);
        w (k,3,n); time (k,).  One small device->host copy per field."""
        st = self.state()
        idx = torch.as_tensor(list(env_indices), dtype=torch.long, device=self.device)
        ne = int(self.cfg.n_elem)

        def rows(name, width):
            return st[name].index_select(1, idx)[:, :, :width].permute(1, 0, 2).cpu().numpy()

        return {
            "x": rows("position", ne + 1), "v": rows("velocity", ne + 1), "w": rows("omega", ne),
            "Q": rows("director", ne).reshape(len(env_indices), 3, 3, ne),
            "time": st["time"].index_select(0, idx).cpu().numpy(),
        }

    def kernel_tier(self) -> str:
        """softrod_kernel_tier: which step kernel this batch runs (follows from the config alone; the
        A/B switches in the environment count only under SOFTROD_DEBUG_SWITCHES=1)."""
        return self._lib.softrod_kernel_tier(self._h).decode()

    def octo_state_numpy(self) -> Dict[str, np.ndarray]:
        """Host copy of an OctoFlat batch: arms as x,v (N,A,3,n+1), Q (N,A,3,3,n), w (N,A,3,n),
        kappa/rest_kappa (N,A,3,n-1); head as x,v,w (N,3), Q (N,3,3); target (N,2); time (N,)."""
        st = self.state()
        ne, na, seg = int(self.cfg.n_elem), int(self.cfg.n_arm), int(st["arm_stride"])
        torch.cuda.synchronize(self.device)

        def arms(t, comps, width):
            a = t[:, :, : na * seg].reshape(comps, self.n_envs, na, seg)[..., :width]
            return a.permute(1, 2, 0, 3).cpu().numpy()

        hd = st["head"].cpu().numpy()
        return {
            "x": arms(st["position"], 3, ne + 1),
            "v": arms(st["velocity"], 3, ne + 1),
            "w": arms(st["omega"], 3, ne),
            "Q": arms(st["director"], 9, ne).reshape(self.n_envs, na, 3, 3, ne),
            "kappa": arms(st["kappa"], 3, ne - 1),
            "rest_kappa": arms(st["rest_kappa"], 3, ne - 1),
            "head_x": hd[0:3].T.copy(), "head_v": hd[3:6].T.copy(),
            "head_Q": hd[6:15].T.reshape(self.n_envs, 3, 3).copy(), "head_w": hd[15:18].T.copy(),
            "target": hd[18:20].T.copy(),
            "time": st["time"].cpu().numpy(),
        }

    def state_numpy(self) -> Dict[str, np.ndarray]:
        """Host copy in the reference's per-rod shapes: x,v (N,3,n+1); Q (N,3,3,n); w (N,3,n)."""
        st = self.state()
        ne = int(self.cfg.n_elem)
        torch.cuda.synchronize(self.device)
        out = {
            "x": st["position"][:, :, : ne + 1].permute(1, 0, 2).cpu().numpy(),
            "v": st["velocity"][:, :, : ne + 1].permute(1, 0, 2).cpu().numpy(),
            "w": st["omega"][:, :, :ne].permute(1, 0, 2).cpu().numpy(),
            "tangents": st["tangents"][:, :, :ne].permute(1, 0, 2).cpu().numpy(),
            "time": st["time"].cpu().numpy(),
            "control": st["control"].permute(1, 0).cpu().numpy(),
            "kappa": st["kappa"][:, :, : ne - 1].permute(1, 0, 2).cpu().numpy(),
            "rest_kappa": st["rest_kappa"][:, :, : ne - 1].permute(1, 0, 2).cpu().numpy(),
        }
        q = st["director"][:, :, :ne].permute(1, 0, 2).cpu().numpy()
        out["Q"] = q.reshape(self.n_envs, 3, 3, ne)
        return out
