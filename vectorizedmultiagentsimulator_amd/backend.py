"""``HipWorld`` - the packed, GPU-resident world state and the calls into libvmas_hip.so.

PyTorch is plumbing only: it owns the device buffers and provides the stream.  All
physics runs in the hand-written HIP kernels behind the C ABI (include/vmas_hip.h);
there is no torch-op or CPU fallback anywhere in this module.

Packed layout (see include/vmas_hip.h):
    state[E, 6, ld]     f: pos.x pos.y vel.x vel.y rot ang_vel
    agent_ft[A, 3, ld]  f: force.x force.y torque
with the environment index fastest, ``ld`` a multiple of 64.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional

import torch

from . import _abi as A
from .spec import WorldSpec


class VmasHipError(RuntimeError):
    pass


def _round_up(x: int, m: int) -> int:
    return (x + m - 1) // m * m


_PARKED: list = []  # releases (callables: a world, a pinned host word) deferred while a HIP-graph capture was in progress


def _drain_parked():
    if not _PARKED:
        return
    try:
        if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
            return
    except Exception:
        return
    while _PARKED:
        _PARKED.pop()()


def release_later(free) -> None:
    """Run ``free()`` (a hipFree / hipHostFree behind the C ABI) now, or - while any stream is being captured into a HIP graph,
    where such a call invalidates the capture, and the garbage collector may run a ``__del__`` at any point - at the next
    release that happens outside a capture."""
    _PARKED.append(free)
    _drain_parked()


class HipWorld:
    """One world description instantiated for ``batch`` environments on one GPU."""

    def __init__(self, spec: WorldSpec, batch: int, device="cuda:0", lanes_per_env: int = 0,
                 state: Optional[torch.Tensor] = None, agent_ft: Optional[torch.Tensor] = None):
        self.lib = A.load_library()  # raises VmasHipLibraryMissing - no fallback
        self.spec = spec
        self.batch = int(batch)
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise VmasHipError(f"HipWorld needs a GPU device, got {self.device}")
        if not torch.cuda.is_available():
            raise VmasHipError("HipWorld: no GPU visible (torch.cuda.is_available() is False)")
        self.device_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self.ld = A.leading_dim(self.batch)
        self.cdesc = spec.to_ctypes()
        handle = C.c_void_p()
        with torch.cuda.device(self.device_index):
            rc = self.lib.vmas_world_create(C.byref(self.cdesc.world), self.batch, self.device_index, C.byref(handle))
        if rc != 0:
            raise VmasHipError(A.last_error())
        self._h = handle
        dev = torch.device("cuda", self.device_index)
        if state is None:
            state = torch.zeros(spec.n_entities, A.STATE_FIELDS, self.ld, device=dev, dtype=torch.float32)
        if agent_ft is None:
            agent_ft = torch.zeros(max(spec.n_agents, 1), A.AGENT_FIELDS, self.ld, device=dev, dtype=torch.float32)
        # caller-owned packed buffers (core.World) are adopted, not copied
        assert state.shape == (spec.n_entities, A.STATE_FIELDS, self.ld) and state.is_contiguous()
        assert agent_ft.shape == (max(spec.n_agents, 1), A.AGENT_FIELDS, self.ld) and agent_ft.is_contiguous()
        assert state.dtype == torch.float32 and state.device == dev and agent_ft.device == dev
        self.state, self.agent_ft = state, agent_ft
        self._mask = torch.zeros(max((len(spec.pairs) + 31) // 32, 1), device=dev, dtype=torch.int32)
        self._lidar_out: Optional[torch.Tensor] = None
        if lanes_per_env:
            self.set_lanes_per_env(lanes_per_env)
        if spec.lidars:
            rc = self.lib.vmas_world_set_lidars(self._h, self.cdesc.lidars, len(spec.lidars))
            if rc != 0:
                raise VmasHipError(A.last_error())
            self._lidar_out = torch.zeros(len(spec.lidars), self.cdesc.max_rays, self.ld, device=dev, dtype=torch.float32)

    # ------------------------------------------------------------------ lifecycle
    def close(self):
        """Free the native world.  ``hipFree`` is illegal while any stream of the process is being
        captured into a HIP graph (it invalidates the capture), and the garbage collector can run a
        ``__del__`` at any point - so a handle released during a capture is parked and freed by the
        next ``close()`` / constructor that runs outside one."""
        h, self._h = getattr(self, "_h", None), None
        if h:
            lib = self.lib
            _PARKED.append(lambda: lib.vmas_world_destroy(h))
        _drain_parked()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ------------------------------------------------------------------ knobs
    def set_lanes_per_env(self, lanes: int):
        if self.lib.vmas_world_set_lanes_per_env(self._h, int(lanes)) != 0:
            raise VmasHipError(A.last_error())

    def set_queues(self, queues: int):
        """``step_n`` over 0 (library's choice) / 1 / 2 HIP queues (include/vmas_hip.h, vmas_world_set_queues)."""
        if self.lib.vmas_world_set_queues(self._h, int(queues)) != 0:
            raise VmasHipError(A.last_error())

    def queues(self, n_steps: int) -> int:
        """How many queues a ``step_n`` of ``n_steps`` steps uses."""
        return int(self.lib.vmas_world_get_queues(self._h, int(n_steps)))

    def set_specialized(self, on: bool):
        """Allow / forbid the world-specialised kernel for plain World.step launches (include/vmas_hip.h)."""
        if self.lib.vmas_world_set_specialized(self._h, 1 if on else 0) != 0:
            raise VmasHipError(A.last_error())

    @property
    def specialized(self) -> bool:
        return bool(self.lib.vmas_world_get_specialized(self._h))

    def specialize(self, post: int = 0, cache_dir: Optional[str] = None, cached_only: bool = False) -> bool:
        """Compile (or fetch from the cache) and load a world-specialised kernel for THIS world and batch geometry
        (specialize.py; tens of seconds on a cache miss).  ``post``: the fused epilogue its steps carry (VMAS_POST_*)."""
        from .specialize import specialize

        return specialize(self, post, cache_dir, cached_only=cached_only)

    def set_compact(self, mode: int):
        """-1: the library's choice, 0: never, 1: whenever the world qualifies - the lane-compacted step kernel
        (include/vmas_hip.h, vmas_world_set_compact)."""
        if self.lib.vmas_world_set_compact(self._h, int(mode)) != 0:
            raise VmasHipError(A.last_error())

    def compact_stats(self) -> dict:
        """Test / diagnostics hook (include/vmas_debug_hip.h): what the adaptive choice between the compacted kernel and the
        interpreter has seen.  Synchronises the device."""
        out = (C.c_int64 * 4)()
        self.lib.vmas_debug_compact_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        if self.lib.vmas_debug_compact_stats(self._h, out) != 0:
            raise VmasHipError(A.last_error())
        return {"contacts": int(out[0]), "tiles": int(out[1]), "switches": int(out[2]), "backoff": int(out[3])}

    def lazy_stats(self) -> dict:
        """Test / diagnostics hook (include/vmas_debug_hip.h): the lazy exact broad phase's counters.  Synchronises the device."""
        out = (C.c_int64 * 4)()
        self.lib.vmas_debug_lazy_stats.argtypes = [C.c_void_p, C.POINTER(C.c_int64)]
        if self.lib.vmas_debug_lazy_stats(self._h, out) != 0:
            raise VmasHipError(A.last_error())
        return {"launches": int(out[0]), "tiles_asked": int(out[1]), "found_pair_off": int(out[2]), "repeated_polls": int(out[3])}

    @property
    def compact(self) -> bool:
        """True if plain steps of this world run the lane-compacted kernel (csrc/vmas_compact.h)."""
        return bool(self.lib.vmas_world_get_compact(self._h))

    def reserve_epilogue(self, post_kind: int, n_packages: int = 0):
        """The steps of this world will be one-launch Environment.step calls with this post-step epilogue: let the
        library choose its kernel geometry with the epilogue's LDS included (include/vmas_env_hip.h)."""
        if self.lib.vmas_world_reserve_epilogue(self._h, int(post_kind), int(n_packages)) != 0:
            raise VmasHipError(A.last_error())

    @property
    def lanes_per_env(self) -> int:
        return self.lib.vmas_world_get_lanes_per_env(self._h)

    def step_bytes_per_env(self) -> int:
        return int(self.lib.vmas_world_step_bytes_per_env(self._h))

    # ------------------------------------------------------------------ snapshots
    def get_state(self):
        """(state [E, 6, batch], agent_ft [A, 3, batch]) - copies of the packed buffers without their padding columns."""
        return self.state[:, :, : self.batch].clone(), self.agent_ft[:, :, : self.batch].clone()

    def set_state(self, state: torch.Tensor, agent_ft: Optional[torch.Tensor] = None) -> None:
        """Write a snapshot taken with ``get_state`` (same shapes) back into the packed buffers."""
        assert tuple(state.shape) == (self.state.shape[0], A.STATE_FIELDS, self.batch), tuple(state.shape)
        self.state[:, :, : self.batch].copy_(state)
        if agent_ft is not None:
            assert tuple(agent_ft.shape) == (self.agent_ft.shape[0], A.AGENT_FIELDS, self.batch), tuple(agent_ft.shape)
            self.agent_ft[:, :, : self.batch].copy_(agent_ft)

    # ------------------------------------------------------------------ views
    def pos(self, e: int) -> torch.Tensor:  # [B, 2] view
        return self.state[e, 0:2, : self.batch].T

    def vel(self, e: int) -> torch.Tensor:
        return self.state[e, 2:4, : self.batch].T

    def rot(self, e: int) -> torch.Tensor:  # [B, 1] view
        return self.state[e, 4:5, : self.batch].T

    def ang_vel(self, e: int) -> torch.Tensor:
        return self.state[e, 5:6, : self.batch].T

    def force(self, a: int) -> torch.Tensor:
        return self.agent_ft[a, 0:2, : self.batch].T

    def torque(self, a: int) -> torch.Tensor:
        return self.agent_ft[a, 2:3, : self.batch].T

    # ------------------------------------------------------------------ compute
    def _stream(self, stream) -> C.c_void_p:
        s = stream if stream is not None else torch.cuda.current_stream(self.device_index)
        return C.c_void_p(s.cuda_stream)

    @staticmethod
    def _dptr(t: Optional[torch.Tensor]):
        return None if t is None else C.c_void_p(t.data_ptr())

    def step(
        self,
        pair_mask: Optional[torch.Tensor] = None,
        joint_fixed_rot: Optional[torch.Tensor] = None,
        entity_gravity: Optional[torch.Tensor] = None,
        first_substep: int = 0,
        n_substeps: int = 0,
        stream=None,
        exact: bool = False,
    ) -> None:
        """World.step() for the whole batch, asynchronous on the (current) stream.  ``exact``: the reference's
        batch-global broad phase re-evaluated at every substep (include/vmas_hip.h, VmasStepArgs.exact_broad_phase)."""
        args = A.StepArgs()
        args.exact_broad_phase = 1 if exact else 0
        args.pair_mask = self._dptr(pair_mask)
        if joint_fixed_rot is not None:
            assert joint_fixed_rot.shape == (len(self.spec.joints), self.ld) and joint_fixed_rot.is_contiguous()
        if entity_gravity is not None:
            assert entity_gravity.shape == (self.spec.n_entities, 2, self.ld) and entity_gravity.is_contiguous()
        args.joint_fixed_rot = self._dptr(joint_fixed_rot)
        args.entity_gravity = self._dptr(entity_gravity)
        args.first_substep, args.n_substeps = first_substep, n_substeps
        rc = self.lib.vmas_world_step(
            self._h, self._dptr(self.state), self._dptr(self.agent_ft), self.ld, C.byref(args), self._stream(stream)
        )
        if rc != 0:
            raise VmasHipError(A.last_error())

    def make_stepper(self, exact: bool = False):
        """A zero-argument callable that enqueues ``World.step()`` without per-call inputs on the current stream: every
        argument that does not change between steps (handle, buffer pointers, the optional-arguments struct) is marshalled
        once, a call is one foreign call (what ``adapter.attach`` rebinds ``world.step`` to: ~1.5 us of host time per
        step instead of ~6 through ``step()``'s checks and struct building)."""
        fn = self.lib.vmas_world_step
        h, st, ft, ld = self._h, C.c_void_p(self.state.data_ptr()), C.c_void_p(self.agent_ft.data_ptr()), self.ld
        args = None
        if exact:
            self._stepper_args = sa = A.StepArgs()  # (kept alive by the world: the closure holds a pointer into it)
            sa.exact_broad_phase = 1
            args = C.byref(sa)
        dev = self.device_index
        raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)

        def step():
            stream = raw(dev) if raw is not None else torch.cuda.current_stream(dev).cuda_stream
            if fn(h, st, ft, ld, args, stream) != 0:
                raise VmasHipError(A.last_error())

        return step

    def step_env(self, ingest_args, err_flags: Optional[torch.Tensor], post_kind: int, post_desc, post_buffers,
                 joint_fixed_rot: Optional[torch.Tensor] = None, entity_gravity: Optional[torch.Tensor] = None,
                 stream=None, exact: bool = False) -> None:
        """World.step() with the action ingest as prologue and a scenario post-step as epilogue, ONE
        launch (``vmas_world_step_env``, include/vmas_env_hip.h)."""
        args = None
        if joint_fixed_rot is not None or entity_gravity is not None or exact:
            sa = A.StepArgs()
            sa.joint_fixed_rot = self._dptr(joint_fixed_rot)
            sa.entity_gravity = self._dptr(entity_gravity)
            sa.exact_broad_phase = 1 if exact else 0
            args = C.byref(sa)
        rc = self.lib.vmas_world_step_env(
            self._h, self._dptr(self.state), self._dptr(self.agent_ft), self.ld, args,
            C.byref(ingest_args) if ingest_args is not None else None, self._dptr(err_flags), int(post_kind),
            C.cast(C.pointer(post_desc), C.c_void_p) if post_desc is not None else None,
            C.cast(C.pointer(post_buffers), C.c_void_p) if post_buffers is not None else None, self._stream(stream),
        )
        if rc != 0:
            raise VmasHipError(A.last_error())

    def _exact_args(self, exact: bool):
        if not exact:
            return None
        ref = getattr(self, "_exact_step_ref", None)
        if ref is None:  # (ONE struct for the world's lifetime: it never changes, and making one per call is microseconds of host time)
            self._exact_step_args = sa = A.StepArgs()
            sa.exact_broad_phase = 1
            ref = self._exact_step_ref = C.byref(sa)
        return ref

    def step_n(self, n_steps: int, forces: Optional[torch.Tensor] = None, stream=None, exact: bool = False) -> None:
        """``n_steps`` World.step() launches enqueued from C.  ``forces`` [n_steps, A, 3, ld]
        (packed like ``agent_ft``) supplies per-step agent forces; None re-uses ``agent_ft``.  ``exact``: the reference's
        batch-global broad phase (one queue; include/vmas_hip.h)."""
        if forces is not None:
            assert forces.shape == (n_steps,) + tuple(self.agent_ft.shape) and forces.is_contiguous()
            assert forces.device == self.agent_ft.device and forces.dtype == torch.float32
            ft, stride = forces, self.agent_ft.numel()
        else:
            ft, stride = self.agent_ft, 0
        rc = self.lib.vmas_world_step_n(
            self._h, self._dptr(self.state), self._dptr(ft), self.ld, stride, int(n_steps), self._exact_args(exact),
            self._stream(stream)
        )
        if rc != 0:
            raise VmasHipError(A.last_error())

    def rollout(self, n_steps: int, forces: Optional[torch.Tensor] = None, stream=None, exact: bool = False) -> None:
        """The same steps as ``step_n`` in ONE persistent launch (state stays in LDS)."""
        if forces is not None:
            assert forces.shape == (n_steps,) + tuple(self.agent_ft.shape) and forces.is_contiguous()
            assert forces.device == self.agent_ft.device and forces.dtype == torch.float32
            ft, stride = forces, self.agent_ft.numel()
        else:
            ft, stride = self.agent_ft, 0
        rc = self.lib.vmas_world_rollout(
            self._h, self._dptr(self.state), self._dptr(ft), self.ld, stride, int(n_steps), self._exact_args(exact),
            self._stream(stream)
        )
        if rc != 0:
            raise VmasHipError(A.last_error())

    def pair_mask(self, stream=None) -> torch.Tensor:
        """Batch-global broad phase of World.collides on the current state."""
        rc = self.lib.vmas_world_pair_mask(
            self._h, self._dptr(self.state), self.ld, self._dptr(self._mask), self._stream(stream)
        )
        if rc != 0:
            raise VmasHipError(A.last_error())
        return self._mask

    def step_exact(self, joint_fixed_rot=None, entity_gravity=None, stream=None) -> None:
        """World.step() with the reference's batch-global broad phase (core.py:2797-2801) re-evaluated at every
        substep: ONE launch for batches of at most 64 x CUs environments (mask + grid barrier inside the step kernel),
        a mask launch + a substep launch per substep beyond (the library decides)."""
        self.step(None, joint_fixed_rot, entity_gravity, 0, 0, stream, exact=True)

    def step_exact_launches(self, joint_fixed_rot=None, entity_gravity=None, stream=None) -> None:
        """The same step as explicit launches (``vmas_world_pair_mask`` + a one-substep ``vmas_world_step`` with that
        mask, per substep): what ``step_exact`` must equal bit for bit."""
        for s in range(self.spec.substeps):
            m = self.pair_mask(stream)
            self.step(m, joint_fixed_rot, entity_gravity, s, 1, stream)

    def exact_form(self) -> int:
        """How ``exact=True`` steps of this world run outside graph capture (include/vmas_hip.h, vmas_world_exact_form):
        0 nothing to do (sphere-sphere pairs only), 1 the lazy form inside the step launch (any batch size; fused epilogues
        and gated launches allowed), 2 the grid-barrier form inside the launch, 3 a mask launch + a launch per substep."""
        rc = int(self.lib.vmas_world_exact_form(self._h))
        if rc < 0:
            raise VmasHipError(A.last_error())
        return rc

    def exact_status(self) -> int:
        """0 = every in-kernel grid barrier of the exact steps so far completed (synchronises the device)."""
        rc = int(self.lib.vmas_world_exact_status(self._h))
        if rc < 0:
            raise VmasHipError(A.last_error())
        return rc

    def set_queries(self, queries) -> None:
        """Register (kind, a, b) geometric queries: kind "distance" | "overlap", entity indices."""
        arr = (A.Query * max(len(queries), 1))()
        for i, (kind, a, b) in enumerate(queries):
            arr[i].kind = A.QUERY_OVERLAP if kind == "overlap" else A.QUERY_DISTANCE
            arr[i].a, arr[i].b = int(a), int(b)
        if self.lib.vmas_world_set_queries(self._h, arr, len(queries)) != 0:
            raise VmasHipError(A.last_error())
        self._query_out = torch.zeros(max(len(queries), 1), self.ld, device=self.state.device, dtype=torch.float32)

    def run_queries(self, stream=None) -> torch.Tensor:
        """All registered queries in one launch: [n_queries, ld] (overlap as 1.0 / 0.0)."""
        rc = self.lib.vmas_world_run_queries(
            self._h, self._dptr(self.state), self.ld, self._dptr(self._query_out), self._stream(stream)
        )
        if rc != 0:
            raise VmasHipError(A.last_error())
        return self._query_out

    def set_lidar_compact(self, mode: int):
        """-1: the library's choice, 0: the plain kernel, 1: whenever the sensor set qualifies - the lane-compacted
        World.cast_rays of sphere-only sensor sets (include/vmas_hip.h, vmas_world_set_lidar_compact)."""
        if self.lib.vmas_world_set_lidar_compact(self._h, int(mode)) != 0:
            raise VmasHipError(A.last_error())

    @property
    def lidar_compact(self) -> bool:
        return bool(self.lib.vmas_world_get_lidar_compact(self._h))

    def cast_rays(self, stream=None) -> torch.Tensor:
        """World.cast_rays for every registered Lidar: [n_lidars, max_rays, ld]."""
        if self._lidar_out is None:
            raise VmasHipError("this world has no Lidar sensors")
        rc = self.lib.vmas_world_cast_rays(
            self._h, self._dptr(self.state), self.ld, self._dptr(self._lidar_out), self._stream(stream)
        )
        if rc != 0:
            raise VmasHipError(A.last_error())
        return self._lidar_out
