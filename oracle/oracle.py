"""ctypes front-end of the CPU oracle (oracle/vmas_oracle.c).

TEST INFRASTRUCTURE ONLY - imported by tests/, __graft_entry__.smoke() and the
cpu_baseline leg of bench.py as the checker of the HIP path.  Nothing under
vectorizedmultiagentsimulator_amd/ imports this module.

State arrays use the packed layout of include/vmas_hip.h:
``state[E, 6, ld]`` and ``agent_ft[A, 3, ld]`` float32, C-contiguous.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from typing import Optional

import numpy as np

from vectorizedmultiagentsimulator_amd import _abi as A
from vectorizedmultiagentsimulator_amd.spec import WorldSpec

_DIR = os.path.dirname(os.path.abspath(__file__))
SRC = os.path.join(_DIR, "vmas_oracle.c")
# VMAS_ORACLE_LIB: another build of the same source (tests/test_oracle_sanitizers.py runs an ASan + UBSan build)
LIB = os.environ.get("VMAS_ORACLE_LIB") or os.path.join(_DIR, "libvmas_oracle.so")
_lib = None


def build(force: bool = False) -> str:
    """gcc the restatement (no FMA contraction; OpenMP only parallelises over envs)."""
    hdr = os.path.join(_DIR, "..", "include", "vmas_hip.h")
    if os.environ.get("VMAS_ORACLE_LIB"):
        return LIB  # (built by whoever set it)
    if (
        not force
        and os.path.exists(LIB)
        and os.path.getmtime(LIB) >= max(os.path.getmtime(SRC), os.path.getmtime(hdr))
    ):
        return LIB
    cmd = ["gcc", "-O2", "-ffp-contract=off", "-fopenmp", "-fPIC", "-shared", "-o", LIB, SRC, "-lm"]
    subprocess.check_call(cmd)
    return LIB


def _load():
    global _lib
    if _lib is None:
        build()
        lib = C.CDLL(LIB)
        vp, i32, i64 = C.c_void_p, C.c_int32, C.c_int64
        lib.vmas_oracle_step.argtypes = [C.POINTER(A.WorldDesc), i32, vp, vp, i64, C.POINTER(A.StepArgs), i32]
        lib.vmas_oracle_step.restype = C.c_int
        lib.vmas_oracle_pair_mask.argtypes = [C.POINTER(A.WorldDesc), i32, vp, i64, vp]
        lib.vmas_oracle_pair_mask.restype = C.c_int
        lib.vmas_oracle_cast_rays.argtypes = [C.POINTER(A.WorldDesc), i32, vp, i64, C.POINTER(A.LidarDesc), i32, vp, i32]
        lib.vmas_oracle_cast_rays.restype = C.c_int
        lib.vmas_oracle_pair_forces.argtypes = [C.POINTER(A.WorldDesc), i32, vp, i64, i32, vp]
        lib.vmas_oracle_pair_forces.restype = C.c_int
        lib.vmas_oracle_queries.argtypes = [C.POINTER(A.WorldDesc), i32, vp, i64, C.POINTER(A.Query), i32, vp]
        lib.vmas_oracle_queries.restype = C.c_int
        lib.vmas_oracle_set_jitter.argtypes = [C.c_uint32]
        lib.vmas_oracle_set_jitter.restype = None
        _lib = lib
    return _lib


def set_jitter(seed: int) -> None:
    """0 = exact libm; otherwise sin/cos/exp/log1p results move by -1/0/+1 ulp
    pseudo-randomly (tolerance calibration, see vmas_oracle.c)."""
    _load().vmas_oracle_set_jitter(seed)


def _ptr(a: Optional[np.ndarray]):
    if a is None:
        return None
    assert a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


class Oracle:
    """Scalar CPU twin of ``HipWorld`` for one WorldSpec."""

    def __init__(self, spec: WorldSpec):
        self.spec = spec
        self.cdesc = spec.to_ctypes()
        self.lib = _load()

    def mask_words(self) -> int:
        return (len(self.spec.pairs) + 31) // 32

    def step(
        self,
        state: np.ndarray,
        agent_ft: np.ndarray,
        batch: Optional[int] = None,
        pair_mask: Optional[np.ndarray] = None,
        joint_fixed_rot: Optional[np.ndarray] = None,
        entity_gravity: Optional[np.ndarray] = None,
        first_substep: int = 0,
        n_substeps: int = 0,
        threads: int = 1,
    ) -> None:
        """In-place World.step on ``state`` / ``agent_ft`` (float32, [.., .., ld])."""
        assert state.dtype == np.float32 and agent_ft.dtype == np.float32
        ld = state.shape[-1]
        batch = ld if batch is None else batch
        assert state.shape == (self.spec.n_entities, A.STATE_FIELDS, ld)
        assert agent_ft.shape == (max(self.spec.n_agents, 0), A.AGENT_FIELDS, ld) or self.spec.n_agents == 0
        args = A.StepArgs()
        if pair_mask is not None:
            assert pair_mask.dtype == np.uint32 and pair_mask.size >= self.mask_words()
        args.pair_mask = _ptr(pair_mask)
        args.joint_fixed_rot = _ptr(joint_fixed_rot)
        args.entity_gravity = _ptr(entity_gravity)
        args.first_substep, args.n_substeps = first_substep, n_substeps
        rc = self.lib.vmas_oracle_step(
            C.byref(self.cdesc.world), batch, _ptr(state), _ptr(agent_ft), ld, C.byref(args), threads
        )
        assert rc == 0, "vmas_oracle_step failed"

    def pair_mask(self, state: np.ndarray, batch: Optional[int] = None) -> np.ndarray:
        ld = state.shape[-1]
        batch = ld if batch is None else batch
        mask = np.zeros(max(self.mask_words(), 1), dtype=np.uint32)
        rc = self.lib.vmas_oracle_pair_mask(C.byref(self.cdesc.world), batch, _ptr(state), ld, _ptr(mask))
        assert rc == 0
        return mask

    def step_exact(self, state, agent_ft, batch=None, joint_fixed_rot=None, entity_gravity=None, threads=1):
        """World.step with the reference's batch-global broad phase re-evaluated at
        every substep (core.py:2797-2801): mask + one substep at a time."""
        for s in range(self.spec.substeps):
            m = self.pair_mask(state, batch)
            self.step(state, agent_ft, batch, m, joint_fixed_rot, entity_gravity, s, 1, threads)

    def cast_rays(self, state: np.ndarray, batch: Optional[int] = None, threads: int = 1) -> np.ndarray:
        ld = state.shape[-1]
        batch = ld if batch is None else batch
        n = len(self.spec.lidars)
        out = np.zeros((n, max(self.cdesc.max_rays, 1), ld), dtype=np.float32)
        rc = self.lib.vmas_oracle_cast_rays(
            C.byref(self.cdesc.world), batch, _ptr(state), ld, self.cdesc.lidars, n, _ptr(out), threads
        )
        assert rc == 0
        return out

    def pair_forces(self, state: np.ndarray, p: int, batch: Optional[int] = None) -> np.ndarray:
        """[6, ld]: fa.x fa.y ta fb.x fb.y tb of static pair ``p`` (ungated)."""
        ld = state.shape[-1]
        batch = ld if batch is None else batch
        out = np.zeros((6, ld), np.float32)
        rc = self.lib.vmas_oracle_pair_forces(C.byref(self.cdesc.world), batch, _ptr(state), ld, p, _ptr(out))
        assert rc == 0
        return out

    def queries(self, state: np.ndarray, queries, batch: Optional[int] = None) -> np.ndarray:
        """[(kind, a, b)] -> [n, ld] distances / overlaps (1.0 / 0.0)."""
        ld = state.shape[-1]
        batch = ld if batch is None else batch
        arr = (A.Query * max(len(queries), 1))()
        for i, (kind, a, b) in enumerate(queries):
            arr[i].kind = A.QUERY_OVERLAP if kind == "overlap" else A.QUERY_DISTANCE
            arr[i].a, arr[i].b = int(a), int(b)
        out = np.zeros((max(len(queries), 1), ld), np.float32)
        rc = self.lib.vmas_oracle_queries(C.byref(self.cdesc.world), batch, _ptr(state), ld, arr, len(queries), _ptr(out))
        assert rc == 0
        return out
