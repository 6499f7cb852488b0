"""Host side of ``include/vmas_env_hip.h``: the stages of ``Environment.step`` on either side of
``World.step()`` as one kernel launch each.

* ``ActionIngest``   - ``Environment._set_action`` (environment.py:616-749, continuous branch)
  followed by ``Holonomic(.WithRotation).process_action`` for every policy agent.
* ``BalancePost`` / ``TransportPost`` / ``NavigationPost`` - the scenario's
  reward / observation / done / info (balance.py:218-267, transport.py:131-191,
  navigation.py:200-285) over the packed state, writing the reference's own output shapes.

Each class owns the descriptor and the scenario's persistent tensors (shaping terms, flags);
outputs are fresh tensors per call unless ``static_outputs`` (HIP-graph replay) is set.  There is
no torch fallback in here: the scenario classes keep their tensor-op methods for user code that
calls ``reward()`` / ``observation()`` directly, the environment uses the kernels.
"""
from __future__ import annotations

import ctypes as C
import math
import sys
from typing import List, Optional

import torch
from torch import Tensor

from . import _abi as A
from .backend import VmasHipError
from .core import Holonomic, HolonomicWithRotation


try:  # the current stream's handle without building a torch.cuda.Stream object per step (1.5 us -> 0.2 us)
    _raw_stream = torch._C._cuda_getCurrentRawStream
except AttributeError:  # pragma: no cover
    _raw_stream = None


def _stream(device) -> int:
    if _raw_stream is not None and device.index is not None:
        return _raw_stream(device.index)
    return torch.cuda.current_stream(device).cuda_stream


def _check(rc: int):
    if rc != 0:
        raise VmasHipError(A.last_error())


def _per_dim(v, n) -> List[float]:
    return [float(x) for x in v] if isinstance(v, (list, tuple)) else [float(v)] * n


class ActionIngest:
    """One launch for all agents' ``_set_action`` + ``process_action``."""

    @staticmethod
    def supports(env) -> Optional[str]:
        """None if the environment's action path can be fused, else the reason it cannot."""
        if len(env.agents) > A.ENV_MAX_AGENTS:
            return "too many agents"
        from .scenario import BaseScenario
        sc = env.scenario
        known = {id(s["agent"]) for s in sc.fused_agent_scripts()} if hasattr(sc, "fused_agent_scripts") else set()
        if any(id(a) not in known for a in env.world.scripted_agents) or len(known) > A.ENV_MAX_SCRIPTS:
            return "scripted agents"
        if type(sc).process_action is not BaseScenario.process_action and not hasattr(sc, "fused_action_factors"):
            return "scenario overrides process_action"
        for a in env.agents:
            if type(a.dynamics) not in (Holonomic, HolonomicWithRotation):
                return f"dynamics {type(a.dynamics).__name__}"
            if a.action_size != a.dynamics.needed_action_size:
                return "action_size != dynamics.needed_action_size"  # (the kernel maps component k to force/torque row k)
            if a.action.u_noise not in (0, 0.0, None):
                return "action noise"
            if not a.silent and env.world.dim_c > 0:
                return "communication actions"
        return None

    def __init__(self, env):
        self.env = env
        self.lib = A.load_library()
        w = env.world
        B = env.num_envs
        self.args = A.IngestArgs()
        self.args.n_agents = len(env.agents)
        self.args.clamp = 1 if env.clamp_action else 0
        self.u = []
        for i, a in enumerate(env.agents):
            n = a.action_size
            s = self.args.agents[i]
            s.action_size = n
            s.agent_index = a._agent_index
            for k, v in enumerate(_per_dim(a.action.u_range, n)):
                s.u_range[k] = v
            # what scenario.process_action does to the scaled action, when it is a per-dimension factor
            extra = env.scenario.fused_action_factors(a) if hasattr(env.scenario, "fused_action_factors") else None
            for k, v in enumerate(_per_dim(a.action.u_multiplier, n)):
                s.u_multiplier[k] = v * (extra[k] if extra is not None else 1.0)
            if not env.continuous_actions:
                for k, nv in enumerate(a.discrete_action_nvec):
                    s.nvec[k] = int(nv)
            u = torch.zeros(B, n, device=env.device, dtype=torch.float32)
            s.u_out = u.data_ptr()
            a.action.u = u  # agent.action.u stays readable by scenario code
            self.u.append(u)
        scripts = env.scenario.fused_agent_scripts() if hasattr(env.scenario, "fused_agent_scripts") else []
        self.args.n_scripts = len(scripts)
        for i, sp in enumerate(scripts):  # scripted agents whose script runs on the device
            s = self.args.scripts[i]
            agent = sp["agent"]
            s.kind, s.agent_index, s.entity = sp["kind"], agent._agent_index, agent._index
            for k, v in enumerate(sp["params"]):
                s.params[k] = v
            u = torch.zeros(B, 2, device=env.device, dtype=torch.float32)
            s.u_out = u.data_ptr()
            agent.action.u = u
            self.u.append(u)
        # the action-error flags: ONE word in pinned host memory the kernels OR into (include/vmas_env_hip.h,
        # vmas_host_word_create): read here without a device-to-host copy
        dev = torch.device(env.device)
        h, d = C.c_void_p(), C.c_void_p()
        _check(self.lib.vmas_host_word_create(dev.index if dev.index is not None else torch.cuda.current_device(), C.byref(h), C.byref(d)))
        self._err_host_ptr, self.err_ptr = h.value, d.value
        self._err_word = C.c_uint32.from_address(h.value)
        self.gate_ptr = self.lib.vmas_host_word_gate(h.value)  # device memory: what a gated step launch reads (StepLauncher.gated)
        self._ft = w._packed_agent_ft()
        self._keep = None
        self._fast = None
        self._slots = [self.args.agents[i] for i in range(len(env.agents))]  # (ctypes array indexing builds a wrapper per access)

    def prepare(self, actions: List[Tensor]):
        """Check the action tensors and point the slots at them (no launch)."""
        env = self.env
        held = []
        fast = self._fast
        if fast is None:  # per agent: (action columns, dtype) - what a well-formed action tensor looks like
            dtype = torch.float32 if env.continuous_actions else torch.int64
            fast = self._fast = [((env.num_envs, env.get_agent_action_size(a)), dtype, (env.get_agent_action_size(a), 1))
                                 for a in env.agents]
        B, dev, cont = env.num_envs, env.device, env.continuous_actions
        slots, prev = self._slots, self._keep
        if prev is not None and len(prev) == len(actions):
            same = True
            for a, b, f in zip(actions, prev, fast):
                # the very tensor objects of the last call (a policy writing into its own buffers) - and still the layout they
                # were checked with: `.data = ...`, resize_() or set_() re-point / re-shape a tensor without changing its identity
                if a is not b or a.shape != f[0] or a.dtype is not f[1] or a.stride() != f[2]:
                    same = False
                    break
            if same:
                for i, act in enumerate(actions):  # only the pointer is read again
                    if cont:
                        slots[i].action = act.data_ptr()
                    else:
                        slots[i].action_index = act.data_ptr()
                return
        for i, (agent, act) in enumerate(zip(env.agents, actions)):
            shape, dtype, strides = fast[i]
            if (type(act) is Tensor and act.dtype is dtype and act.shape == shape and act.stride() == strides
                    and act.device == dev):  # the usual case: nothing to convert ([B, size] contiguous, on the env's device)
                held.append(act)
                if cont:
                    slots[i].action = act.data_ptr()
                else:
                    slots[i].action_index = act.data_ptr()
                continue
            if not isinstance(act, Tensor):
                act = torch.tensor(act, dtype=torch.float32, device=env.device)
            if act.dim() == 1:
                act = act.unsqueeze(-1)
            assert act.shape[0] == env.num_envs, (
                f"Actions used in input of env must be of len {env.num_envs}, got {act.shape[0]}")
            want = env.get_agent_action_size(agent)
            assert act.shape[1] == want, f"Agent {agent.name} has wrong action size, got {act.shape[1]}, expected {want}"
            dtype = torch.float32 if env.continuous_actions else torch.int64
            if act.dtype != dtype or act.device != env.device or not act.is_contiguous():
                act = act.detach().to(device=env.device, dtype=dtype).contiguous()
            held.append(act)
            if env.continuous_actions:
                self.args.agents[i].action = act.data_ptr()
            else:
                self.args.agents[i].action_index = act.data_ptr()
        self._keep = held  # the launch is asynchronous: keep the inputs alive until the next call

    def prepare_rollout(self, actions: List[Tensor], n_steps: int, validate: bool):
        """Point the slots at K-step action tensors ([K, B, action_size] per agent, continuous; [K, B] or [K, B, 1]
        flat indices, discrete).  ``validate``: the reference's asserts (environment.py:621,651-653) on the whole
        tensors up front - one host sync - because a multi-step launch cannot stop at a bad action."""
        env = self.env
        held, bad_nan, bad_range = [], None, None
        for i, (agent, act) in enumerate(zip(env.agents, actions)):
            want = env.get_agent_action_size(agent)
            if act.dim() == 2:
                act = act.unsqueeze(-1)
            assert act.shape == (n_steps, env.num_envs, want), (
                f"Agent {agent.name}: rollout actions must be [{n_steps}, {env.num_envs}, {want}], got {tuple(act.shape)}")
            dtype = torch.float32 if env.continuous_actions else torch.int64
            if act.dtype != dtype or act.device != env.device or not act.is_contiguous():
                act = act.detach().to(device=env.device, dtype=dtype).contiguous()
            held.append(act)
            if env.continuous_actions:
                self.args.agents[i].action = act.data_ptr()
                if validate:
                    rng = torch.tensor(_per_dim(agent.action.u_range, want), device=env.device)
                    n = (act != act).any()
                    r = (act.abs() > rng).any() if not env.clamp_action else torch.zeros((), dtype=torch.bool, device=env.device)
                    bad_nan = n if bad_nan is None else bad_nan | n
                    bad_range = r if bad_range is None else bad_range | r
            else:
                self.args.agents[i].action_index = act.data_ptr()
                if validate:  # step() raises on an index outside [0, prod(nvec)) (environment.py:657-661): so does this
                    r = ((act < 0) | (act >= math.prod(agent.discrete_action_nvec))).any()
                    bad_range = r if bad_range is None else bad_range | r
                    bad_nan = torch.zeros((), dtype=torch.bool, device=env.device) if bad_nan is None else bad_nan
        if validate and bad_nan is not None:
            flags = torch.stack([bad_nan, bad_range]).cpu()
            assert not bool(flags[0]), "actions contain NaN"
            assert not bool(flags[1]), ("Physical actions of an agent are out of its range" if env.continuous_actions
                                        else "Discrete action of an agent is out of range")
        self._keep = held

    def __del__(self):
        p, self._err_host_ptr = getattr(self, "_err_host_ptr", None), None
        if p:
            try:  # (hipHostFree is illegal while a stream is being captured: deferred like a world's release)
                from .backend import release_later
                lib = self.lib
                release_later(lambda: lib.vmas_host_word_destroy(C.c_void_p(p)))
            except Exception:  # noqa: BLE001 (interpreter shutdown)
                pass

    @staticmethod
    def _raise(flags: int):
        """The reference's asserts (environment.py:621,651-653) for the flags a kernel raised."""
        assert not (flags & A.ACTION_ERR_NAN), "actions contain NaN"
        raise AssertionError("Physical actions of an agent are out of its range")

    def pending(self):
        """Deferred validation: flags a PREVIOUS launch raised (``err_ptr`` passed to the step launch itself), read from the
        host word without any synchronisation - a bad action of step t raises at step t + 1 at the latest once step t's
        kernel has run, for the price of one memory read."""
        flags = self._err_word.value
        if flags:
            self._err_word.value = 0
            self._raise(flags)

    def launch(self, validate: bool):
        """The stand-alone ingest kernel on the prepared action tensors; ``validate``: + ONE stream synchronisation and the
        reference's asserts on what it found (vmas_env_validate_actions)."""
        env = self.env
        ft = env.world._packed_agent_ft()
        if validate:
            flags = self.lib.vmas_env_validate_actions(C.byref(self.args), env.num_envs, env.world._packed_state().data_ptr(),
                                                       ft.data_ptr(), ft.shape[-1], self._err_host_ptr, self.err_ptr, _stream(env.device))
            if flags < 0:
                raise VmasHipError(A.last_error())
            if flags:
                self._raise(flags)
            return
        _check(self.lib.vmas_env_ingest_actions(C.byref(self.args), env.num_envs, env.world._packed_state().data_ptr(),
                                                ft.data_ptr(), ft.shape[-1], None, _stream(env.device)))

    def validate_begin(self) -> int:
        """First half of ``validate()`` - enqueue the check, do not wait: the caller launches the step GATED on its result
        (``StepLauncher.gated``) and then calls ``validate_end``; the wait overlaps the step's own execution."""
        env = self.env
        ft = env.world._packed_agent_ft()
        seq = self.lib.vmas_env_validate_begin(C.byref(self.args), env.num_envs, env.world._packed_state().data_ptr(), ft.data_ptr(),
                                               ft.shape[-1], self._err_host_ptr, self.err_ptr, _stream(env.device))
        if seq < 0:
            raise VmasHipError(A.last_error())
        return seq

    def validate_end(self, seq: int) -> int:
        """The flags of the validation ``seq`` (0: the gated step behind it ran; nonzero: it did nothing - the caller raises)."""
        flags = self.lib.vmas_env_validate_end(self._err_host_ptr, seq, _stream(self.env.device))
        if flags < 0:
            raise VmasHipError(A.last_error())
        return flags

    def validate(self):
        """``validate_actions`` on a path whose ingest is the physics kernel's prologue: the reference asserts BEFORE it
        touches the world (environment.py:621,651-653), so the actions are checked by the small stand-alone kernel (it
        writes only ``agent.action.u`` and the agent-force rows, which the reference has also set by then) and the step is
        launched only if that passed - a bad action never reaches the integrator."""
        self.launch(True)

    def __call__(self, actions: List[Tensor], validate: bool):
        self.prepare(actions)
        self.launch(validate)


class MaskedReset:
    """``vmas_env_reset_where``: Environment.reset_at for every masked environment in one launch.

    The scenario describes its ``reset_world_at`` as a spawn program (``scenario.fused_reset_program()``):
    ``ops``   - ("uniform", entity, (x_lo, x_hi), (y_lo, y_hi), min_dist, avoid_from_op[, rot]) |
                ("offset", entity, base_entity, (dx_lo, dx_hi), dy) | ("fixed", entity, x, y[, rot]), in placement order
                (``rot``: the entity's rotation is set as well);
    ``terms`` - (tensor [B], entity_a, entity_b, factor): tensor[env] = |pos(a) - pos(b)| * factor, or
                (tensor [B], None, None, value): tensor[env] = value, or
                (tensor [B], "point", entity, (px, py), factor): |pos(entity) - (px, py)| * factor, or
                (tensor [B], "min", [entities], entity_b, factor): min over the (consecutive) entities of |pos - pos(b)| * factor;
    ``flags`` - bool tensors [B] cleared for a reset environment."""

    def __init__(self, env, program: dict, seed: int):
        self.env, self.lib = env, A.load_library()
        w = env.world
        args = A.ResetArgs()
        ops, terms, flags = program["ops"], program.get("terms", []), program.get("flags", [])
        assert len(ops) <= A.RESET_MAX_OPS and len(terms) <= A.RESET_MAX_TERMS and len(flags) <= 8
        args.n_ops, args.n_terms, args.n_flags = len(ops), len(terms), len(flags)
        for i, op in enumerate(ops):
            o = args.ops[i]
            if op[0] == "uniform":
                _, ent, xb, yb, min_dist, avoid_from = op[:6]
                o.kind, o.entity, o.avoid_from = A.SPAWN_UNIFORM, ent._index, int(avoid_from)
                o.x_lo, o.x_hi, o.y_lo, o.y_hi, o.min_dist = float(xb[0]), float(xb[1]), float(yb[0]), float(yb[1]), float(min_dist)
                if len(op) > 6 and op[6] is not None:
                    o.has_rot, o.rot = 1, float(op[6])
            elif op[0] == "offset":
                _, ent, base, dxb, dy = op
                o.kind, o.entity, o.base = A.SPAWN_OFFSET, ent._index, base._index
                o.x_lo, o.x_hi, o.y_lo = float(dxb[0]), float(dxb[1]), float(dy)
            else:
                _, ent, x, y = op[:4]
                o.kind, o.entity, o.x_lo, o.y_lo = A.SPAWN_FIXED, ent._index, float(x), float(y)
                if len(op) > 4 and op[4] is not None:
                    o.has_rot, o.rot = 1, float(op[4])
        self._terms = terms
        for i, term in enumerate(terms):
            T = args.terms[i]
            if term[1] == "point":
                _, _, ent, (px, py), f = term
                T.kind, T.a, T.b, T.px, T.py, T.factor = A.TERM_DIST_POINT, ent._index, -1, float(px), float(py), float(f)
            elif term[1] == "min":
                _, _, ents, b, f = term
                idx = [e._index for e in ents]
                assert idx == list(range(idx[0], idx[0] + len(idx))), "a 'min' term takes consecutive entities"
                T.kind, T.a, T.n, T.b, T.factor = A.TERM_MIN_DIST, idx[0], len(idx), b._index, float(f)
            else:
                t, a, b, f = term
                T.kind = A.TERM_DIST
                T.a, T.b, T.factor = (a._index, b._index, float(f)) if a is not None else (-1, -1, float(f))
        self._flags = flags
        self.episode = torch.zeros(env.num_envs, dtype=torch.int32, device=env.device)
        args.episode = self.episode.data_ptr()
        #: [1] placements that hit the kernel's try cap and kept an overlapping position (0 for every feasible program)
        self.gave_up = torch.zeros(1, dtype=torch.int32, device=env.device)
        args.gave_up = self.gave_up.data_ptr()
        args.seed = int(seed) & 0xFFFFFFFFFFFFFFFF
        self.args = args
        self.nE, self.nA = len(w.entities), len(w.agents)

    def __call__(self, mask: Tensor):
        env, w, a = self.env, self.env.world, self.args
        for i, (t, *_rest) in enumerate(self._terms):  # (the scenario may have rebound its tensors since the last call)
            t = t() if callable(t) else t
            assert t.shape == (env.num_envs,) and t.dtype == torch.float32 and t.is_contiguous()
            a.terms[i].out = t.data_ptr()
        for i, t in enumerate(self._flags):
            t = t() if callable(t) else t
            assert t.shape == (env.num_envs,) and t.dtype == torch.bool and t.is_contiguous()
            a.flags[i] = t.data_ptr()
        a.steps = env.steps.data_ptr()
        st, ft = w._packed_state(), w._packed_agent_ft()
        _check(self.lib.vmas_env_reset_where(C.byref(a), env.num_envs, self.nE, self.nA, mask.data_ptr(), st.data_ptr(),
                                             ft.data_ptr(), st.shape[-1], _stream(env.device)))


class StepLauncher:
    """``vmas_world_step_env`` with every argument that does not change between steps marshalled once
    (world handle, buffer pointers, the ingest struct): the Python side of a one-launch step is then
    pointer updates + one foreign call.  Re-binds itself when the world rebuilt its backend."""

    def __init__(self, env, ingest: "ActionIngest"):
        self.env, self.ingest = env, ingest
        self.fn = A.load_library().vmas_world_step_env
        self._be = None
        self._cd = self._cb = self._rd = self._rb = None
        self._refs = {}

    def _bind(self):
        w = self.env.world
        be = self._be = w._get_backend()
        self._h = be._h
        self._st, self._ft, self._ld = C.c_void_p(be.state.data_ptr()), C.c_void_p(be.agent_ft.data_ptr()), be.ld
        spec = w.spec
        self._per_env = any(j.per_env_fixed_rotation for j in spec.joints) or any(e.per_env_gravity for e in spec.entities)
        self._exact = bool(w.exact_broad_phase)
        self._exact_sa = None
        self._exact_ref = None
        if self._exact:
            self._exact_sa = A.StepArgs()
            self._exact_sa.exact_broad_phase = 1
            self._exact_ref = C.byref(self._exact_sa)
        self._ing = C.byref(self.ingest.args)
        self._err = C.c_void_p(self.ingest.err_ptr)
        self._gate = C.c_void_p(self.ingest.gate_ptr)
        self._gated_fn = A.load_library().vmas_world_step_env_gated
        self._dev = self.env.device

    def __call__(self, kind: int, desc, buffers, validate: bool):
        w = self.env.world
        if self._be is None or w._backend is not self._be:
            self._bind()
        w._query_cache = None
        args = self._args(w)
        if desc is not self._cd:  # (the structs persist: their references are built once)
            self._cd = desc
            self._rd = C.byref(desc) if desc is not None else None
        if buffers is not self._cb:  # (a few buffer structs alternate - the post-step's output sets: one reference each)
            self._cb = buffers
            if buffers is None:
                self._rb = None
            else:
                hit = self._refs.get(id(buffers))
                if hit is None or hit[0] is not buffers:
                    if len(self._refs) > 64:
                        self._refs.clear()
                    hit = self._refs[id(buffers)] = (buffers, C.byref(buffers))
                self._rb = hit[1]
        rc = self.fn(self._h, self._st, self._ft, self._ld, args, self._ing, self._err if validate else None, kind,
                     self._rd, self._rb, _stream(self._dev))
        if rc != 0:
            raise VmasHipError(A.last_error())


    def _args(self, w):
        """``VmasStepArgs*`` of this step (None: no optional input).  Without per-environment inputs the struct never changes: ONE
        instance made at bind time (round 6: a fresh ctypes struct per call, now that the reference's broad-phase rule makes
        every call carry one, was 2.5 us of host time in front of every launch - balance 32 768: 13.8 -> 11.x us per step)."""
        if not self._per_env:
            return self._exact_ref
        jfr, eg = w._per_env_inputs()
        sa = A.StepArgs()
        sa.joint_fixed_rot = jfr.data_ptr() if jfr is not None else None
        sa.entity_gravity = eg.data_ptr() if eg is not None else None
        sa.exact_broad_phase = 1 if self._exact else 0
        self._sa_keep = sa
        return C.byref(sa)

    def can_gate(self, kind) -> bool:
        """Every kind of step can be launched gated (round 6: each of its kernels reads the gate; ``refused()`` takes back what
        the host advanced for a launch that found it shut) - except with the exact broad phase in its grid-barrier / launch
        per substep forms, whose sequence numbers advance on the host (include/vmas_env_hip.h)."""
        if self._be is None or self.env.world._backend is not self._be:
            self._bind()
        # (the lazy form of the exact broad phase carries no barrier whose count the host advances: it can be gated)
        return not self._exact or self._be.exact_form() <= 1

    def refused(self):
        """The gated launch just made found the gate shut (``validate_end`` returned flags): ``vmas_world_gated_refused``."""
        if A.load_library().vmas_world_gated_refused(self._h) != 0:
            raise VmasHipError(A.last_error())

    def gated(self, kind: int, desc, buffers):
        """``vmas_world_step_env_gated``: the step launch that does nothing if the validation in front of it raised flags."""
        w = self.env.world
        if self._be is None or w._backend is not self._be:
            self._bind()
        w._query_cache = None
        args = self._args(w)
        rc = self._gated_fn(self._h, self._st, self._ft, self._ld, args, self._ing, self._gate, kind,
                            C.byref(desc) if desc is not None else None, C.byref(buffers) if buffers is not None else None,
                            _stream(self._dev))
        if rc != 0:
            raise VmasHipError(A.last_error())

    def rollout(self, kind: int, desc, buffers, n_steps: int):
        """``vmas_world_rollout_env``: n_steps Environment.step() calls in one launch."""
        w = self.env.world
        if self._be is None or w._backend is not self._be:
            self._bind()
        w._query_cache = None
        args = self._args(w)
        rc = A.load_library().vmas_world_rollout_env(
            self._h, self._st, self._ft, self._ld, args, self._ing, None, kind,
            C.byref(desc) if desc is not None else None, C.byref(buffers) if buffers is not None else None, int(n_steps),
            _stream(self._dev))
        if rc != 0:
            raise VmasHipError(A.last_error())


import os as _os

_use_count = getattr(torch._C, "_storage_Use_Count", None)  # references on a storage: every tensor / view over it holds one
_getrefcount = sys.getrefcount
# The pool needs THREE witnesses that the caller holds nothing of a set: the storage's use count (views), the Python
# reference counts of the handed-out tensor objects, and their TensorImpl use counts (a C++-only handle on the same
# TensorImpl - an autograd SavedVariable of ``policy(obs[i])`` kept for a later backward() - need not show in the Python
# count on every torch build).  Without any of them, or with VMAS_AMD_NO_OUTPUT_POOL=1, every step allocates its outputs.
_impl_count = getattr(torch.Tensor, "_use_count", None)
POOLING = (_use_count is not None and _impl_count is not None and not _os.environ.get("VMAS_AMD_NO_OUTPUT_POOL"))


def _rcs(tensors):
    """Python reference counts of the tensor objects (one function for the baseline and for every check: the same transient
    references are counted both times)."""
    return [_getrefcount(t) for t in tensors]


class _OutSet:
    """One step's output tensors, all views of ONE storage, with everything that is the same every time the set is used
    built once: the buffer struct the kernel reads (pointers filled in), the per-agent lists / info dictionaries
    ``Environment.step`` returns (as templates: a step hands out copies of the containers, the tensors themselves), and what
    tells whether the caller still holds any of it - the storage's reference count (a view the caller made keeps the
    storage alive) and the Python reference counts of the handed-out tensor objects.  A set whose counts are back at their
    baselines is invisible to the caller and can be written again: to the caller every step returns fresh tensors, exactly
    as in the reference, without an allocation or a view being made per step (three ``torch.empty`` + three ``unbind`` cost
    the host ~15 us per step around an 11 us kernel)."""

    __slots__ = ("base", "storage", "cdata", "uc0", "tensors", "rc0", "ic0", "buffers", "ref", "v", "obs", "rew", "done", "infos",
                 "extra", "nbytes", "pooled")

    def seal(self):
        """Baselines, taken when nothing but the set itself refers to its tensors (called by the pool, outside the frame
        that built the set)."""
        self.storage = self.base.untyped_storage()
        self.cdata = self.storage._cdata
        self.uc0 = _use_count(self.cdata) if _use_count is not None else -1
        self.rc0 = _rcs(self.tensors)
        self.ic0 = [t._use_count() for t in self.tensors] if _impl_count is not None else None

    def free(self) -> bool:
        if not POOLING or _use_count(self.cdata) != self.uc0:
            return False
        if _rcs(self.tensors) != self.rc0:
            return False
        return [t._use_count() for t in self.tensors] == self.ic0


class _Post:
    """Shared plumbing of the per-scenario post-step kernels."""

    kind = None  # A.POST_*: the post-step can also run as the epilogue of the physics kernel
    POOL_MAX_SETS = 8
    POOL_MAX_BYTES = 4 << 30

    def __init__(self, env):
        self.env = env
        self.lib = A.load_library()
        self.B = env.num_envs
        self.dev = env.device
        self.n = len(env.agents)
        self.static_outputs = False
        self._static_set = None
        self._pool: List[_OutSet] = []

    def persistent_tensors(self) -> List[Tensor]:
        """Scenario tensors the kernel reads AND writes (shaping terms ...)."""
        raise NotImplementedError

    #: (object getter, attribute) pairs ``prepare()`` re-points at the step's output set: saved / restored around a gated
    #: step that a validation refuses (the reference raises before its scenario sees anything of the step)
    def bound_attributes(self):
        return []

    def save_bound(self):
        return [(o, n, o.__dict__.get(n, None) if hasattr(o, "__dict__") else getattr(o, n, None)) for o, n in self.bound_attributes()]

    @staticmethod
    def restore_bound(saved):
        for o, n, v in saved:
            if v is not None:
                setattr(o, n, v)

    def _limit(self) -> A.StepLimit:
        lim = A.StepLimit()
        lim.steps = self.env.steps.data_ptr()
        lim.max_steps = float(self.env.max_steps) if self.env.max_steps is not None else -1.0
        return lim

    # ---- output sets ----------------------------------------------------------------------------------------------
    def _carve(self, fields):
        """``fields``: (name, shape, dtype) -> (the one buffer, {name: view}); every view starts 256-byte aligned."""
        offs, total = [], 0
        for _, shape, dtype in fields:
            n = math.prod(shape) * torch.empty((), dtype=dtype).element_size()
            offs.append((total, n))
            total += (n + 255) // 256 * 256
        base = torch.empty(max(total, 256), dtype=torch.uint8, device=self.dev)
        v = {}
        for (name, shape, dtype), (off, n) in zip(fields, offs):
            v[name] = base[off:off + n].view(dtype).view(shape)
        return base, v

    def _new_set(self) -> _OutSet:
        """Build one output set (tensors, buffer struct with its output pointers, result templates)."""
        raise NotImplementedError

    def _make(self, pooled: bool) -> _OutSet:
        st = self._new_set()
        st.pooled = pooled
        st.nbytes = st.base.numel()
        st.ref = C.byref(st.buffers)
        st.seal()
        return st

    def _acquire(self, dedicated: bool = False) -> _OutSet:
        """The set this step writes.  ``dedicated``: a fresh one that never enters the pool (step_bound's static outputs);
        ``static_outputs`` (HIP-graph replay): always the same one."""
        if dedicated:
            return self._make(False)
        if self.static_outputs:
            if self._static_set is None:
                self._static_set = self._make(False)
            return self._static_set
        pool = self._pool
        for st in pool:  # (always from the front: a caller that keeps one step's results alternates between the first two
            if st.free():  #  sets however large the pool once grew - the working set stays small and warm in the caches)
                return st
        st = self._make(True)
        n = len(pool)
        if n < self.POOL_MAX_SETS and (n + 1) * st.nbytes <= self.POOL_MAX_BYTES and POOLING:
            pool.append(st)
        return st  # (beyond the pool's size the set is simply the caller's: one allocation per step, as before)

    def _state(self):
        st = self.env.world._packed_state()
        return st.data_ptr(), st.shape[-1]

    @staticmethod
    def _check_out(out, name, shape, dtype, dev):
        t = out[name]
        assert tuple(t.shape) == tuple(shape) and t.dtype == dtype and t.is_contiguous() and t.device == dev, (
            f"rollout output {name!r}: expected a contiguous {dtype} tensor of shape {tuple(shape)} on {dev}, got "
            f"{t.dtype} {tuple(t.shape)} on {t.device}")
        return t

    def _rollout_out(self, fields, out):
        """The per-step outputs of a K-step launch: allocated here, or the caller's (``out``: a dict with exactly these
        names, shapes and dtypes - e.g. views of one rollout buffer that is gathered across GPUs as it is, shard.PackedRollout)."""
        if out is None:
            return {name: torch.empty(shape, device=self.dev, dtype=dtype) for name, shape, dtype in fields}
        return {name: self._check_out(out, name, shape, dtype, self.dev) for name, shape, dtype in fields}

    def rollout_fields(self, n_steps: int):
        """(name, shape, dtype) of every per-step output of ``prepare_rollout(n_steps)``."""
        raise NotImplementedError

    def prepare_rollout(self, n_steps: int, out=None):
        """(descriptor, buffers, outputs) for ``vmas_world_rollout_env``: every per-step output gets a leading
        ``n_steps`` axis ([K, n_agents, B, D] observations, [K, n_agents, B] rewards, [K, B] dones ...), the buffer
        struct points at step 0."""
        raise NotImplementedError


class BalancePost(_Post):
    def __init__(self, env):
        super().__init__(env)
        sc = env.scenario
        d = A.BalanceDesc()
        d.n_agents = self.n
        d.goal, d.package, d.line, d.floor = (e._index for e in (sc.goal, sc.package, sc.line, sc.floor))
        d.agent0 = env.world.agents[0]._index
        d.goal_radius, d.package_radius = sc.goal.shape.radius, sc.package.shape.radius
        d.line_length = sc.line.shape.length
        d.floor_length, d.floor_width = sc.floor.shape.length, sc.floor.shape.width
        d.shaping_factor, d.fall_reward = sc.shaping_factor, sc.fall_reward
        self.desc = d

    kind = A.POST_BALANCE

    def persistent_tensors(self):
        return [self.env.scenario.global_shaping]

    def bound_attributes(self):
        sc = self.env.scenario
        sc = getattr(sc, "_sc", sc)  # (attached_env._ScenarioView: the reference's scenario behind it)
        return [(sc, "pos_rew"), (sc, "ground_rew"), (sc, "on_the_ground")]

    def _new_set(self):
        n, B = self.n, self.B
        st = _OutSet()
        # observations | rewards + the two info terms | done + on_the_ground
        st.base, v = self._carve([("obs", (n, B, 16), torch.float32), ("fl32", (n + 2, B), torch.float32),
                                  ("flags", (2, B), torch.bool)])
        rows = v["fl32"].unbind(0)
        done, on_ground = v["flags"].unbind(0)
        st.v = v
        st.obs, st.rew, st.done = list(v["obs"].unbind(0)), list(rows[:n]), done
        st.extra = (rows[n], rows[n + 1], on_ground)  # pos_rew, ground_rew, on_the_ground
        st.infos = {"pos_rew": rows[n], "ground_rew": rows[n + 1]}
        st.tensors = tuple(st.obs) + tuple(rows) + (done, on_ground)
        b = st.buffers = A.BalanceBuffers()
        b.limit = self._limit()
        p32, pfl = v["fl32"].data_ptr(), v["flags"].data_ptr()
        b.obs, b.rew, b.done = v["obs"].data_ptr(), p32, pfl
        b.pos_rew, b.ground_rew, b.on_the_ground = p32 + 4 * n * B, p32 + 4 * (n + 1) * B, pfl + B
        return st

    def prepare(self, dedicated: bool = False):
        """(descriptor, buffers, what env.step returns) - nothing launched."""
        sc = self.env.scenario
        st = self._acquire(dedicated)
        sc.pos_rew, sc.ground_rew, sc.on_the_ground = st.extra
        st.buffers.global_shaping = sc.global_shaping.data_ptr()
        info = st.infos
        return self.desc, st.buffers, (list(st.obs), list(st.rew), st.done, [dict(info) for _ in range(self.n)])

    def rollout_fields(self, n_steps: int):
        K, n, B = int(n_steps), self.n, self.B
        return [("obs", (K, n, B, 16), torch.float32), ("rew", (K, n, B), torch.float32), ("done", (K, B), torch.bool),
                ("pos_rew", (K, B), torch.float32), ("ground_rew", (K, B), torch.float32)]

    def prepare_rollout(self, n_steps: int, out=None):
        sc = self.env.scenario
        out = self._rollout_out(self.rollout_fields(n_steps), out)
        sc.on_the_ground = torch.empty(self.B, device=self.dev, dtype=torch.bool)
        sc.pos_rew, sc.ground_rew = out["pos_rew"][-1], out["ground_rew"][-1]  # scenario attributes: the last step's
        b = A.BalanceBuffers()
        b.limit = self._limit()
        b.global_shaping = sc.global_shaping.data_ptr()
        b.obs, b.rew, b.done = out["obs"].data_ptr(), out["rew"].data_ptr(), out["done"].data_ptr()
        b.pos_rew, b.ground_rew = out["pos_rew"].data_ptr(), out["ground_rew"].data_ptr()
        b.on_the_ground = sc.on_the_ground.data_ptr()
        return self.desc, b, out

    def __call__(self):
        desc, b, result = self.prepare()
        st, ld = self._state()
        _check(self.lib.vmas_balance_post_step(C.byref(desc), C.byref(b), self.B, st, ld, _stream(self.dev)))
        return result


class TransportPost(_Post):
    def __init__(self, env):
        super().__init__(env)
        sc = env.scenario
        P = len(sc.packages)
        assert P <= A.ENV_MAX_PACKAGES
        d = A.TransportDesc()
        d.n_agents, d.n_packages = self.n, P
        d.goal, d.package0, d.agent0 = sc.goal._index, sc.packages[0]._index, env.world.agents[0]._index
        assert [p._index for p in sc.packages] == list(range(d.package0, d.package0 + P))
        d.goal_radius = sc.goal.shape.radius
        d.package_length, d.package_width = sc.packages[0].shape.length, sc.packages[0].shape.width
        d.shaping_factor = sc.shaping_factor
        self.desc = d
        self.P = P
        self.D = 4 + 7 * P
        # the packages' persistent terms live in one [P, B] block each; the objects hold row views
        self.global_shaping = torch.stack([p.global_shaping for p in sc.packages]).contiguous()
        self.on_goal = torch.stack([p.on_goal for p in sc.packages]).contiguous()
        self._gs_rows, self._og_rows = list(self.global_shaping.unbind(0)), list(self.on_goal.unbind(0))
        self._gs_ptr, self._og_ptr = self.global_shaping.data_ptr(), self.on_goal.data_ptr()
        self._bind()

    def _bind(self):
        for i, p in enumerate(self.env.scenario.packages):
            p.global_shaping = self._gs_rows[i]
            p.on_goal = self._og_rows[i]

    kind = A.POST_TRANSPORT

    def persistent_tensors(self):
        return [self.global_shaping, self.on_goal]

    def bound_attributes(self):
        sc = self.env.scenario
        return [(getattr(sc, "_sc", sc), "rew")]

    def _rebind(self):
        for i, p in enumerate(self.env.scenario.packages):  # reset() may have rebound them
            if p.global_shaping is not self._gs_rows[i]:
                self._gs_rows[i].copy_(p.global_shaping)
                p.global_shaping = self._gs_rows[i]
            if p.on_goal is not self._og_rows[i]:
                self._og_rows[i].copy_(p.on_goal)
                p.on_goal = self._og_rows[i]

    def _new_set(self):
        n, B = self.n, self.B
        st = _OutSet()
        st.base, v = self._carve([("obs", (n, B, self.D), torch.float32), ("rew", (n, B), torch.float32), ("done", (B,), torch.bool)])
        st.v = v
        st.obs, st.rew, st.done = list(v["obs"].unbind(0)), list(v["rew"].unbind(0)), v["done"]
        st.extra, st.infos = None, None
        st.tensors = tuple(st.obs) + tuple(st.rew) + (st.done,)
        b = st.buffers = A.TransportBuffers()
        b.limit = self._limit()
        b.global_shaping, b.on_goal = self._gs_ptr, self._og_ptr
        b.obs, b.rew, b.done = v["obs"].data_ptr(), v["rew"].data_ptr(), v["done"].data_ptr()
        return st

    def prepare(self, dedicated: bool = False):
        self._rebind()
        st = self._acquire(dedicated)
        self.env.scenario.rew = st.rew[0]
        return self.desc, st.buffers, (list(st.obs), list(st.rew), st.done, [{} for _ in range(self.n)])

    def rollout_fields(self, n_steps: int):
        K = int(n_steps)
        return [("obs", (K, self.n, self.B, self.D), torch.float32), ("rew", (K, self.n, self.B), torch.float32),
                ("done", (K, self.B), torch.bool)]

    def prepare_rollout(self, n_steps: int, out=None):
        self._rebind()  # (re-binds the packages' persistent terms)
        out = self._rollout_out(self.rollout_fields(n_steps), out)
        self.env.scenario.rew = out["rew"][-1, 0]
        b = A.TransportBuffers()
        b.limit = self._limit()
        b.global_shaping, b.on_goal = self._gs_ptr, self._og_ptr
        b.obs, b.rew, b.done = out["obs"].data_ptr(), out["rew"].data_ptr(), out["done"].data_ptr()
        return self.desc, b, out

    def __call__(self):
        desc, b, result = self.prepare()
        st, ld = self._state()
        _check(self.lib.vmas_transport_post_step(C.byref(desc), C.byref(b), self.B, st, ld, _stream(self.dev)))
        return result


class NavigationPost(_Post):
    ONE_LAUNCH_MAX_TILES_PER_CU = 64  # the one-launch step is used up to this many tiles per CU (measured, see __init__)

    @staticmethod
    def supports(env) -> Optional[str]:
        sc = env.scenario
        agents = env.world.agents
        if len(agents) > A.ENV_MAX_AGENTS:
            return "too many agents"
        if len({a.shape.radius for a in agents}) != 1 or len({a.goal.shape.radius for a in agents}) != 1:
            return "non-uniform radii"
        if sc.collisions and len({(s._angles.shape[-1], s._max_range) for a in agents for s in a.sensors}) != 1:
            return "non-uniform sensors"
        return None

    def __init__(self, env):
        super().__init__(env)
        sc, w = env.scenario, env.world
        agents = w.agents
        d = A.NavigationDesc()
        d.n_agents, d.agent0 = self.n, agents[0]._index
        assert [a._index for a in agents] == list(range(d.agent0, d.agent0 + self.n))
        for i, a in enumerate(agents):
            d.goal_of[i] = a.goal._index
        d.shared_rew, d.collisions, d.observe_all_goals = int(sc.shared_rew), int(sc.collisions), int(sc.observe_all_goals)
        d.agent_radius, d.goal_radius = agents[0].shape.radius, agents[0].goal.shape.radius
        d.pos_shaping_factor, d.final_reward = sc.pos_shaping_factor, sc.final_reward
        d.agent_collision_penalty, d.min_collision_distance = sc.agent_collision_penalty, sc.min_collision_distance
        if sc.collisions:
            s = agents[0].sensors[0]
            d.n_rays, d.lidar_range = s._angles.shape[-1], s._max_range
        self.desc = d
        self._side = None
        # as the physics kernel's epilogue only if the library says this world allows it (sensors as the epilogue casts
        # them, tile + scratch within the CU's LDS); otherwise the separate launches of __call__
        # (round 2 kept the one-launch step to one tile per CU: its epilogue staged 43 KB of observation and ray rows in
        # LDS and lost to the separate launches beyond that.  With the lane-compacted LIDAR and the block writer -
        # DESIGN.md 3.4 - it wins at every size measured: 65 536 environments 64 us against 99 in four launches)
        n_cu = torch.cuda.get_device_properties(self.dev).multi_processor_count
        # several steps per launch (rollout): a grid barrier per step - every tile must be resident at once
        self.rollout_ok = (self.B + 63) // 64 <= n_cu or not sc.collisions
        if (self.B + 63) // 64 > self.ONE_LAUNCH_MAX_TILES_PER_CU * n_cu or \
                self.lib.vmas_world_step_env_check(w._get_backend()._h, A.POST_NAVIGATION, C.byref(d)) != 0:
            self.kind = None
        self.obs_dim = 4 + 2 * (self.n if sc.observe_all_goals else 1) + (d.n_rays if sc.collisions else 0)
        self.pos_shaping = torch.stack([a.pos_shaping for a in agents]).contiguous()
        self._shaping_rows = list(self.pos_shaping.unbind(0))
        for a, row in zip(agents, self._shaping_rows):
            a.pos_shaping = row
        self._pair_index_ptr = None
        if sc.collisions:  # (i, j) -> index in the world's static pair list, for World.collides' global reduction
            spec = w.spec
            where = {}
            for k, p in enumerate(spec.pairs):
                where[(p.a, p.b)] = where[(p.b, p.a)] = k
            table = [[where.get((a._index, b._index), -1) for b in agents] for a in agents]
            self.pair_index = torch.tensor(table, dtype=torch.int32, device=self.dev).contiguous()
            self._pair_index_ptr = self.pair_index.data_ptr()

    def persistent_tensors(self):
        return [self.pos_shaping]

    kind = A.POST_NAVIGATION  # as the epilogue of the physics kernel: LIDAR cast and collision reduction in the launch
    rollout_ok = True         # (instance attribute, see __init__: several steps per launch need a grid barrier per step)

    def _rebind_shaping(self):
        for a, row in zip(self.env.world.agents, self._shaping_rows):  # reset() may have rebound it
            if a.pos_shaping is not row:
                row.copy_(a.pos_shaping)
                a.pos_shaping = row

    def rollout_fields(self, n_steps: int):
        K, n, B = int(n_steps), self.n, self.B
        f32 = torch.float32
        return [("obs", (K, n, B, self.obs_dim), f32), ("rew", (K, n, B), f32), ("done", (K, B), torch.bool),
                ("agent_pos_rew", (K, n, B), f32), ("pos_rew", (K, B), f32), ("final_rew", (K, B), f32),
                ("agent_collisions", (K, n, B), f32)]

    def prepare_rollout(self, n_steps: int, out=None):
        env, sc = self.env, self.env.scenario
        self._rebind_shaping()  # (re-binds the agents' shaping rows after a reset)
        out = self._rollout_out(self.rollout_fields(n_steps), out)
        sc.pos_rew, sc.final_rew = out["pos_rew"][-1], out["final_rew"][-1]  # scenario / agent attributes: the last step's
        for i, a in enumerate(env.world.agents):
            a.pos_rew, a.agent_collision_rew = out["agent_pos_rew"][-1, i], out["agent_collisions"][-1, i]
        sc._lidar_cache = None
        b = A.NavigationBuffers()
        b.pos_shaping, b.limit = self.pos_shaping.data_ptr(), self._limit()
        b.pair_index = self._pair_index_ptr
        b.obs, b.rew, b.done = out["obs"].data_ptr(), out["rew"].data_ptr(), out["done"].data_ptr()
        b.agent_pos_rew, b.collision_rew = out["agent_pos_rew"].data_ptr(), out["agent_collisions"].data_ptr()
        b.pos_rew, b.final_rew = out["pos_rew"].data_ptr(), out["final_rew"].data_ptr()
        return self.desc, b, out

    def _new_set(self):
        n, B, sc = self.n, self.B, self.env.scenario
        st = _OutSet()
        # observations | rewards | agent_pos_rew [n] | collision_rew [n] | pos_rew | final_rew | done
        st.base, v = self._carve([("obs", (n, B, self.obs_dim), torch.float32), ("rew", (n, B), torch.float32),
                                  ("terms", (2 * n + 2, B), torch.float32), ("done", (B,), torch.bool)])
        st.v = v
        rows = v["terms"].unbind(0)
        st.obs, st.rew, st.done = list(v["obs"].unbind(0)), list(v["rew"].unbind(0)), v["done"]
        st.extra = rows
        pos_rew, final_rew = rows[2 * n], rows[2 * n + 1]
        st.infos = [{"pos_rew": pos_rew if sc.shared_rew else rows[i], "final_rew": final_rew, "agent_collisions": rows[n + i]}
                    for i in range(n)]
        st.tensors = tuple(st.obs) + tuple(st.rew) + tuple(rows) + (st.done,)
        b = st.buffers = A.NavigationBuffers()
        b.pos_shaping, b.limit = self.pos_shaping.data_ptr(), self._limit()
        b.pair_index = self._pair_index_ptr
        b.obs, b.rew, b.done = v["obs"].data_ptr(), v["rew"].data_ptr(), v["done"].data_ptr()
        p, row_bytes = v["terms"].data_ptr(), 4 * B
        b.agent_pos_rew, b.collision_rew = p, p + n * row_bytes
        b.pos_rew, b.final_rew = p + 2 * n * row_bytes, p + (2 * n + 1) * row_bytes
        return st

    def bound_attributes(self):
        sc = self.env.scenario
        sc = getattr(sc, "_sc", sc)
        out = [(sc, "pos_rew"), (sc, "final_rew")]
        for a in self.env.world.agents:
            out += [(a, "pos_rew"), (a, "agent_collision_rew")]
        return out

    def _bind_outputs(self, dedicated: bool = False):
        """The output set of this step, bound to the scenario's / agents' attributes (nothing launched)."""
        sc, n = self.env.scenario, self.n
        self._rebind_shaping()
        st = self._acquire(dedicated)
        rows = st.extra
        sc.pos_rew, sc.final_rew = rows[2 * n], rows[2 * n + 1]
        for i, a in enumerate(self.env.world.agents):
            a.pos_rew, a.agent_collision_rew = rows[i], rows[n + i]
        return st, (list(st.obs), list(st.rew), st.done, [dict(d) for d in st.infos])

    def prepare(self, dedicated: bool = False):
        """(descriptor, buffers, what env.step returns) for the one-launch step (vmas_world_step_env)."""
        st, result = self._bind_outputs(dedicated)
        self.env.scenario._lidar_cache = None  # (the sensors' measurements are made inside the launch, on the LDS tile)
        st.buffers.lidar = st.buffers.pair_any = None
        return self.desc, st.buffers, result

    def __call__(self):
        env, sc, w = self.env, self.env.scenario, self.env.world
        st, result = self._bind_outputs()
        b = st.buffers
        main = torch.cuda.current_stream(self.dev)
        if sc.collisions:
            # the batch-global collision mask (a short latency-bound kernel) runs beside the LIDAR cast on a
            # second stream: both only read the new state
            be = w._get_backend()
            if self._side is None:
                self._side = torch.cuda.Stream(self.dev)
                self._fork, self._join = torch.cuda.Event(), torch.cuda.Event()
            self._fork.record(main)
            self._side.wait_event(self._fork)
            pair_any = be.pair_mask(stream=self._side)
            self._join.record(self._side)
            lidar = be.cast_rays(stream=main)  # [n_sensors, max_rays, ld] of the post-step state
            env._lidar_cache = sc._lidar_cache = lidar
            main.wait_event(self._join)
            b.lidar, b.lidar_max_rays = lidar.data_ptr(), lidar.shape[1]
            b.pair_any = pair_any.data_ptr()
        state = w._packed_state()
        _check(self.lib.vmas_navigation_post_step(C.byref(self.desc), C.byref(b), self.B, state.data_ptr(), state.shape[-1],
                                                  main.cuda_stream))
        return result


class FootballPost(_Post):
    """football.py:1121-1515 for the learning-vs-learning game (scenarios/football.py)."""

    TERMS = ("sparse_reward_blue", "pos_rew_blue", "pos_rew_red", "pos_rew_agent_blue", "pos_rew_agent_red",
             "min_agent_dist_to_ball_blue", "min_agent_dist_to_ball_red", "dist_ball_to_goal_blue", "dist_ball_to_goal_red")

    def __init__(self, env):
        super().__init__(env)
        sc, w = env.scenario, env.world
        d = A.FootballDesc()
        d.n_blue, d.n_red, d.agent0 = len(sc.blue_agents), len(sc.red_agents), sc.blue_agents[0]._index
        team = sc.blue_agents + sc.red_agents + [sc.ball]
        assert [a._index for a in team] == list(range(d.agent0, d.agent0 + len(team)))
        assert [a._agent_index for a in team] == list(range(len(team)))
        d.observe_teammates, d.observe_adversaries = int(sc.observe_teammates), int(sc.observe_adversaries)
        d.dense_reward = int(sc.dense_reward)
        d.goal_x = sc.pitch_length / 2 + sc.ball_size / 2
        d.goal_half = sc.goal_size / 2
        d.touch_dist = sc.agent_size + sc.ball_size + 1e-2
        d.pos_shaping_factor_ball_goal = sc.pos_shaping_factor_ball_goal
        d.pos_shaping_factor_agent_ball = sc.pos_shaping_factor_agent_ball
        d.distance_to_ball_trigger, d.scoring_reward = sc.distance_to_ball_trigger, sc.scoring_reward
        self.desc = d
        self.obs_dims = [16 + 8 * ((d.n_red if blue else d.n_blue) * d.observe_adversaries
                                   + ((d.n_blue if blue else d.n_red) - 1) * d.observe_teammates)
                         for blue in [True] * d.n_blue + [False] * d.n_red]
        assert len(set(self.obs_dims)) == 1, "the fused kernel writes one [n_agents, batch, obs_dim] block"
        self.obs_dim = self.obs_dims[0]
        ball = sc.ball
        names = ("pos_shaping_blue", "pos_shaping_red", "pos_shaping_agent_blue", "pos_shaping_agent_red")
        self.pos_shaping = torch.stack([getattr(ball, n) for n in names]).contiguous()
        self._shaping_rows = list(self.pos_shaping.unbind(0))
        for n, row in zip(names, self._shaping_rows):  # the ball's shaping terms live in one [4, B] block
            setattr(ball, n, row)
        self._shaping_names = names
        self._blue = [a in sc.blue_agents for a in env.agents]
        # as the epilogue of the step kernel (one launch per Environment.step, K steps per launch in rollout()): worlds that
        # run the lane-compacted kernel (csrc/vmas_compact.h) - the default for football
        be = w._get_backend()
        self.kind = A.POST_FOOTBALL if (be.compact and self.lib.vmas_world_step_env_check(be._h, A.POST_FOOTBALL, C.byref(d)) == 0) else None

    def persistent_tensors(self):
        return [self.pos_shaping]

    def _rebind_shaping(self):
        ball = self.env.scenario.ball
        for n, row in zip(self._shaping_names, self._shaping_rows):  # reset() may have rebound them
            cur = getattr(ball, n)
            if cur is not row:
                row.copy_(cur)
                setattr(ball, n, row)

    def _new_set(self):
        n, B = self.n, self.B
        st = _OutSet()
        st.base, v = self._carve([("obs", (n, B, self.obs_dim), torch.float32), ("rew", (n, B), torch.float32),
                                  ("terms", (len(self.TERMS), B), torch.float32), ("touching", (2, B), torch.bool),
                                  ("done", (B,), torch.bool)])
        st.v = v
        st.obs, st.rew, st.done = list(v["obs"].unbind(0)), list(v["rew"].unbind(0)), v["done"]
        terms, touching = v["terms"].unbind(0), v["touching"].unbind(0)
        t = dict(zip(self.TERMS, terms))
        st.extra = t
        st.infos = []
        for blue in self._blue:
            side = "blue" if blue else "red"
            st.infos.append({
                "sparse_reward": t["sparse_reward_blue"],  # (red: negated in finish(), once the kernel has been enqueued)
                "ball_goal_pos_rew": t[f"pos_rew_{side}"], "all_agent_ball_pos_rew": t[f"pos_rew_agent_{side}"],
                "ball_pos": None, "dist_ball_to_goal": t[f"dist_ball_to_goal_{side}"],
                "min_agent_dist_to_ball": t[f"min_agent_dist_to_ball_{side}"],
                "touching_ball": touching[0 if blue else 1],
            })
        st.tensors = tuple(st.obs) + tuple(st.rew) + tuple(terms) + tuple(touching) + (st.done,)
        b = st.buffers = A.FootballBuffers()
        b.limit = self._limit()
        b.pos_shaping = self.pos_shaping.data_ptr()
        b.obs, b.rew, b.done = v["obs"].data_ptr(), v["rew"].data_ptr(), v["done"].data_ptr()
        b.terms, b.touching = v["terms"].data_ptr(), v["touching"].data_ptr()
        return st

    def bound_attributes(self):
        sc = self.env.scenario
        sc = getattr(sc, "_sc", sc)
        ball = sc.ball
        return [(sc, "_sparse_reward_blue"), (sc, "_done"), (sc, "min_agent_dist_to_ball_blue"), (sc, "min_agent_dist_to_ball_red"),
                (ball, "pos_rew_blue"), (ball, "pos_rew_red"), (ball, "pos_rew_agent_blue"), (ball, "pos_rew_agent_red")]

    def prepare(self, dedicated: bool = False):
        """(descriptor, buffers, what env.step returns) - outputs bound to the scenario's attributes, nothing launched.
        ``finish(result)`` completes the infos once the step has been enqueued."""
        sc, w = self.env.scenario, self.env.world
        ball = sc.ball
        self._rebind_shaping()
        st = self._acquire(dedicated)
        st.buffers.agent_ft = w._packed_agent_ft().data_ptr()
        t = st.extra
        sc._sparse_reward_blue, sc._done = t["sparse_reward_blue"], st.done
        ball.pos_rew_blue, ball.pos_rew_red = t["pos_rew_blue"], t["pos_rew_red"]
        ball.pos_rew_agent_blue, ball.pos_rew_agent_red = t["pos_rew_agent_blue"], t["pos_rew_agent_red"]
        sc.min_agent_dist_to_ball_blue, sc.min_agent_dist_to_ball_red = (
            t["min_agent_dist_to_ball_blue"], t["min_agent_dist_to_ball_red"])
        return self.desc, st.buffers, (list(st.obs), list(st.rew), st.done, [dict(d) for d in st.infos])

    def finish(self, result):
        """The parts of the infos that are tensor ops on the step's outputs (stream-ordered behind the launch)."""
        env, sc = self.env, self.env.scenario
        ball = sc.ball
        ball_pos = ball.state.pos if self.static_outputs else ball.state.pos.clone()  # (not a live view of the state)
        sparse_red = None
        for blue, info in zip(self._blue, result[3]):
            info["ball_pos"] = ball_pos
            if not blue:
                if sparse_red is None:
                    sparse_red = sc._sparse_reward_red = -sc._sparse_reward_blue
                info["sparse_reward"] = sparse_red
        return result

    def rollout_fields(self, n_steps: int):
        K, n, B = int(n_steps), self.n, self.B
        return [("obs", (K, n, B, self.obs_dim), torch.float32), ("rew", (K, n, B), torch.float32), ("done", (K, B), torch.bool),
                ("terms", (K, len(self.TERMS), B), torch.float32), ("touching_ball", (K, 2, B), torch.bool)]

    def prepare_rollout(self, n_steps: int, out=None):
        env, sc = self.env, self.env.scenario
        self._rebind_shaping()
        out = dict(self._rollout_out(self.rollout_fields(n_steps), out))
        b = A.FootballBuffers()
        b.limit = self._limit()
        b.pos_shaping = self.pos_shaping.data_ptr()
        b.obs, b.rew, b.done = out["obs"].data_ptr(), out["rew"].data_ptr(), out["done"].data_ptr()
        b.terms, b.touching = out["terms"].data_ptr(), out["touching_ball"].data_ptr()
        b.agent_ft = env.world._packed_agent_ft().data_ptr()
        t = dict(zip(self.TERMS, out["terms"][-1].unbind(0)))  # scenario / ball attributes: the last step's
        ball = sc.ball
        sc._sparse_reward_blue, sc._done = t["sparse_reward_blue"], out["done"][-1]
        ball.pos_rew_blue, ball.pos_rew_red = t["pos_rew_blue"], t["pos_rew_red"]
        ball.pos_rew_agent_blue, ball.pos_rew_agent_red = t["pos_rew_agent_blue"], t["pos_rew_agent_red"]
        sc.min_agent_dist_to_ball_blue, sc.min_agent_dist_to_ball_red = (
            t["min_agent_dist_to_ball_blue"], t["min_agent_dist_to_ball_red"])
        for i, name in enumerate(self.TERMS):
            out[name] = out["terms"][:, i]
        return self.desc, b, out

    def __call__(self):
        desc, b, result = self.prepare()
        st, ld = self._state()
        _check(self.lib.vmas_football_post_step(C.byref(desc), C.byref(b), self.B, st, ld, _stream(self.dev)))
        return self.finish(result)
