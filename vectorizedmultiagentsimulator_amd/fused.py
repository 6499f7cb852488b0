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
        self.err = torch.zeros(1, device=env.device, dtype=torch.int32)
        self._ft = w._packed_agent_ft()
        self._keep = None
        self._fast = None

    def prepare(self, actions: List[Tensor]):
        """Check the action tensors and point the slots at them (no launch)."""
        env = self.env
        held = []
        fast = self._fast
        if fast is None:  # per agent: (action columns, dtype) - what a well-formed action tensor looks like
            dtype = torch.float32 if env.continuous_actions else torch.int64
            fast = self._fast = [(env.get_agent_action_size(a), dtype) for a in env.agents]
        B, dev, cont = env.num_envs, env.device, env.continuous_actions
        for i, (agent, act) in enumerate(zip(env.agents, actions)):
            want, dtype = fast[i]
            if (type(act) is Tensor and act.dtype is dtype and act.dim() == 2 and act.shape[0] == B and act.shape[1] == want
                    and act.is_contiguous() and act.device == dev):  # the usual case: nothing to convert
                held.append(act)
                if cont:
                    self.args.agents[i].action = act.data_ptr()
                else:
                    self.args.agents[i].action_index = act.data_ptr()
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

    def check(self):
        """The reference asserts on the host (environment.py:621,651-653): one sync, not 2 per agent."""
        flags = int(self.err.item())
        if flags:
            self.err.zero_()
            assert not (flags & A.ACTION_ERR_NAN), "actions contain NaN"
            raise AssertionError("Physical actions of an agent are out of its range")

    def launch(self, validate: bool):
        """The stand-alone ingest kernel on the prepared action tensors (+ the reference's asserts when ``validate``)."""
        env = self.env
        ft = env.world._packed_agent_ft()
        err = self.err.data_ptr() if validate else None
        _check(self.lib.vmas_env_ingest_actions(C.byref(self.args), env.num_envs, env.world._packed_state().data_ptr(),
                                                ft.data_ptr(), ft.shape[-1], err, _stream(env.device)))
        if validate:
            self.check()

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

    def _bind(self):
        w = self.env.world
        be = self._be = w._get_backend()
        self._h = be._h
        self._st, self._ft, self._ld = C.c_void_p(be.state.data_ptr()), C.c_void_p(be.agent_ft.data_ptr()), be.ld
        spec = w.spec
        self._per_env = any(j.per_env_fixed_rotation for j in spec.joints) or any(e.per_env_gravity for e in spec.entities)
        self._exact = bool(w.exact_broad_phase)
        self._ing = C.byref(self.ingest.args)
        self._err = C.c_void_p(self.ingest.err.data_ptr())
        self._dev = self.env.device

    def __call__(self, kind: int, desc, buffers, validate: bool):
        w = self.env.world
        if self._be is None or w._backend is not self._be:
            self._bind()
        w._query_cache = None
        args = None
        if self._per_env or self._exact:
            jfr, eg = w._per_env_inputs() if self._per_env else (None, None)
            sa = A.StepArgs()
            sa.joint_fixed_rot = jfr.data_ptr() if jfr is not None else None
            sa.entity_gravity = eg.data_ptr() if eg is not None else None
            sa.exact_broad_phase = 1 if self._exact else 0
            args = C.byref(sa)
        if desc is not self._cd or buffers is not self._cb:  # (the structs persist: their references are built once)
            self._cd, self._cb = desc, buffers
            self._rd = C.byref(desc) if desc is not None else None
            self._rb = C.byref(buffers) if buffers is not None else None
        rc = self.fn(self._h, self._st, self._ft, self._ld, args, self._ing, self._err if validate else None, kind,
                     self._rd, self._rb, _stream(self._dev))
        if rc != 0:
            raise VmasHipError(A.last_error())


    def rollout(self, kind: int, desc, buffers, n_steps: int):
        """``vmas_world_rollout_env``: n_steps Environment.step() calls in one launch."""
        w = self.env.world
        if self._be is None or w._backend is not self._be:
            self._bind()
        w._query_cache = None
        args = None
        if self._per_env or self._exact:
            jfr, eg = w._per_env_inputs() if self._per_env else (None, None)
            sa = A.StepArgs()
            sa.joint_fixed_rot = jfr.data_ptr() if jfr is not None else None
            sa.entity_gravity = eg.data_ptr() if eg is not None else None
            sa.exact_broad_phase = 1 if self._exact else 0
            args = C.byref(sa)
        rc = A.load_library().vmas_world_rollout_env(
            self._h, self._st, self._ft, self._ld, args, self._ing, None, kind,
            C.byref(desc) if desc is not None else None, C.byref(buffers) if buffers is not None else None, int(n_steps),
            _stream(self._dev))
        if rc != 0:
            raise VmasHipError(A.last_error())


class _Post:
    """Shared plumbing of the per-scenario post-step kernels."""

    kind = None  # A.POST_*: the post-step can also run as the epilogue of the physics kernel

    def __init__(self, env):
        self.env = env
        self.lib = A.load_library()
        self.B = env.num_envs
        self.dev = env.device
        self.n = len(env.agents)
        self.static_outputs = False
        self._out = None

    def persistent_tensors(self) -> List[Tensor]:
        """Scenario tensors the kernel reads AND writes (shaping terms ...)."""
        raise NotImplementedError

    def _limit(self) -> A.StepLimit:
        lim = A.StepLimit()
        lim.steps = self.env.steps.data_ptr()
        lim.max_steps = float(self.env.max_steps) if self.env.max_steps is not None else -1.0
        return lim

    def _buffers(self, cls):
        """The kernel's buffer struct, built once; per step only the pointers that change are rewritten."""
        b = getattr(self, "_b", None)
        if b is None:
            b = self._b = cls()
            b.limit = self._limit()
        return b

    def _outputs(self, obs_dim: int):
        if self._out is None or not self.static_outputs:
            self._out = (
                torch.empty(self.n, self.B, obs_dim, device=self.dev, dtype=torch.float32),
                torch.empty(self.n, self.B, device=self.dev, dtype=torch.float32),
                torch.empty(self.B, device=self.dev, dtype=torch.bool),
            )
        return self._out

    def _state(self):
        st = self.env.world._packed_state()
        return st.data_ptr(), st.shape[-1]

    def prepare_rollout(self, n_steps: int):
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

    def prepare(self):
        """(descriptor, buffers, what env.step returns) - outputs allocated, nothing launched."""
        sc, n, B = self.env.scenario, self.n, self.B
        if self._out is None or not self.static_outputs:
            # three allocations per step: observations | rewards + the two info terms | done + on_the_ground
            # (every torch call costs the host ~1.5 us and the step is host-bound around a 14 us kernel)
            self._out = (torch.empty(n, B, 16, device=self.dev, dtype=torch.float32),
                         torch.empty(n + 2, B, device=self.dev, dtype=torch.float32),
                         torch.empty(2, B, device=self.dev, dtype=torch.bool))
        obs, fl32, flags = self._out
        rows = fl32.unbind(0)
        done, sc.on_the_ground = flags.unbind(0)
        sc.pos_rew, sc.ground_rew = rows[n], rows[n + 1]
        b = self._buffers(A.BalanceBuffers)
        b.global_shaping = sc.global_shaping.data_ptr()
        p32, pfl = fl32.data_ptr(), flags.data_ptr()
        b.obs, b.rew, b.done = obs.data_ptr(), p32, pfl
        b.pos_rew, b.ground_rew, b.on_the_ground = p32 + 4 * n * B, p32 + 4 * (n + 1) * B, pfl + B
        info = {"pos_rew": sc.pos_rew, "ground_rew": sc.ground_rew}
        return self.desc, b, (list(obs.unbind(0)), list(rows[:n]), done, [dict(info) for _ in range(n)])

    def prepare_rollout(self, n_steps: int):
        sc, K = self.env.scenario, int(n_steps)
        out = {
            "obs": torch.empty(K, self.n, self.B, 16, device=self.dev), "rew": torch.empty(K, self.n, self.B, device=self.dev),
            "done": torch.empty(K, self.B, device=self.dev, dtype=torch.bool),
            "pos_rew": torch.empty(K, self.B, device=self.dev), "ground_rew": torch.empty(K, self.B, device=self.dev),
        }
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
        # the packages' persistent terms live in one [P, B] block each; the objects hold row views
        self.global_shaping = torch.stack([p.global_shaping for p in sc.packages]).contiguous()
        self.on_goal = torch.stack([p.on_goal for p in sc.packages]).contiguous()
        self._bind()

    def _bind(self):
        for i, p in enumerate(self.env.scenario.packages):
            p.global_shaping = self.global_shaping[i]
            p.on_goal = self.on_goal[i]

    kind = A.POST_TRANSPORT

    def persistent_tensors(self):
        return [self.global_shaping, self.on_goal]

    def prepare(self):
        sc = self.env.scenario
        for i, p in enumerate(sc.packages):  # reset() may have rebound them
            if p.global_shaping.data_ptr() != self.global_shaping[i].data_ptr():
                self.global_shaping[i].copy_(p.global_shaping)
                p.global_shaping = self.global_shaping[i]
            if p.on_goal.data_ptr() != self.on_goal[i].data_ptr():
                self.on_goal[i].copy_(p.on_goal)
                p.on_goal = self.on_goal[i]
        obs, rew, done = self._outputs(4 + 7 * self.P)
        b = self._buffers(A.TransportBuffers)
        b.global_shaping, b.on_goal = self.global_shaping.data_ptr(), self.on_goal.data_ptr()
        b.obs, b.rew, b.done = obs.data_ptr(), rew.data_ptr(), done.data_ptr()
        sc.rew = rew[0]
        return self.desc, b, (list(obs.unbind(0)), list(rew.unbind(0)), done, [{} for _ in range(self.n)])

    def prepare_rollout(self, n_steps: int):
        K, D = int(n_steps), 4 + 7 * self.P
        self.prepare()  # (re-binds the packages' persistent terms)
        out = {
            "obs": torch.empty(K, self.n, self.B, D, device=self.dev), "rew": torch.empty(K, self.n, self.B, device=self.dev),
            "done": torch.empty(K, self.B, device=self.dev, dtype=torch.bool),
        }
        self.env.scenario.rew = out["rew"][-1, 0]
        b = A.TransportBuffers()
        b.limit = self._limit()
        b.global_shaping, b.on_goal = self.global_shaping.data_ptr(), self.on_goal.data_ptr()
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
        if sc.collisions and len({(s._angles.shape[0], s._max_range) for a in agents for s in a.sensors}) != 1:
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
            d.n_rays, d.lidar_range = s._angles.shape[0], s._max_range
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
        self._shaping_ptrs = [r.data_ptr() for r in self._shaping_rows]
        for a, row in zip(agents, self._shaping_rows):
            a.pos_shaping = row
        self._buf = A.NavigationBuffers()
        self._buf.pos_shaping = self.pos_shaping.data_ptr()
        self._buf.limit = self._limit()
        if sc.collisions:  # (i, j) -> index in the world's static pair list, for World.collides' global reduction
            spec = w.spec
            where = {}
            for k, p in enumerate(spec.pairs):
                where[(p.a, p.b)] = where[(p.b, p.a)] = k
            table = [[where.get((a._index, b._index), -1) for b in agents] for a in agents]
            self.pair_index = torch.tensor(table, dtype=torch.int32, device=self.dev).contiguous()
            self._buf.pair_index = self.pair_index.data_ptr()

    def persistent_tensors(self):
        return [self.pos_shaping]

    kind = A.POST_NAVIGATION  # as the epilogue of the physics kernel: LIDAR cast and collision reduction in the launch
    rollout_ok = True         # (instance attribute, see __init__: several steps per launch need a grid barrier per step)

    def prepare_rollout(self, n_steps: int):
        env, sc, K, n, B = self.env, self.env.scenario, int(n_steps), self.n, self.B
        self._bind_outputs()  # (re-binds the agents' shaping rows after a reset)
        out = {
            "obs": torch.empty(K, n, B, self.obs_dim, device=self.dev), "rew": torch.empty(K, n, B, device=self.dev),
            "done": torch.empty(K, B, device=self.dev, dtype=torch.bool),
            "agent_pos_rew": torch.empty(K, n, B, device=self.dev), "pos_rew": torch.empty(K, B, device=self.dev),
            "final_rew": torch.empty(K, B, device=self.dev), "agent_collisions": torch.empty(K, n, B, device=self.dev),
        }
        sc.pos_rew, sc.final_rew = out["pos_rew"][-1], out["final_rew"][-1]  # scenario / agent attributes: the last step's
        for i, a in enumerate(env.world.agents):
            a.pos_rew, a.agent_collision_rew = out["agent_pos_rew"][-1, i], out["agent_collisions"][-1, i]
        sc._lidar_cache = None
        b = A.NavigationBuffers()
        b.pos_shaping, b.limit = self.pos_shaping.data_ptr(), self._limit()
        b.limit.steps = env.steps.data_ptr()
        b.pair_index = self._buf.pair_index
        b.obs, b.rew, b.done = out["obs"].data_ptr(), out["rew"].data_ptr(), out["done"].data_ptr()
        b.agent_pos_rew, b.collision_rew = out["agent_pos_rew"].data_ptr(), out["agent_collisions"].data_ptr()
        b.pos_rew, b.final_rew = out["pos_rew"].data_ptr(), out["final_rew"].data_ptr()
        return self.desc, b, out

    def _bind_outputs(self):
        """Output tensors + the buffer struct's pointers (nothing launched)."""
        env, sc, w = self.env, self.env.scenario, self.env.world
        agents = w.agents
        for a, row, ptr in zip(agents, self._shaping_rows, self._shaping_ptrs):  # reset() may have rebound it
            if a.pos_shaping.data_ptr() != ptr:
                row.copy_(a.pos_shaping)
                a.pos_shaping = row
        fresh = self._out is None or not self.static_outputs
        obs, rew, done = self._outputs(self.obs_dim)
        if fresh or getattr(self, "_terms", None) is None:
            # one block: agent_pos_rew [n] | collision_rew [n] | pos_rew | final_rew
            self._terms = torch.empty(2 * self.n + 2, self.B, device=self.dev)
        t = self._terms
        rows = t.unbind(0)
        n = self.n
        sc.pos_rew, sc.final_rew = rows[2 * n], rows[2 * n + 1]
        for i, a in enumerate(agents):
            a.pos_rew, a.agent_collision_rew = rows[i], rows[n + i]
        b = self._buf
        b.obs, b.rew, b.done = obs.data_ptr(), rew.data_ptr(), done.data_ptr()
        p, row_bytes = t.data_ptr(), 4 * self.B
        b.agent_pos_rew, b.collision_rew = p, p + n * row_bytes
        b.pos_rew, b.final_rew = p + 2 * n * row_bytes, p + (2 * n + 1) * row_bytes
        b.limit.steps = env.steps.data_ptr()
        infos = [{"pos_rew": sc.pos_rew if sc.shared_rew else a.pos_rew, "final_rew": sc.final_rew,
                  "agent_collisions": a.agent_collision_rew} for a in env.agents]
        return list(obs.unbind(0)), list(rew.unbind(0)), done, infos

    def prepare(self):
        """(descriptor, buffers, what env.step returns) for the one-launch step (vmas_world_step_env)."""
        result = self._bind_outputs()
        self.env.scenario._lidar_cache = None  # (the sensors' measurements are made inside the launch, on the LDS tile)
        self._buf.lidar = self._buf.pair_any = None
        return self.desc, self._buf, result

    def __call__(self):
        env, sc, w = self.env, self.env.scenario, self.env.world
        result = self._bind_outputs()
        b = self._buf
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
        st = w._packed_state()
        _check(self.lib.vmas_navigation_post_step(C.byref(self.desc), C.byref(b), self.B, st.data_ptr(), st.shape[-1],
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
            if cur.data_ptr() != row.data_ptr():
                row.copy_(cur)
                setattr(ball, n, row)

    def prepare(self):
        """(descriptor, buffers, what env.step returns) - outputs allocated and bound to the scenario's attributes, nothing
        launched.  ``finish(result)`` completes the infos once the step has been enqueued."""
        env, sc, w = self.env, self.env.scenario, self.env.world
        ball = sc.ball
        self._rebind_shaping()
        obs, rew, done = self._outputs(self.obs_dim)
        if not self.static_outputs or getattr(self, "_terms", None) is None:
            self._terms = (torch.empty(len(self.TERMS), self.B, device=self.dev),
                           torch.empty(2, self.B, device=self.dev, dtype=torch.bool))
        terms, touching = self._terms
        b = self._buffers(A.FootballBuffers)
        b.pos_shaping = self.pos_shaping.data_ptr()
        b.obs, b.rew, b.done = obs.data_ptr(), rew.data_ptr(), done.data_ptr()
        b.terms, b.touching = terms.data_ptr(), touching.data_ptr()
        b.agent_ft = w._packed_agent_ft().data_ptr()
        t = dict(zip(self.TERMS, terms.unbind(0)))
        sc._sparse_reward_blue, sc._done = t["sparse_reward_blue"], done
        ball.pos_rew_blue, ball.pos_rew_red = t["pos_rew_blue"], t["pos_rew_red"]
        ball.pos_rew_agent_blue, ball.pos_rew_agent_red = t["pos_rew_agent_blue"], t["pos_rew_agent_red"]
        sc.min_agent_dist_to_ball_blue, sc.min_agent_dist_to_ball_red = (
            t["min_agent_dist_to_ball_blue"], t["min_agent_dist_to_ball_red"])
        infos = []
        for a in env.agents:
            side = "blue" if a in sc.blue_agents else "red"
            infos.append({
                "sparse_reward": t["sparse_reward_blue"],  # (red: negated in finish(), once the kernel has been enqueued)
                "ball_goal_pos_rew": t[f"pos_rew_{side}"], "all_agent_ball_pos_rew": t[f"pos_rew_agent_{side}"],
                "ball_pos": None, "dist_ball_to_goal": t[f"dist_ball_to_goal_{side}"],
                "min_agent_dist_to_ball": t[f"min_agent_dist_to_ball_{side}"],
                "touching_ball": touching[0 if side == "blue" else 1],
            })
        return self.desc, b, (list(obs.unbind(0)), list(rew.unbind(0)), done, infos)

    def finish(self, result):
        """The parts of the infos that are tensor ops on the step's outputs (stream-ordered behind the launch)."""
        env, sc = self.env, self.env.scenario
        ball = sc.ball
        ball_pos = ball.state.pos if self.static_outputs else ball.state.pos.clone()  # (not a live view of the state)
        sparse_red = None
        for a, info in zip(env.agents, result[3]):
            info["ball_pos"] = ball_pos
            if a not in sc.blue_agents:
                if sparse_red is None:
                    sparse_red = sc._sparse_reward_red = -sc._sparse_reward_blue
                info["sparse_reward"] = sparse_red
        return result

    def prepare_rollout(self, n_steps: int):
        env, sc, K, n, B = self.env, self.env.scenario, int(n_steps), self.n, self.B
        self._rebind_shaping()
        out = {
            "obs": torch.empty(K, n, B, self.obs_dim, device=self.dev), "rew": torch.empty(K, n, B, device=self.dev),
            "done": torch.empty(K, B, device=self.dev, dtype=torch.bool),
            "terms": torch.empty(K, len(self.TERMS), B, device=self.dev),
            "touching_ball": torch.empty(K, 2, B, device=self.dev, dtype=torch.bool),
        }
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
