"""The fused one-launch ``Environment.step`` for a LIVE reference environment: ``attach(env)`` rebinds not only
``World.step`` (adapter.AttachedWorld) but - for the four benchmark scenarios, when their configuration is one the
post-step kernels cover - ``env.step`` itself (vmas/simulator/environment/environment.py:325-405) to
``vmas_world_step_env``: ``_set_action`` + ``process_action`` + the dynamics (environment.py:616-749, scenario.py:92-98)
as the kernel's prologue, the scenario's ``reward`` / ``observation`` / ``done`` / ``info`` (balance.py:218-267,
transport.py:131-191, navigation.py:200-285, football.py:1121-1515) as its epilogue, returned in the reference's own
shapes and containers.  ``vmas.make_env`` / the ``Environment`` / the ``Scenario`` / the ``World`` stay the reference's
objects; the scenario's attributes its own methods maintain (shaping terms, ``pos_rew`` ...) keep being maintained, so
``reset`` / ``reset_at`` / ``get_from_scenario`` / a ``detach()`` work at any point.

The kernels and their host classes are the ones ``environment.Environment`` drives (fused.py): they read the
environment, its world and its scenario by attribute, so here they are handed VIEWS that serve those attributes from
the reference's objects and the ``AttachedWorld`` handle.  Nothing is re-implemented in tensor ops: a configuration no
kernel covers keeps the reference's own ``Environment.step`` (with the native ``World.step`` inside), and
``handle.fused_reason`` says why.
"""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import Tensor

from . import _abi as A


# ---------------------------------------------------------------------------------------------------------- views
class _WorldView:
    """What fused.py reads from a ``core.World``, served from an ``AttachedWorld`` handle (packed buffers, backend,
    spec) and, for everything else, from the reference's world."""

    def __init__(self, handle):
        object.__setattr__(self, "_h", handle)
        object.__setattr__(self, "_query_cache", None)

    def _packed_state(self):
        return self._h.state

    def _packed_agent_ft(self):
        return self._h.agent_ft

    def _get_backend(self):
        return self._h.backend

    def _per_env_inputs(self):
        return self._h._per_env_inputs()

    _backend = property(lambda self: self._h.backend)
    spec = property(lambda self: self._h.spec)
    exact_broad_phase = property(lambda self: self._h.exact_broad_phase)

    def __getattr__(self, name):
        return getattr(self._h.world, name)

    def __setattr__(self, name, value):
        if name == "_query_cache":  # (core.World's query cache: the reference world has none)
            object.__setattr__(self, name, value)
        else:
            setattr(self._h.world, name, value)


class _ScenarioView:
    """The reference's scenario under the attribute names fused.py uses: ``aliases`` first (entities the reference
    reaches another way - balance's goal is ``package.goal`` - and the hooks a reference scenario does not have), then
    the scenario itself.  Writes go to the scenario: ``pos_rew``, ``on_the_ground`` ... stay what its own methods and
    the user read."""

    _LOCAL = ("_lidar_cache",)

    def __init__(self, scenario, aliases: Dict[str, object]):
        object.__setattr__(self, "_sc", scenario)
        object.__setattr__(self, "_aliases", aliases)
        object.__setattr__(self, "_lidar_cache", None)

    def __getattr__(self, name):
        al = self._aliases
        if name in al:
            return al[name]
        return getattr(self._sc, name)

    def __setattr__(self, name, value):
        if name in self._LOCAL:
            object.__setattr__(self, name, value)
        else:
            setattr(self._sc, name, value)


class _EnvView:
    """What fused.py reads from an ``environment.Environment``."""

    def __init__(self, env, world_view, scenario_view, steps: Tensor, split_truncation: bool):
        self.env = env
        self.world = world_view
        self.scenario = scenario_view
        self.agents = env.agents
        self.n_agents = len(env.agents)
        self.num_envs = int(env.num_envs)
        self.device = torch.device(env.device)
        if self.device.index is None and self.device.type == "cuda":
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.continuous_actions = bool(env.continuous_actions)
        self.clamp_action = bool(env.clamp_action)
        self.steps = steps  # ONE tensor for the life of the attachment (the reference rebinds env.steps at reset)
        self._split = split_truncation
        self._lidar_cache = None

    @property
    def max_steps(self):
        # terminated_truncated=True: the kernel's done is the scenario's alone, the truncation is a tensor op beside it
        return None if self._split else self.env.max_steps

    def get_agent_action_size(self, agent):
        return self.env.get_agent_action_size(agent)


# ------------------------------------------------------------------------------------------------------- profiles
def _defined_in(cls, names, module: str) -> Optional[str]:
    """None if every named method of ``cls`` is the one ``module`` defines or the base class's hook (no subclass changed
    what the kernel restates), else the first that is not."""
    for n in names:
        m = getattr(getattr(cls, n, None), "__module__", None) or ""
        if m != module and not m.endswith("simulator.scenario"):
            return n
    return None


_SCENARIO_METHODS = ("reward", "observation", "done", "info", "process_action", "pre_step", "post_step")

# Scenario / action parameters the kernels' descriptors are built from (fused.*Post, fused.ActionIngest): the reference reads
# them at every step, so a write to one - ``scenario.shaping_factor = 50`` mid-run - must reach the kernel.  A ``__setattr__`` hook
# on the scenario's class (and the agents' Action class) marks the attached step dirty; the next ``env.step`` re-checks the
# configuration and rebuilds the descriptors, or hands ``env.step`` back to the reference if no kernel covers it any more.
_DIRTY = "_vmas_amd_fused_dirty"
_PARAMS = frozenset((
    "shaping_factor", "fall_reward",                                                                   # balance, transport
    "shared_rew", "collisions", "observe_all_goals", "pos_shaping_factor", "final_reward", "agent_collision_penalty",
    "min_collision_distance",                                                                          # navigation
    "observe_teammates", "observe_adversaries", "dense_reward", "pitch_length", "pitch_width", "ball_size", "goal_size",
    "agent_size", "pos_shaping_factor_ball_goal", "pos_shaping_factor_agent_ball", "distance_to_ball_trigger", "scoring_reward",
    "ai_red_agents", "ai_blue_agents", "enable_shooting", "physically_different", "dict_obs",          # football
    "_u_range", "_u_multiplier", "_u_noise",                                                           # Action
))


def _hook_params(cls) -> None:
    cur = cls.__setattr__
    if getattr(cur, "_vmas_amd_params", False):
        return

    def hook(self, name, value, _orig=cur):
        _orig(self, name, value)
        if name in _PARAMS:
            owner = self.__dict__.get(_DIRTY)
            if owner is not None:
                owner[0] = True

    hook._vmas_amd_params = True
    hook._vmas_amd = getattr(cur, "_vmas_amd", False)  # (adapter._patch_setattr's own mark, if its hook is underneath)
    cls.__setattr__ = hook


class _Profile:
    """One benchmark scenario of the reference: how to recognise it, which of its configurations the kernel covers, and
    its objects under the names fused.py's post-step class reads."""

    module_tail = ""   # the reference module: vmas.scenarios.<module_tail>
    post_kind = 0      # VMAS_POST_*: the epilogue the world's steps carry (kernel geometry, run-time specialisation)

    def check(self, env) -> Optional[str]:
        return None

    def aliases(self, env) -> Dict[str, object]:
        return {}

    def reserve(self, env):
        """(post_kind, n_packages) for ``vmas_world_reserve_epilogue``, or None."""
        return None

    def make_post(self, view):
        raise NotImplementedError

    def reset_program(self, scenario_view):
        """The scenario's ``reset_world_at`` as a spawn program for ``vmas_env_reset_where`` (fused.MaskedReset) - stated ONCE,
        by this package's scenario of the same name (``scenarios/<name>.py::fused_reset_program``, whose law is pinned to the
        reference's in tests/test_reset_law_vs_reference.py), and evaluated here on the reference's objects through the view.
        None: this configuration's reset is not a spawn program (shared goals, formation spawning)."""
        import importlib

        mod = importlib.import_module(f"{__package__}.scenarios.{self.module_tail}")
        return mod.Scenario.fused_reset_program(scenario_view)


class _Balance(_Profile):
    module_tail, post_kind = "balance", A.POST_BALANCE

    def aliases(self, env):
        return {"goal": env.scenario.package.goal}

    def reserve(self, env):
        return (A.POST_BALANCE, 0)

    def make_post(self, view):
        from .fused import BalancePost
        return BalancePost(view)


class _Transport(_Profile):
    module_tail, post_kind = "transport", A.POST_TRANSPORT

    def check(self, env):
        if len(env.scenario.packages) > A.ENV_MAX_PACKAGES:
            return "too many packages"
        return None

    def aliases(self, env):
        return {"goal": env.scenario.packages[0].goal}

    def reserve(self, env):
        return (A.POST_TRANSPORT, len(env.scenario.packages))

    def make_post(self, view):
        from .fused import TransportPost
        return TransportPost(view)


class _Navigation(_Profile):
    module_tail, post_kind = "navigation", A.POST_NAVIGATION

    def check(self, env):
        from .fused import NavigationPost
        sc = env.scenario
        if sc.collisions and any(len(a.sensors) != 1 or not hasattr(a.sensors[0], "_angles") for a in env.world.agents):
            return "agents without exactly one Lidar"
        return NavigationPost.supports(env)

    def make_post(self, view):
        from .fused import NavigationPost
        return NavigationPost(view)


class _Football(_Profile):
    module_tail, post_kind = "football", 0  # (its worlds run the lane-compacted kernel: no run-time specialisation)

    def check(self, env):
        sc = env.scenario
        for flag, what in ((sc.ai_red_agents, "ai_red_agents=True (the heuristic AgentPolicy opponents)"),
                           (sc.ai_blue_agents, "ai_blue_agents=True"), (sc.enable_shooting, "enable_shooting=True"),
                           (sc.physically_different, "physically_different=True"), (getattr(sc, "dict_obs", False), "dict_obs=True"),
                           (not sc.dense_reward, "dense_reward=False")):
            if flag:
                return what
        n = len(sc.blue_agents) + len(sc.red_agents)
        same_dims = (sc.n_blue_agents == sc.n_red_agents or not (sc.observe_teammates or sc.observe_adversaries)
                     or (sc.observe_teammates and sc.observe_adversaries))
        if n + 1 > A.ENV_MAX_AGENTS or not same_dims:
            return "team sizes whose observations differ in length"
        script = getattr(sc.ball, "action_script", None) or getattr(sc.ball, "_action_script", None)
        if getattr(script, "__name__", "") != "ball_action_script" or getattr(script, "__module__", "") != type(sc).__module__:
            return "the ball's action script is not the scenario's ball_action_script"
        return None

    def aliases(self, env):
        sc = env.scenario
        red = list(sc.red_agents)

        def fused_action_factors(agent):  # football.py:1050-1057: red agents act in a mirrored frame
            return [-1.0, 1.0] if agent in red else None

        def fused_agent_scripts():  # football.py:1620-1680 on the device (VMAS_SCRIPT_FOOTBALL_BALL)
            return [dict(kind=A.SCRIPT_FOOTBALL_BALL, agent=sc.ball,
                         params=[sc.agent_size * 2, sc.pitch_width / 2, sc.pitch_length / 2, sc.goal_size / 2])]

        # the walls, goal lines and nets: where the scenario's reset_walls / reset_goals (football.py:686-1020) put them - read
        # from environment 0 as it stands after the reset make_env made (the same pose in every environment, every episode)
        static = list(env.world.landmarks)
        poses = [(lm.name, None, None, tuple(float(x) for x in lm.state.pos[0].tolist()), float(lm.state.rot[0, 0])) for lm in static]
        return {"fused_action_factors": fused_action_factors, "fused_agent_scripts": fused_agent_scripts,
                "_static_landmarks": static, "_static": poses}

    def make_post(self, view):
        from .fused import FootballPost
        return FootballPost(view)


_PROFILES = (_Balance(), _Transport(), _Navigation(), _Football())


def find_profile(env):
    """(profile, None) if ``env.scenario`` is one of the reference's four benchmark scenarios as shipped, else
    (None, reason)."""
    sc = getattr(env, "scenario", None)
    if sc is None:
        return None, "not an Environment"
    mod = type(sc).__module__
    for p in _PROFILES:
        if mod == "vmas.scenarios." + p.module_tail:
            odd = _defined_in(type(sc), _SCENARIO_METHODS, mod)
            if odd is not None:
                return None, f"scenario.{odd} is not the reference's"
            if any(n in sc.__dict__ for n in _SCENARIO_METHODS):
                return None, "a scenario method is overridden on the instance"
            return p, None
    return None, f"no fused post-step kernel for scenario {mod}"


def _ingest_reason(env, known_scripted) -> Optional[str]:
    """None if ``Environment._set_action`` + ``env_process_action`` of this environment is what the ingest kernel does
    (environment.py:616-749 continuous / flat-discrete branch, Holonomic dynamics), else the reason it is not."""
    w = env.world
    if len(env.agents) > A.ENV_MAX_AGENTS:
        return "too many agents"
    if getattr(env, "multidiscrete_actions", False):
        return "multidiscrete actions"
    if any(id(a) not in known_scripted for a in w.scripted_agents) or len(known_scripted) > A.ENV_MAX_SCRIPTS:
        return "scripted agents"
    for a in env.agents:
        dyn = type(a.dynamics)
        if (dyn.__name__, dyn.__module__.rsplit(".", 1)[-1]) not in (("Holonomic", "holonomic"), ("HolonomicWithRotation", "holonomic_with_rot")):
            return f"dynamics {dyn.__name__}"
        if a.action_size != a.dynamics.needed_action_size:
            return "action_size != dynamics.needed_action_size"
        if a.action.u_noise not in (0, 0.0, None):
            return "action noise"
        if not a.silent and w.dim_c > 0:
            return "communication actions"
    return None


# ------------------------------------------------------------------------------------------------------ the step
class FusedEnvStep:
    """``env.step`` of an attached reference environment as one launch.  Built by ``adapter.attach``; rebuilt by the
    handle whenever the native world is (``AttachedWorld.refresh``)."""

    def __init__(self, env, handle, profile: _Profile, validate_actions: bool = True):
        self.env, self.handle, self.profile = env, handle, profile
        # True: the reference's asserts before the world is touched (one stream synchronisation per step); "deferred": the
        # step launch itself flags a bad action and the NEXT env.step (or check_actions()) raises - no synchronisation;
        # False: no check
        assert validate_actions in (True, False, "deferred"), validate_actions
        self.validate_actions = validate_actions is True
        self.deferred = validate_actions == "deferred"
        self.dict_spaces = bool(getattr(env, "dict_spaces", False))
        self.split = bool(getattr(env, "terminated_truncated", False))
        self.names = [a.name for a in env.agents]
        for i, e in enumerate(env.world.entities):  # the indices fused.py addresses entities / agent rows by
            e.__dict__["_index"] = i
        for i, a in enumerate(env.world.agents):
            a.__dict__["_agent_index"] = i
        self.steps = torch.zeros(env.num_envs, device=env.device, dtype=torch.float32)
        self.reset_seed = 0  # reset_where's counter-based generator: keyed by (this, environment, that environment's episode)
        self._masked_reset = None
        self._orig_step = env.__dict__.get("step")
        self._dirty = [False]  # (a one-element list: the hooked objects hold a reference to it)
        self._hooked = [env.scenario] + [a.action for a in env.world.agents]
        for o in self._hooked:
            _hook_params(type(o))
            o.__dict__[_DIRTY] = self._dirty
        try:
            self.build()
        except BaseException:  # (attach() falls back to the reference's step: leave none of this layer's marks behind)
            for o in self._hooked:
                o.__dict__.pop(_DIRTY, None)
            raise
        self._adopt_steps()
        env.step = self.step

    # ---- construction -------------------------------------------------------------------------------------------
    def build(self):
        """(Re)create the host objects of the kernels on the handle's CURRENT backend."""
        from . import fused as F

        env, h = self.env, self.handle
        wv = _WorldView(h)
        sv = _ScenarioView(env.scenario, self.profile.aliases(env))
        self.view = view = _EnvView(env, wv, sv, self.steps, self.split)
        self.ingest = F.ActionIngest(view)
        self.post = self.profile.make_post(view)
        exact_in_launch = (not h.exact_broad_phase) or h.backend.exact_form() != 3  # (vmas_world_exact_form: lazy, in the launch)
        # (environment.Environment._setup_fused: the same three forms - the whole step as one launch; ingest + physics as one
        #  launch and the post-step as another; three launches where the exact broad phase cannot run inside the step's)
        self.ingest_in_step = exact_in_launch and env.world.dim_c == 0
        self.one_launch = self.ingest_in_step and self.post.kind is not None
        self.launch = F.StepLauncher(view, self.ingest) if self.ingest_in_step else None
        self._masked_reset = None
        self._finish = getattr(self.post, "finish", None)
        self._max_steps = env.max_steps
        self._backend = h.backend

    def _params_changed(self) -> bool:
        """Re-plan after a parameter write: True = the fused step goes on with rebuilt descriptors, False = it was taken out
        (``handle.fused`` is None, ``handle.fused_reason`` says why, ``env.step`` is the reference's again)."""
        self._dirty[0] = False
        env = self.env
        reason = self.profile.check(env)
        if reason is None:
            al = self.profile.aliases(env)
            scripts = al["fused_agent_scripts"]() if "fused_agent_scripts" in al else []
            reason = _ingest_reason(env, {id(s_["agent"]) for s_ in scripts})
        if reason is None:
            self.build()
            return True
        self.handle.fused, self.handle.fused_reason = None, f"a parameter changed after attach(): {reason}"
        self.detach()
        return False

    def _adopt_steps(self):
        """``Environment._reset`` rebinds ``env.steps`` (environment.py:222): the kernel's counter is ONE tensor - take the
        new values over and put that tensor back."""
        cur = self.env.steps
        if cur is not self.steps:
            self.steps.copy_(cur)
            self.env.steps = self.steps

    def _replan(self) -> bool:
        """What every entry (``step``, ``rollout``, ``reset_where``) does first: follow static changes of the world, re-plan
        after a parameter write / a rebuilt backend / a changed time limit, adopt a rebound ``env.steps``.  False = the fused
        step was taken out (``_params_changed``)."""
        env, h = self.env, self.handle
        h._sync_static()  # (masses, filters ... written since the last step: the native world follows - adapter.py)
        if self._dirty[0]:  # a scenario / action parameter the descriptors were built from was written
            if not self._params_changed():
                return False
        elif h.backend is not self._backend or env.max_steps != self._max_steps:  # (the time limit is part of the output sets)
            self.build()
        if env.steps is not self.steps:
            self._adopt_steps()
        return True

    # ---- the replaced method ------------------------------------------------------------------------------------
    def step(self, actions):
        """Environment.step (environment.py:325-405): same arguments, same return value."""
        env = self.env
        if isinstance(actions, dict):
            try:
                ordered = [actions[n] for n in self.names]
            except KeyError as e:
                raise AssertionError(f"Agent '{e.args[0]}' not contained in action dict")
            assert len(actions) == len(self.names), f"Expecting actions for {len(self.names)}, got {len(actions)} actions"
            actions = ordered
        assert len(actions) == len(self.names), f"Expecting actions for {len(self.names)}, got {len(actions)} actions"
        h = self.handle
        if not self._replan():
            return env.step(actions)  # (no kernel covers the configuration any more: the reference's own step, restored)
        ingest, post = self.ingest, self.post
        deferred = self.deferred
        if deferred:
            ingest.pending()  # what an earlier step's launch found
        if self.one_launch:
            ingest.prepare(actions)
            if self.validate_actions and self.launch.can_gate(post.kind):
                # the reference's asserts without an idle queue: the check is enqueued, the step launched GATED on its result
                # (it does nothing if an action was refused), and only then does the host wait for the check - while the step
                # is already running behind it
                seq = ingest.validate_begin()
                saved = post.save_bound()
                desc, buffers, result = post.prepare()
                self.launch.gated(post.kind, desc, buffers)
                flags = ingest.validate_end(seq)
                if flags:
                    self.launch.refused()
                    post.restore_bound(saved)  # (the scenario keeps the previous step's pos_rew ...: it saw nothing of this one)
                    ingest._raise(flags)
            else:
                if self.validate_actions:
                    ingest.validate()  # raises before the world is touched, like the reference's asserts
                desc, buffers, result = post.prepare()
                self.launch(post.kind, desc, buffers, deferred)
            if self._finish is not None:
                result = self._finish(result)
        else:
            if self.ingest_in_step:
                ingest.prepare(actions)
                if self.validate_actions:
                    ingest.validate()
                self.launch(0, None, None, deferred)
            else:
                # (three launches: no launch here carries the deferred error word - "deferred" checks up front like True)
                ingest(actions, self.validate_actions or deferred)
                h.step()
            result = post()
        obs, rews, dones, infos = result
        if self.dict_spaces:
            names = self.names
            obs, rews, infos = dict(zip(names, obs)), dict(zip(names, rews)), dict(zip(names, infos))
        if self.split:
            ms = env.max_steps
            truncated = (self.steps >= ms) if ms is not None else torch.zeros_like(dones)
            return [obs, rews, dones, truncated, infos]
        return [obs, rews, dones, infos]

    def rollout_fields(self, n_steps: int):
        """(name, shape, dtype) of every per-step output ``rollout()`` writes for ``n_steps`` steps - what a caller-provided
        ``out`` must hold (``shard.NativeRollout(shard, handle.fused.rollout_fields(K), device)`` lays them out in the ONE
        buffer that the end-of-rollout ``all_gather_into_tensor`` sends as it is: SURVEY.md 8e on the reference's objects)."""
        return self.post.rollout_fields(n_steps)

    def rollout(self, actions, out=None):
        """K consecutive ``env.step`` calls with given actions in ONE kernel launch (``vmas_world_rollout_env``)
        on the reference's environment: ``actions[i]`` = agent i's ``[K, num_envs, action_size]``; returns ``{"obs": [K, n_agents,
        num_envs, obs_dim], "rew": [K, n_agents, num_envs], "done": [K, num_envs], ...info terms}`` - entry k is what the k-th
        ``env.step`` would have returned, bit for bit; no resets in between.  The scenario's attributes are the last step's.
        Validation (``validate_actions`` not False): the reference's asserts on the whole action tensors up front."""
        env, h = self.env, self.handle
        assert self.one_launch and getattr(self.post, "rollout_ok", True), (
            "rollout() needs a configuration whose env.step is one launch without a per-step reduction over more tiles than CUs")
        assert len(actions) == len(self.names), f"Expecting actions for {len(self.names)}, got {len(actions)} actions"
        if not self._replan():
            raise NotImplementedError(f"rollout(): {h.fused_reason}")
        assert self.one_launch, "rollout(): the re-planned configuration's env.step is no longer one launch"
        K = int(actions[0].shape[0])
        self.ingest.prepare_rollout(list(actions), K, self.validate_actions or self.deferred)
        desc, buffers, out = self.post.prepare_rollout(K, out)
        self.launch.rollout(self.post.kind, desc, buffers, K)
        return out

    def reset_where(self, mask, return_observations: bool = False):
        """``env.reset_at(i)`` for every environment where ``mask`` [num_envs] is set - ONE launch, no host sync
        (``vmas_env_reset_where``).  The reference resets one environment per Python call (each a few dozen
        small launches on a GPU): a rollout over 32 768 environments that resets hundreds per step cannot afford it.  The
        placement law is the scenario's ``reset_world_at`` restated as a spawn program on a counter-based generator keyed by
        (``reset_seed``, environment, that environment's episode count); unmasked environments keep their bits, the scenario's
        cached terms and flags of the reset environments are re-initialised in place, ``env.steps`` zeroed there.
        ``return_observations``: by the reference's own ``get_from_scenario`` on the new state."""
        from . import fused as F

        env = self.env
        if not self._replan():
            raise NotImplementedError(f"reset_where(): {self.handle.fused_reason}")
        if self._masked_reset is not None and self._masked_reset_seed != int(self.reset_seed):
            self._masked_reset = None  # (``handle.fused.reset_seed = n`` after the first use: a new generator key)
        if self._masked_reset is None:
            prog = self.profile.reset_program(self.view.scenario)
            if prog is None or len(prog["ops"]) > A.RESET_MAX_OPS or len(prog.get("terms", [])) > A.RESET_MAX_TERMS:
                raise NotImplementedError("reset_where(): this configuration's reset is not a spawn program the kernel runs "
                                          "(shared goals / formation spawning / too many entities): use env.reset_at(i)")
            self._masked_reset = F.MaskedReset(self.view, prog, int(self.reset_seed))
            self._masked_reset_seed = int(self.reset_seed)
        mask = mask.to(self.view.device).reshape(self.view.num_envs).bool().contiguous()
        self._masked_reset(mask)
        if return_observations:
            return env.get_from_scenario(get_observations=True, get_rewards=False, get_infos=False, get_dones=False)[0]
        return None

    def check_actions(self):
        """Deferred validation: wait for the steps enqueued so far and raise if one of them was given a bad action."""
        torch.cuda.current_stream(self.view.device).synchronize()
        self.ingest.pending()

    def detach(self):
        env = self.env
        if self._orig_step is not None:
            env.step = self._orig_step
        else:
            env.__dict__.pop("step", None)
        env.steps = self.steps.clone()
        for o in self._hooked:  # (the class hooks stay: they do nothing for objects without an owner)
            o.__dict__.pop(_DIRTY, None)


def plan_fuse(env):
    """(profile, None) if ``env.step`` can be the one-launch kernel, else (None, reason).  Looks only at the environment:
    ``adapter.attach`` builds the native world for the profile's epilogue afterwards."""
    if not hasattr(env, "scenario") or not hasattr(env, "step") or not hasattr(env, "agents"):
        return None, "attach() was given a World, not an Environment"
    if torch.device(env.device).type != "cuda":
        return None, "not a GPU environment"
    profile, reason = find_profile(env)
    if profile is None:
        return None, reason
    reason = profile.check(env)
    if reason is None:
        al = profile.aliases(env)
        scripts = al["fused_agent_scripts"]() if "fused_agent_scripts" in al else []
        reason = _ingest_reason(env, {id(s["agent"]) for s in scripts})
    return (profile, None) if reason is None else (None, reason)
