"""Batched environment on top of the native world: ``make_env`` / ``Environment.step`` /
``reset`` / ``reset_at`` with the reference's call order and return conventions
(vmas/make_env.py:14-101, vmas/simulator/environment/environment.py:65-429, 616-749).

Differences that are deliberate:
* one device-side RNG (``torch.Generator`` on the env device) instead of the process-global RNG
  swapping of ``local_seed`` (environment.py:31-47) - no hidden coupling between environments;
* action validation (NaN / range asserts, environment.py:621,651-653) costs two host syncs per
  agent per step in the reference; here it is one small kernel + one sync in FRONT of the step launch
  (same error behaviour: a bad action raises before the world is touched), on by default, and can be
  switched off with ``validate_actions=False`` for throughput runs;
* ``World.step()`` is one kernel launch (core.World), Lidar sensors are cast by one launch for
  all agents right after the step and cached for ``observation()``.
"""
from __future__ import annotations

import importlib
import math
from typing import Callable, Dict, List, Optional, Union

import torch
from torch import Tensor

from ._abi import RESET_MAX_OPS as A_RESET_MAX_OPS, RESET_MAX_TERMS as A_RESET_MAX_TERMS
from .core import Agent, World
from .scenario import BaseScenario


class Environment:
    def __init__(
        self,
        scenario: BaseScenario,
        num_envs: int = 32,
        device: Union[torch.device, str] = "cuda:0",
        max_steps: Optional[int] = None,
        continuous_actions: bool = True,
        seed: Optional[int] = None,
        clamp_actions: bool = False,
        validate_actions: bool = True,
        graph: bool = False,
        fused: Optional[bool] = None,
        specialize: Optional[bool] = None,
        **kwargs,
    ):
        self.scenario = scenario
        self.use_graph = graph
        self.fused = fused
        self._graph = None
        self.num_envs = num_envs
        self.device = torch.device(device)
        self.max_steps = max_steps
        self.continuous_actions = continuous_actions
        self.clamp_action = clamp_actions
        self.validate_actions = validate_actions
        self.world: World = scenario.env_make_world(num_envs, self.device, **kwargs)
        self.agents: List[Agent] = self.world.policy_agents
        self.n_agents = len(self.agents)
        if self.world.dim_c > 0 and any(not a.silent for a in self.agents):
            # environment.py:718-749 (communication part of the action, one-hot mapping, c_noise) is not on the
            # World.step path and is not mirrored here: refuse loudly instead of dropping state.c after the first step
            raise NotImplementedError("communication actions (world.dim_c > 0 with non-silent policy agents) are not "
                                      "supported by this Environment; attach() the reference's environment instead")
        self.steps = torch.zeros(num_envs, device=self.device)
        self._lidar_cache: Optional[Tensor] = None
        self.seed(seed)
        self.reset(return_observations=False)
        self._ingest = self._post = self._masked_reset = None
        self._one_launch = self._ingest_in_step = False
        self._bound_actions = self._bound = None
        self._setup_fused()
        if specialize is not False and self.device.type == "cuda":
            # a step kernel compiled for THIS world (specialize.py).  True: compile it if the on-disk cache does not have
            # it (tens of seconds, once); None (default): take it from the cache if it is there (milliseconds) - worlds
            # someone has specialised before, or __graft_entry__.build() pre-compiled - and never compile; False: never
            self.world.specialize(cached_only=specialize is None)

    def _setup_fused(self):
        """``fused``: run action ingest and the scenario's reward/observation/done/info as one HIP
        kernel each (fused.py, include/vmas_env_hip.h) instead of per-agent tensor ops.  ``None`` =
        wherever the scenario and the agents allow it on a GPU device; ``True`` = required."""
        if self.fused is False or self.device.type != "cuda":
            assert not self.fused, "fused=True needs a GPU device"
            return
        from . import fused as F
        why_not = F.ActionIngest.supports(self)
        if why_not is None:
            self._ingest = F.ActionIngest(self)
        make_post = getattr(self.scenario, "make_fused_post", None)
        if make_post is not None:
            self._post = make_post(self)  # None when this configuration is not covered by a kernel
        if self.fused:
            assert self._ingest is not None, f"fused=True: action path cannot be fused ({why_not})"
            assert self._post is not None, "fused=True: the scenario has no fused post-step kernel"
        if self._post is not None:
            self._post.static_outputs = self.use_graph
        # the whole step as ONE launch (vmas_world_step_env): ingest = prologue, post-step = epilogue of
        # the physics kernel.  Needs the hooks between the stages to be the base class no-ops.
        w = self.world
        # (the exact broad phase runs inside the step launch - the lazy form, at any size - unless the library says it takes a
        #  launch per substep for this world: vmas_world_exact_form)
        exact_in_launch = not w.exact_broad_phase or w._get_backend().exact_form() != 3
        # (graph=True captures the generic step - ingest prologue + physics - when the scenario has no fused post-step; under
        # capture the exact broad phase runs one launch per substep (a replay would repeat a grid-barrier number), which
        # cannot carry the prologue: the stand-alone ingest kernel + World.step are captured instead)
        captured_exact = self.use_graph and w.exact_broad_phase and self._post is None
        self._ingest_in_step = (  # action ingest as the physics kernel's prologue
            self._ingest is not None and type(self.scenario).pre_step is BaseScenario.pre_step
            and w.dim_c == 0 and exact_in_launch and not captured_exact
        )
        self._one_launch = (
            self._ingest_in_step and self._post is not None and self._post.kind is not None
            and type(self.scenario).post_step is BaseScenario.post_step
        )
        if self._ingest_in_step:
            self._launch = F.StepLauncher(self, self._ingest)
        # reset_where as one kernel over the masked environments (scenarios that state their reset as a spawn program)
        prog = getattr(self.scenario, "fused_reset_program", None)
        prog = prog() if prog is not None and type(self.scenario).env_reset_world_at is BaseScenario.env_reset_world_at else None
        if prog is not None and len(prog["ops"]) <= A_RESET_MAX_OPS and len(prog.get("terms", [])) <= A_RESET_MAX_TERMS:
            self._masked_reset = F.MaskedReset(self, prog, self._seed)

    batch_dim = property(lambda self: self.num_envs)

    # ------------------------------------------------------------------ seeding / reset
    def seed(self, seed: Optional[int] = None):
        if seed is None:
            seed = 0
        self._seed = int(seed)
        if getattr(self, "_masked_reset", None) is not None:
            # the masked resets are keyed by (seed, environment, episode): a re-seeded environment starts its episodes
            # over, so that reset(seed=s) reproduces what a fresh environment built with seed s draws
            self._masked_reset.args.seed = self._seed & 0xFFFFFFFFFFFFFFFF
            self._masked_reset.episode.zero_()
        torch.manual_seed(seed)
        if self.device.type == "cuda":
            torch.cuda.manual_seed(seed)
        return [seed]

    def reset(self, seed: Optional[int] = None, return_observations: bool = True):
        if seed is not None:
            self.seed(seed)
        self.scenario.env_reset_world_at(env_index=None)
        self.steps.zero_()  # in place: a captured step graph keeps its pointer
        self._lidar_cache = None
        self._bound = None
        return self._observations() if return_observations else None

    def reset_at(self, index: int, return_observations: bool = True):
        assert 0 <= index < self.num_envs, f"Index must be between 0 and {self.num_envs}, got {index}"
        self.scenario.env_reset_world_at(index)
        self.steps[index] = 0
        self._lidar_cache = None
        self._bound = None
        return [o[index] for o in self._observations()] if return_observations else None

    def reset_where(self, mask: Tensor, return_observations: bool = True):
        """Reset exactly the environments where ``mask`` [num_envs] is True - all of them at once, on the
        device, without a host sync.  (The reference only has ``reset_at(i)``, one Python call per
        environment - SURVEY.md section 8f-4; a rollout over 32 768 environments resets hundreds per step.)

        Scenarios that state their reset as a spawn program (``fused_reset_program``: balance, transport, navigation)
        run ONE kernel that draws only the masked environments - ``World.reset`` + the placement law of
        ``reset_world_at`` with the reference's minimum-distance rejection (utils.py:241-319) on a counter-based
        generator keyed by (seed, environment, that environment's episode number) + the scenario's cached terms -
        and touches nothing else.  Otherwise a fresh initial state is drawn for every environment by the scenario's
        own vectorised reset and blended in where the mask is set (packed state, agent forces, step counter, the
        scenario's in-place tensors).  Either way unmasked environments keep their bits."""
        mask = mask.to(self.device).reshape(self.num_envs).bool().contiguous()
        self._bound = None
        if self._masked_reset is not None:
            # ONE launch over the masked environments (vmas_env_reset_where): World.reset + the scenario's spawn program on
            # a counter-based generator + its cached terms; unmasked environments are not touched
            self._masked_reset(mask)
            self.world.invalidate_queries()
            self._lidar_cache = None
            return self._observations() if return_observations else None
        persistent = self._persistent_tensors()
        saved = [t.clone() for t in persistent]
        self.scenario.env_reset_world_at(env_index=None)
        self.steps.zero_()
        after = self._persistent_tensors()
        assert len(after) == len(saved) and all(a.data_ptr() == p.data_ptr() for a, p in zip(after, persistent)), (
            "a full reset must update the scenario's persistent tensors in place (scenario.keep)")
        B = self.num_envs
        ld = persistent[0].shape[-1]
        mask_ld = torch.zeros(ld, dtype=torch.bool, device=self.device)
        mask_ld[:B] = mask
        for k, (t, old) in enumerate(zip(persistent, saved)):
            if k < 2:  # the packed state / agent forces [.., .., ld]: environment = last axis
                m = mask_ld
            elif t.shape[0] == B:
                m = mask.reshape((B,) + (1,) * (t.dim() - 1))
            else:
                assert t.shape[-1] == B, f"cannot locate the environment axis of a tensor of shape {tuple(t.shape)}"
                m = mask
            t.copy_(torch.where(m, t, old))
        self.world.invalidate_queries()
        self._lidar_cache = None
        return self._observations() if return_observations else None

    # ------------------------------------------------------------------ actions
    def get_agent_action_size(self, agent: Agent) -> int:
        return agent.action_size if self.continuous_actions else 1

    def get_random_action(self, agent: Agent) -> Tensor:
        """environment.py:536-548"""
        if self.continuous_actions:
            u = agent.action.u_range_tensor_on(self.device)
            return (torch.rand(self.num_envs, agent.action_size, device=self.device) * 2 - 1) * u
        n = math.prod(agent.discrete_action_nvec)
        return torch.randint(0, n, (self.num_envs, 1), device=self.device)

    def _set_action(self, action: Tensor, agent: Agent):
        action = action.detach().to(self.device)
        if self.validate_actions:
            assert not action.isnan().any()
        assert action.shape[1] == self.get_agent_action_size(agent), (
            f"Agent {agent.name} has wrong action size, got {action.shape[1]}, "
            f"expected {self.get_agent_action_size(agent)}"
        )
        u_range = agent.action.u_range_tensor_on(self.device)
        if self.continuous_actions:
            u = action[:, : agent.action_size].to(torch.float32)
            if self.clamp_action:
                u = torch.maximum(torch.minimum(u, u_range), -u_range)
            elif self.validate_actions:
                assert not torch.any(
                    torch.abs(u) > u_range
                ), f"Physical actions of agent {agent.name} are out of its range {agent.u_range}"
        else:  # flat index -> per-dimension index -> [-u_max, u_max] (environment.py:657-705)
            flat = action.squeeze(-1).to(torch.int64)
            nvec = list(agent.discrete_action_nvec)
            if self.validate_actions:
                assert torch.all((flat >= 0) & (flat < math.prod(nvec))), f"Discrete action of {agent.name} out of range"
            cols = []
            for i, n in enumerate(nvec):
                m = math.prod(nvec[i + 1 :])
                a = flat // m
                flat = flat % m
                if n % 2 != 0:  # first action maps to u = 0
                    stay = a == 0
                    dec = (a > 0) & (a <= n // 2)
                    a = torch.where(stay, torch.full_like(a, n // 2), torch.where(dec, a - 1, a))
                cols.append((a.to(torch.float32) / (n - 1)) * (2 * u_range[i]) - u_range[i])
            u = torch.stack(cols, dim=-1)
        u = u * agent.action.u_multiplier_tensor_on(self.device)
        if isinstance(agent.action.u_noise, (int, float)) and agent.action.u_noise > 0:
            u = u + torch.randn_like(u) * agent.action.u_noise
        agent.action.u = u

    # ------------------------------------------------------------------ step
    def step(self, actions: Union[List[Tensor], Dict[str, Tensor]]):
        """environment.py:325-405: returns (obs, rews, dones, infos) as per-agent lists.

        With ``graph=True`` the whole step - action ingest, the physics kernel, the LIDAR kernel and
        the scenario's observation/reward/done tensor ops - is captured once into a HIP graph and
        replayed: one launch per step instead of hundreds of small ones (the scenario code must be
        free of host syncs, which the shipped scenarios are).  The returned tensors are then static
        buffers that the next step overwrites."""
        if isinstance(actions, dict):
            actions = [actions[a.name] for a in self.agents]
        if self.use_graph:
            return self._step_graphed(actions)
        return self._step_eager(actions)

    def _step_graphed(self, actions):
        if self._one_launch:  # the whole step already is a single kernel: nothing to capture
            return self._step_eager(actions)
        if self._ingest_in_step and self._post is not None:
            return self._step_graphed_post(actions)
        if self._graph is None:
            assert self.device.type == "cuda", "graph=True needs a GPU device"
            assert not self.validate_actions, "graph=True needs validate_actions=False (the asserts are host syncs)"
            self._static_actions = [torch.zeros(self.num_envs, self.get_agent_action_size(a), device=self.device,
                                                dtype=torch.float32 if self.continuous_actions else torch.int64)
                                    for a in self.agents]
            for s_, a in zip(self._static_actions, actions):
                s_.copy_(a.reshape(s_.shape))
            persistent = self._persistent_tensors()
            saved = [t.clone() for t in persistent]
            side = torch.cuda.Stream(self.device)
            side.wait_stream(torch.cuda.current_stream(self.device))
            with torch.cuda.stream(side):  # warm-up on a side stream (allocations, lazy inits) ...
                for _ in range(3):
                    self._step_eager(self._static_actions)
            torch.cuda.current_stream(self.device).wait_stream(side)
            for t, s_ in zip(persistent, saved):  # ... undone: this call makes exactly one step, like any other
                t.copy_(s_)
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self._static_out = self._step_eager(self._static_actions)
        else:
            for s_, a in zip(self._static_actions, actions):
                s_.copy_(a.reshape(s_.shape))
        self._graph.replay()
        return self._static_out

    def get_state(self) -> List[Tensor]:
        """Snapshot of everything a step reads and writes (checkpoint / resume; the reference has no counterpart - its state
        is spread over per-entity attributes): the packed world state, the agent forces, the step counter, the scenario's
        persistent terms and the masked-reset episode counters.  ``set_state`` of the snapshot followed by the same
        actions reproduces the same steps bit for bit."""
        ts = [t.clone() for t in self._persistent_tensors()]
        if self._masked_reset is not None:
            ts.append(self._masked_reset.episode.clone())
        return ts

    def set_state(self, snapshot: List[Tensor]) -> None:
        cur = self._persistent_tensors()
        if self._masked_reset is not None:
            cur = cur + [self._masked_reset.episode]
        assert len(cur) == len(snapshot), f"snapshot has {len(snapshot)} tensors, this environment {len(cur)}"
        for t, s_ in zip(cur, snapshot):
            assert t.shape == s_.shape and t.dtype == s_.dtype, (tuple(t.shape), tuple(s_.shape))
            t.copy_(s_)
        self.world.invalidate_queries()
        self._lidar_cache = None
        self._bound = None

    def _persistent_tensors(self):
        """Everything a step reads and writes: the packed world state, the step counter and the scenario's
        in-place tensors (``scenario.keep``)."""
        ts = [self.world._packed_state(), self.world._packed_agent_ft(), self.steps]
        for obj in [self.scenario] + list(self.world.entities):
            ts += [getattr(obj, n) for n in sorted(getattr(obj, "_kept_names", ()))]
        return ts

    def _step_graphed_post(self, actions):
        """Prologue + physics as one eager launch (it reads the caller's action tensors directly - no
        copies into static buffers), the launches after it (LIDAR, collision mask, post-step kernel) as one
        HIP-graph replay."""
        assert len(actions) == self.n_agents, f"Expecting actions for {self.n_agents}, got {len(actions)} actions"
        self._ingest.prepare(actions)
        if self.validate_actions:
            self._ingest.validate()  # raises before the world is touched, like the reference
        self._launch(0, None, None, False)
        self.scenario.post_step()
        if self._graph is None:
            dev = self.device
            persistent = self._post.persistent_tensors() + [self.steps]
            saved = [t.clone() for t in persistent]
            side = torch.cuda.Stream(dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            with torch.cuda.stream(side):  # warm-up on a side stream (allocations, lazy inits) ...
                self._post()
            torch.cuda.current_stream(dev).wait_stream(side)
            for t, s_ in zip(persistent, saved):  # ... undone: it advanced the shaping terms and the step counter
                t.copy_(s_)
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self._static_out = self._post()
        self._graph.replay()
        return self._static_out

    def rollout_fields(self, n_steps: int):
        """(name, shape, dtype) of every per-step output ``rollout()`` writes for ``n_steps`` steps: what a caller-provided
        ``out`` must hold (shard.PackedRollout lays these out in ONE buffer that is gathered across GPUs as it is)."""
        assert self._one_launch, "rollout() needs a scenario whose Environment.step is one launch"
        return self._post.rollout_fields(n_steps)

    def rollout(self, actions: List[Tensor], out: Optional[Dict[str, Tensor]] = None) -> Dict[str, Tensor]:
        """K consecutive ``step()`` calls with given actions in ONE kernel launch.

        ``actions[i]`` = agent i's actions for all K steps, ``[K, num_envs, action_size]``.  Returns
        ``{"obs": [K, n_agents, num_envs, obs_dim], "rew": [K, n_agents, num_envs], "done": [K, num_envs]}`` (+ the
        scenario's per-step info terms): entry k is what the k-th ``step()`` would have returned, bit for bit.  No
        resets in between (like K plain ``step()`` calls: ``done`` environments keep running until the caller
        resets them); the 64 environments of a tile stay in LDS for the whole rollout.  For the scenarios whose step
        is one launch (balance, transport); others: loop over ``step()``.  ``out``: the per-step outputs are written
        into the caller's tensors (names, shapes, dtypes of ``rollout_fields(K)``) instead of fresh ones - the kernel
        stores straight into a rollout buffer, no copy between the rollout and whatever consumes it."""
        assert self._one_launch and getattr(self._post, "rollout_ok", True), \
            "rollout() needs a scenario whose Environment.step is one launch without a batch-wide reduction (balance, transport)"
        assert len(actions) == self.n_agents, f"Expecting actions for {self.n_agents}, got {len(actions)} actions"
        K = int(actions[0].shape[0])
        self._bound = None  # (the ingest slots and the post-step's buffer struct are re-pointed below: step_bound() re-binds)
        self._ingest.prepare_rollout(actions, K, self.validate_actions)
        desc, buffers, out = self._post.prepare_rollout(K, out)
        self._launch.rollout(self._post.kind, desc, buffers, K)
        self._lidar_cache = None
        return out

    def bind(self, actions: List[Tensor]) -> None:
        """Caller-owned action tensors for ``step_bound()``: the policy writes the next actions INTO these tensors
        (``actions[i].copy_(...)``, an ``out=`` argument ...) and every ``step_bound()`` is then a single foreign call -
        no per-step checks, allocations or views on the host (a ``step()`` from Python costs 18 us of them around an
        11 us kernel; SURVEY.md 8f-3, the policy-in-the-loop case that ``rollout()`` does not cover).  The outputs are
        static buffers owned by the environment, overwritten by the next ``step_bound()``.  For the scenarios whose
        step is one launch (balance, transport, football, navigation while a tile has a CU to itself), ``validate_actions=False``
        (the reference's asserts are a host sync per step)."""
        assert self._one_launch, "bind() needs a scenario whose Environment.step is one launch"
        assert not self.validate_actions, "bind() needs validate_actions=False (the asserts are a host sync per step)"
        assert len(actions) == self.n_agents, f"Expecting actions for {self.n_agents}, got {len(actions)} actions"
        self._ingest.prepare(actions)
        assert all(k is a for k, a in zip(self._ingest._keep, actions)), (
            "bind() takes the action tensors as they are: float32 (int64 when discrete) [num_envs, action_size], contiguous, "
            "on the environment's device")
        self._bound_actions, self._bound = list(actions), None

    def step_bound(self):
        """One ``Environment.step()`` on the bound action tensors: (obs, rews, dones, infos) - static buffers."""
        b = self._bound
        if b is None:  # first call, or a reset / step() / set_state since: pointers may have moved
            assert self._bound_actions is not None, "step_bound() needs bind(actions) first"
            self._ingest.prepare(self._bound_actions)
            b = self._bound = self._post.prepare(dedicated=True)  # a set of output buffers of its own
        self._launch(self._post.kind, b[0], b[1], False)
        self._lidar_cache = None
        fin = getattr(self._post, "finish", None)  # (tensor ops on the step's outputs: football's red rewards)
        return b[2] if fin is None else fin(b[2])

    def _step_eager(self, actions):
        assert len(actions) == self.n_agents, f"Expecting actions for {self.n_agents}, got {len(actions)} actions"
        self._bound = None
        if self._one_launch:
            self._ingest.prepare(actions)
            if self.validate_actions and self._launch.can_gate(self._post.kind):
                # the check enqueued, the step launched GATED on its result, the host's wait behind both (fused.StepLauncher.gated)
                seq = self._ingest.validate_begin()
                saved = self._post.save_bound()
                desc, buffers, result = self._post.prepare()
                self._launch.gated(self._post.kind, desc, buffers)
                flags = self._ingest.validate_end(seq)
                if flags:
                    self._launch.refused()
                    self._post.restore_bound(saved)
                    self._ingest._raise(flags)
            else:
                if self.validate_actions:
                    self._ingest.validate()  # raises before the world is touched, like the reference
                desc, buffers, result = self._post.prepare()
                self._launch(self._post.kind, desc, buffers, False)
            self._lidar_cache = None
            fin = getattr(self._post, "finish", None)  # (tensor ops on the step's outputs: football's red rewards, ball_pos)
            return result if fin is None else fin(result)
        if self._ingest_in_step:
            self._ingest.prepare(actions)
            if self.validate_actions:
                self._ingest.validate()
            self._launch(0, None, None, False)
        else:
            if self._ingest is not None:
                self._ingest(actions, self.validate_actions)
            else:
                self._ingest_torch(actions)
            self.scenario.pre_step()
            self.world.step()
        self.scenario.post_step()
        self._lidar_cache = None
        if self._post is not None:  # one launch: reward, observation, done, info, step counter
            return self._post()
        self.steps += 1
        rews = [self.scenario.reward(a).clone() for a in self.agents]
        obs = self._observations()
        infos = [self.scenario.info(a) for a in self.agents]
        dones = self.done()
        return obs, rews, dones, infos

    def _ingest_torch(self, actions):
        for i, agent in enumerate(self.agents):
            a = actions[i]
            if not isinstance(a, Tensor):
                a = torch.tensor(a, dtype=torch.float32, device=self.device)
            if a.dim() == 1:
                a = a.unsqueeze(-1)
            assert a.shape[0] == self.num_envs, f"Actions used in input of env must be of len {self.num_envs}, got {a.shape[0]}"
            self._set_action(a, agent)
        for agent in self.world.agents:  # scripted agents included (environment.py:390-391)
            self.scenario.env_process_action(agent)

    def _observations(self):
        if any(a.sensors for a in self.world.agents) and self.device.type == "cuda":
            self._lidar_cache = self.world.cast_rays_all()  # one launch for every sensor of every agent
        self.scenario._lidar_cache = self._lidar_cache
        return [self.scenario.observation(a) for a in self.agents]

    def done(self) -> Tensor:
        dones = self.scenario.done().clone()
        if self.max_steps is not None:
            dones = dones | (self.steps >= self.max_steps)
        return dones


def make_env(
    scenario: Union[str, BaseScenario],
    num_envs: int,
    device: Union[torch.device, str] = "cuda:0",
    continuous_actions: bool = True,
    max_steps: Optional[int] = None,
    seed: Optional[int] = None,
    clamp_actions: bool = False,
    validate_actions: bool = True,
    graph: bool = False,
    fused: Optional[bool] = None,
    specialize: Optional[bool] = None,
    **kwargs,
) -> Environment:
    """vmas.make_env(...) for the scenarios shipped in ``vectorizedmultiagentsimulator_amd.scenarios``."""
    if isinstance(scenario, str):
        name = scenario[:-3] if scenario.endswith(".py") else scenario
        try:
            mod = importlib.import_module(f"{__package__}.scenarios.{name}")
        except ModuleNotFoundError as e:
            raise ValueError(f"scenario {name!r} is not available natively; attach() the reference's instead") from e
        scenario = mod.Scenario()
    return Environment(
        scenario, num_envs=num_envs, device=device, continuous_actions=continuous_actions, max_steps=max_steps,
        seed=seed, clamp_actions=clamp_actions, validate_actions=validate_actions, graph=graph, fused=fused,
        specialize=specialize, **kwargs,
    )
