"""Scenario contract (vmas/simulator/scenario.py ``BaseScenario``): the same five
methods - make_world / reset_world_at / observation / reward / done (+ info,
process_action, pre_step, post_step) - and the same ``env_*`` entry points the
environment drives (scenario.py:82-98)."""
from __future__ import annotations

from typing import Dict, Optional

import torch
from torch import Tensor

from .core import Agent, World


class BaseScenario:
    def __init__(self):
        self._world: Optional[World] = None

    @property
    def world(self) -> World:
        assert self._world is not None, "You first need to set `self._world` in the `make_world` method"
        return self._world

    # -- driven by the environment (scenario.py:82-98)
    def env_make_world(self, batch_dim: int, device, **kwargs) -> World:
        self._world = self.make_world(batch_dim, device, **kwargs)
        return self._world

    def env_reset_world_at(self, env_index: Optional[int]):
        self.world.reset(env_index)
        self.reset_world_at(env_index)

    def env_process_action(self, agent: Agent):
        if agent.action_script is not None:
            agent.action_callback(self.world)
        self.process_action(agent)
        agent.dynamics.process_action()

    # -- to be implemented by scenarios
    def make_world(self, batch_dim: int, device, **kwargs) -> World:
        raise NotImplementedError

    def reset_world_at(self, env_index: Optional[int] = None):
        raise NotImplementedError

    def observation(self, agent: Agent) -> Tensor:
        raise NotImplementedError

    def reward(self, agent: Agent) -> Tensor:
        raise NotImplementedError

    def done(self) -> Tensor:
        return torch.zeros(self.world.batch_dim, dtype=torch.bool, device=self.world.device)

    def info(self, agent: Agent) -> Dict[str, Tensor]:
        return {}

    def process_action(self, agent: Agent):
        return

    def pre_step(self):
        return

    def post_step(self):
        return


def keep(obj, name: str, value: Tensor) -> Tensor:
    """Persistent per-scenario tensor (shaping caches ...): created once, then updated IN PLACE, so a
    HIP graph captured over ``Environment.step`` keeps reading and writing the same memory."""
    # (registered on BOTH paths: a tensor created elsewhere - football's `_done` in make_world - and only ever updated
    # in place is persistent state too; Environment snapshots / blends these around graph warm-ups and masked resets)
    obj.__dict__.setdefault("_kept_names", set()).add(name)
    cur = getattr(obj, name, None)
    if isinstance(cur, Tensor) and cur.shape == value.shape and cur.dtype == value.dtype and cur.device == value.device:
        cur.copy_(value)
        return cur
    setattr(obj, name, value.clone())
    return getattr(obj, name)


def check_kwargs_consumed(kwargs: dict, warn: bool = True):
    """utils.py:321-330: unknown scenario kwargs only warn."""
    if kwargs and warn:
        import warnings

        warnings.warn(f"Scenario kwargs: {kwargs} passed but not used by the scenario.")


def spawn_entities_randomly(entities, world: World, env_index: Optional[int], min_dist_between_entities: float,
                            x_bounds, y_bounds, occupied_positions: Optional[Tensor] = None, max_tries: int = 64):
    """ScenarioUtils.spawn_entities_randomly (utils.py:241-319) without the host round trips:
    the reference loops ``while torch.any(overlaps)``; here a fixed number of rejection rounds
    runs on the device (each re-draws only the environments that still overlap), which samples
    the same distribution as long as the placement is not nearly infeasible."""
    n = 1 if env_index is not None else world.batch_dim
    dev = world.device
    if occupied_positions is None:
        occupied_positions = torch.zeros((n, 0, 2), device=dev)

    def draw():
        x = torch.empty((n, 1, 1), device=dev, dtype=torch.float32).uniform_(*x_bounds)
        y = torch.empty((n, 1, 1), device=dev, dtype=torch.float32).uniform_(*y_bounds)
        return torch.cat([x, y], dim=2)

    for entity in entities:
        pos = draw()
        if occupied_positions.shape[1] > 0:
            for _ in range(max_tries):
                overlaps = (torch.cdist(occupied_positions, pos) < min_dist_between_entities).squeeze(2).any(dim=1)
                pos = torch.where(overlaps[:, None, None], draw(), pos)
        occupied_positions = torch.cat([occupied_positions, pos], dim=1)
        entity.set_pos(pos.squeeze(1), batch_index=env_index)
