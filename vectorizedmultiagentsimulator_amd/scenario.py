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


def check_kwargs_consumed(kwargs: dict, warn: bool = True):
    """utils.py:321-330: unknown scenario kwargs only warn."""
    if kwargs and warn:
        import warnings

        warnings.warn(f"Scenario kwargs: {kwargs} passed but not used by the scenario.")
