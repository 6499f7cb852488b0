"""`transport` - BASELINE config 3 (reference: vmas/scenarios/transport.py:14-191): agents push
box packages onto a goal.  Entity order: goal, package 0.., agent_0..; bounded world."""
from __future__ import annotations

from typing import Optional

import torch

from ..core import Agent, Box, Landmark, Sphere, World
from ..scenario import BaseScenario, check_kwargs_consumed, keep, spawn_entities_randomly


class Scenario(BaseScenario):
    def make_world(self, batch_dim: int, device, **kwargs) -> World:
        n_agents = kwargs.pop("n_agents", 4)
        self.n_packages = kwargs.pop("n_packages", 1)
        self.package_width = kwargs.pop("package_width", 0.15)
        self.package_length = kwargs.pop("package_length", 0.15)
        self.package_mass = kwargs.pop("package_mass", 50)
        world_kwargs = {k: kwargs.pop(k) for k in ("exact_broad_phase", "lanes_per_env") if k in kwargs}
        check_kwargs_consumed(kwargs)
        self.shaping_factor = 100
        self.world_semidim = 1
        self.agent_radius = 0.03
        semidim = self.world_semidim + 2 * self.agent_radius + max(self.package_length, self.package_width)
        world = World(batch_dim, device, x_semidim=semidim, y_semidim=semidim, **world_kwargs)
        for i in range(n_agents):
            world.add_agent(Agent(name=f"agent_{i}", shape=Sphere(self.agent_radius), u_multiplier=0.6))
        self.goal = Landmark(name="goal", collide=False, shape=Sphere(radius=0.15))
        world.add_landmark(self.goal)
        self.packages = []
        for i in range(self.n_packages):
            p = Landmark(name=f"package {i}", collide=True, movable=True, mass=self.package_mass,
                         shape=Box(length=self.package_length, width=self.package_width))
            p.goal = self.goal
            self.packages.append(p)
            world.add_landmark(p)
        world.epilogue_hint = (2, self.n_packages)  # VMAS_POST_TRANSPORT (one-launch Environment.step)
        return world

    def reset_world_at(self, env_index: Optional[int] = None):
        w = self.world
        b = (-self.world_semidim, self.world_semidim)
        spawn_entities_randomly(w.agents, w, env_index, self.agent_radius * 2, b, b)
        occupied = torch.stack([a.state.pos for a in w.agents], dim=1)
        if env_index is not None:
            occupied = occupied[env_index].unsqueeze(0)
        min_dist = max(p.shape.circumscribed_radius() + self.goal.shape.radius + 0.01 for p in self.packages)
        spawn_entities_randomly([self.goal] + self.packages, w, env_index, min_dist, b, b, occupied_positions=occupied)
        for p in self.packages:
            keep(p, "on_goal", w.is_overlapping(p, p.goal))
            shaping = torch.linalg.vector_norm(p.state.pos - p.goal.state.pos, dim=1) * self.shaping_factor
            if env_index is None:
                keep(p, "global_shaping", shaping)
            else:
                p.global_shaping[env_index] = shaping[env_index]

    def reward(self, agent):  # transport.py:131-163
        if agent is self.world.agents[0]:
            self.rew = torch.zeros(self.world.batch_dim, device=self.world.device, dtype=torch.float32)
            for p in self.packages:
                p.dist_to_goal = torch.linalg.vector_norm(p.state.pos - p.goal.state.pos, dim=1)
                p.on_goal = self.world.is_overlapping(p, p.goal)
                shaping = p.dist_to_goal * self.shaping_factor
                self.rew = self.rew + torch.where(p.on_goal, torch.zeros_like(shaping), p.global_shaping - shaping)
                keep(p, "global_shaping", shaping)
        return self.rew

    def observation(self, agent):  # transport.py:165-182
        obs = [agent.state.pos, agent.state.vel]
        for p in self.packages:
            obs += [p.state.pos - p.goal.state.pos, p.state.pos - agent.state.pos, p.state.vel,
                    p.on_goal.unsqueeze(-1).to(torch.float32)]
        return torch.cat(obs, dim=-1)

    def done(self):
        return torch.stack([p.on_goal for p in self.packages], dim=1).all(dim=-1)

    def fused_reset_program(self):
        """``reset_world_at`` (transport.py:86-129) as a spawn program: the agents, then the goal and the packages, each
        kept at its call's minimum distance from everything placed before (ScenarioUtils.spawn_entities_randomly)."""
        w = self.world
        b = (-self.world_semidim, self.world_semidim)
        ops = [("uniform", a, b, b, self.agent_radius * 2, 0) for a in w.agents]
        min_dist = max(p.shape.circumscribed_radius() + self.goal.shape.radius + 0.01 for p in self.packages)
        ops += [("uniform", e, b, b, min_dist, 0) for e in [self.goal] + self.packages]
        terms = [(lambda p=p: p.global_shaping, p, p.goal, self.shaping_factor) for p in self.packages]
        return {"ops": ops, "terms": terms, "flags": [(lambda p=p: p.on_goal) for p in self.packages]}

    def make_fused_post(self, env):
        """reward + observation + done of every agent as one kernel (fused.TransportPost)."""
        from .. import _abi
        from ..fused import TransportPost
        if len(self.packages) > _abi.ENV_MAX_PACKAGES or len(env.agents) > _abi.ENV_MAX_AGENTS:
            return None
        return TransportPost(env)
