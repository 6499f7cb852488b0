"""`navigation` - BASELINE config 4 (reference: vmas/scenarios/navigation.py:23-285): every agent
drives to its own goal; agents carry a 12-ray LIDAR that sees the other agents; substeps=2.
Entity order: goal 0..n-1 (landmarks), agent_0..n-1.  The LIDAR of all agents is one kernel launch
(``World.cast_rays_all``); ``observation`` slices the cached result."""
from __future__ import annotations

from typing import Optional

import torch

from ..core import Agent, Landmark, Sphere, World
from ..scenario import BaseScenario, check_kwargs_consumed, keep, spawn_entities_randomly
from ..sensors import Lidar


class Scenario(BaseScenario):
    def make_world(self, batch_dim: int, device, **kwargs) -> World:
        self.n_agents = kwargs.pop("n_agents", 4)
        self.collisions = kwargs.pop("collisions", True)
        self.world_spawning_x = kwargs.pop("world_spawning_x", 1)
        self.world_spawning_y = kwargs.pop("world_spawning_y", 1)
        self.enforce_bounds = kwargs.pop("enforce_bounds", False)
        self.agents_with_same_goal = kwargs.pop("agents_with_same_goal", 1)
        self.split_goals = kwargs.pop("split_goals", False)
        self.observe_all_goals = kwargs.pop("observe_all_goals", False)
        self.lidar_range = kwargs.pop("lidar_range", 0.35)
        self.agent_radius = kwargs.pop("agent_radius", 0.1)
        self.n_lidar_rays = kwargs.pop("n_lidar_rays", 12)
        self.shared_rew = kwargs.pop("shared_rew", True)
        self.pos_shaping_factor = kwargs.pop("pos_shaping_factor", 1)
        self.final_reward = kwargs.pop("final_reward", 0.01)
        self.agent_collision_penalty = kwargs.pop("agent_collision_penalty", -1)
        kwargs.pop("comms_range", None)
        world_kwargs = {k: kwargs.pop(k) for k in ("exact_broad_phase", "lanes_per_env") if k in kwargs}
        check_kwargs_consumed(kwargs)
        self.min_distance_between_entities = self.agent_radius * 2 + 0.05
        self.min_collision_distance = 0.005
        xs, ys = (self.world_spawning_x, self.world_spawning_y) if self.enforce_bounds else (None, None)
        assert 1 <= self.agents_with_same_goal <= self.n_agents
        if self.agents_with_same_goal > 1:
            assert not self.collisions, "If agents share goals they cannot be collidables"
        world = World(batch_dim, device, substeps=2, x_semidim=xs, y_semidim=ys, **world_kwargs)
        for i in range(self.n_agents):
            sensors = (
                [Lidar(world, n_rays=self.n_lidar_rays, max_range=self.lidar_range,
                       entity_filter=lambda e: isinstance(e, Agent))]
                if self.collisions else None
            )
            agent = Agent(name=f"agent_{i}", collide=self.collisions, shape=Sphere(radius=self.agent_radius),
                          sensors=sensors)
            world.add_agent(agent)
            goal = Landmark(name=f"goal {i}", collide=False)
            world.add_landmark(goal)
            agent.goal = goal
        self._lidar_cache = None
        world.fused_post = 3  # VMAS_POST_NAVIGATION: the epilogue a run-time specialisation of this world carries (World.specialize)
        return world

    def reset_world_at(self, env_index: Optional[int] = None):
        w = self.world
        xb, yb = (-self.world_spawning_x, self.world_spawning_x), (-self.world_spawning_y, self.world_spawning_y)
        spawn_entities_randomly(w.agents, w, env_index, self.min_distance_between_entities, xb, yb)
        occupied = torch.stack([a.state.pos for a in w.agents], dim=1)
        if env_index is not None:
            occupied = occupied[env_index].unsqueeze(0)
        goals = [a.goal for a in w.agents]
        # goals are placed like entities (one position per agent, not overlapping anything placed so far)
        spawn_entities_randomly(goals, w, env_index, self.min_distance_between_entities, xb, yb,
                                occupied_positions=occupied)
        if self.agents_with_same_goal > 1 or self.split_goals:
            poses = [g.state.pos.clone() for g in goals]
            for i, a in enumerate(w.agents):
                gi = int(i // self.agents_with_same_goal) if self.split_goals else (0 if i < self.agents_with_same_goal else i)
                a.goal.set_pos(poses[gi] if env_index is None else poses[gi][env_index], batch_index=env_index)
        for a in w.agents:
            shaping = torch.linalg.vector_norm(a.state.pos - a.goal.state.pos, dim=1) * self.pos_shaping_factor
            if env_index is None:
                keep(a, "pos_shaping", shaping)
                a.pos_rew = torch.zeros_like(shaping)
                a.agent_collision_rew = torch.zeros_like(shaping)
            else:
                a.pos_shaping[env_index] = shaping[env_index]
        if env_index is None:
            self.pos_rew = torch.zeros(w.batch_dim, device=w.device)
            self.final_rew = self.pos_rew.clone()

    def agent_reward(self, agent):  # navigation.py:232-242
        agent.distance_to_goal = torch.linalg.vector_norm(agent.state.pos - agent.goal.state.pos, dim=-1)
        agent.on_goal = agent.distance_to_goal < agent.goal.shape.radius
        pos_shaping = agent.distance_to_goal * self.pos_shaping_factor
        agent.pos_rew = agent.pos_shaping - pos_shaping
        keep(agent, "pos_shaping", pos_shaping)
        return agent.pos_rew

    def reward(self, agent):  # navigation.py:200-230, sync-free
        w = self.world
        if agent is w.agents[0]:
            self.pos_rew = torch.zeros(w.batch_dim, device=w.device)
            for a in w.agents:
                self.pos_rew = self.pos_rew + self.agent_reward(a)
                a.agent_collision_rew = torch.zeros(w.batch_dim, device=w.device)
            all_reached = torch.stack([a.on_goal for a in w.agents], dim=-1).all(dim=-1)
            self.final_rew = torch.where(all_reached, torch.full_like(self.pos_rew, self.final_reward),
                                         torch.zeros_like(self.pos_rew))
            if self.collisions:
                # all agent pairs at once (the reference loops over pairs with a host-synchronising
                # collides() each, navigation.py:218-229): agents are consecutive entities, so their
                # positions are one [A, 2, B] block of the packed state
                A_ = len(w.agents)
                i0 = w.agents[0]._index
                P = w._packed_state()[i0 : i0 + A_, 0:2, : w.batch_dim]  # [A, 2, B] view
                diff = P[:, None] - P[None, :]  # [A, A, 2, B]
                centre = torch.linalg.vector_norm(diff, dim=2)  # [A, A, B]
                r = self.agent_radius
                overlap_any = (centre <= r + r).any(dim=-1)  # [A, A]: World.collides' batch-global reduction
                dist = (centre - r) - r  # World.get_distance for spheres, same operation order
                hit = overlap_any[:, :, None] & (dist <= self.min_collision_distance)
                hit = hit & ~torch.eye(A_, dtype=torch.bool, device=w.device)[:, :, None]
                per_agent = hit.to(torch.float32).sum(dim=1) * self.agent_collision_penalty  # [A, B]
                for i, a in enumerate(w.agents):
                    a.agent_collision_rew = per_agent[i]
        pos_reward = self.pos_rew if self.shared_rew else agent.pos_rew
        return pos_reward + self.final_rew + agent.agent_collision_rew

    def observation(self, agent):  # navigation.py:244-263
        if self.observe_all_goals:
            goal_poses = [agent.state.pos - a.goal.state.pos for a in self.world.agents]
        else:
            goal_poses = [agent.state.pos - agent.goal.state.pos]
        extra = []
        if self.collisions:
            s = agent.sensors[0]
            extra = [s._max_range - s.measure(self._lidar_cache)]
        return torch.cat([agent.state.pos, agent.state.vel] + goal_poses + extra, dim=-1)

    def done(self):
        return torch.stack(
            [torch.linalg.vector_norm(a.state.pos - a.goal.state.pos, dim=-1) < a.shape.radius for a in self.world.agents],
            dim=-1,
        ).all(-1)

    def info(self, agent):
        return {"pos_rew": self.pos_rew if self.shared_rew else agent.pos_rew, "final_rew": self.final_rew,
                "agent_collisions": agent.agent_collision_rew}

    def fused_reset_program(self):
        """``reset_world_at`` (navigation.py:137-198) as a spawn program: agents, then one goal per agent, all at the
        minimum distance from everything placed before.  Shared goals are a post-processing of the draws: tensor path."""
        if self.agents_with_same_goal > 1 or self.split_goals:
            return None
        w = self.world
        xb, yb = (-self.world_spawning_x, self.world_spawning_x), (-self.world_spawning_y, self.world_spawning_y)
        ops = [("uniform", e, xb, yb, self.min_distance_between_entities, 0) for e in list(w.agents) + [a.goal for a in w.agents]]
        terms = [(lambda a=a: a.pos_shaping, a, a.goal, self.pos_shaping_factor) for a in w.agents]
        return {"ops": ops, "terms": terms, "flags": []}

    def make_fused_post(self, env):
        """reward + observation (LIDAR included) + done + info as one kernel (fused.NavigationPost)."""
        from ..fused import NavigationPost
        return NavigationPost(env) if NavigationPost.supports(env) is None else None
