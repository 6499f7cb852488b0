"""`balance` - BASELINE.json's headline scenario (reference: vmas/scenarios/balance.py).

n agents hold a line from below; a package rides on the line and must reach a goal;
everything is pulled down by gravity toward a floor box.  World content and the reset
distribution follow the reference (balance.py:16-84 and 86-216): entity order
goal, package, line, floor, agent_0..n-1; gravity (0,-0.05); y_semidim 1.
"""
from __future__ import annotations

from typing import Optional

import torch

from ..core import Agent, Box, Landmark, Line, Sphere, World
from ..scenario import BaseScenario, check_kwargs_consumed, keep


class Scenario(BaseScenario):
    def make_world(self, batch_dim: int, device, **kwargs) -> World:
        self.n_agents = kwargs.pop("n_agents", 3)
        self.package_mass = kwargs.pop("package_mass", 5)
        self.random_package_pos_on_line = kwargs.pop("random_package_pos_on_line", True)
        world_kwargs = {k: kwargs.pop(k) for k in ("exact_broad_phase", "lanes_per_env") if k in kwargs}
        check_kwargs_consumed(kwargs)
        assert self.n_agents > 1
        self.line_length = 0.8
        self.agent_radius = 0.03
        self.shaping_factor = 100
        self.fall_reward = -10

        world = World(batch_dim, device, gravity=(0.0, -0.05), y_semidim=1, **world_kwargs)
        for i in range(self.n_agents):
            world.add_agent(Agent(name=f"agent_{i}", shape=Sphere(self.agent_radius), u_multiplier=0.7))
        self.goal = Landmark(name="goal", collide=False, shape=Sphere())
        world.add_landmark(self.goal)
        self.package = Landmark(name="package", collide=True, movable=True, shape=Sphere(), mass=self.package_mass)
        self.package.goal = self.goal
        world.add_landmark(self.package)
        self.line = Landmark(name="line", shape=Line(length=self.line_length), collide=True, movable=True,
                             rotatable=True, mass=5)
        world.add_landmark(self.line)
        self.floor = Landmark(name="floor", collide=True, shape=Box(length=10, width=1))
        world.add_landmark(self.floor)
        world.epilogue_hint = (1, 0)  # VMAS_POST_BALANCE: Environment.step is one launch with the post-step as its epilogue
        return world

    def reset_world_at(self, env_index: Optional[int] = None):
        w = self.world
        n = 1 if env_index is not None else w.batch_dim
        dev = w.device

        def uniform(lo, hi):
            return torch.zeros((n, 1), device=dev, dtype=torch.float32).uniform_(lo, hi)

        def const(v):
            return torch.full((n, 1), v, device=dev, dtype=torch.float32)

        half = self.line_length / 2
        r_pkg = self.package.shape.radius
        goal_pos = torch.cat([uniform(-1.0, 1.0), uniform(0.0, w.y_semidim)], dim=1)
        line_pos = torch.cat([uniform(-1.0 + half, 1.0 - half), const(-w.y_semidim + self.agent_radius * 2)], dim=1)
        spread = (half - r_pkg) if self.random_package_pos_on_line else 0.0
        package_rel = torch.cat([uniform(-spread, spread), const(r_pkg)], dim=1)

        span = self.line_length - self.agent_radius
        for i, agent in enumerate(w.agents):
            offset = torch.tensor([-span / 2 + i * span / (self.n_agents - 1), -self.agent_radius * 2],
                                  device=dev, dtype=torch.float32)
            agent.set_pos(line_pos + offset, batch_index=env_index)
        self.line.set_pos(line_pos, batch_index=env_index)
        self.goal.set_pos(goal_pos, batch_index=env_index)
        self.line.set_rot(torch.zeros(1, device=dev, dtype=torch.float32), batch_index=env_index)
        self.package.set_pos(line_pos + package_rel, batch_index=env_index)
        self.floor.set_pos(
            torch.tensor([0, -w.y_semidim - self.floor.shape.width / 2 - self.agent_radius], device=dev,
                         dtype=torch.float32),
            batch_index=env_index,
        )
        # reward bookkeeping (balance.py:202-216)
        self.compute_on_the_ground()
        shaping = torch.linalg.vector_norm(self.package.state.pos - self.goal.state.pos, dim=1) * self.shaping_factor
        if env_index is None:
            keep(self, "global_shaping", shaping)
            self.pos_rew = torch.zeros(w.batch_dim, device=dev, dtype=torch.float32)
            self.ground_rew = self.pos_rew.clone()
        else:
            self.global_shaping[env_index] = shaping[env_index]

    def compute_on_the_ground(self):  # balance.py:218-221
        w = self.world
        self.on_the_ground = w.is_overlapping(self.line, self.floor) | w.is_overlapping(self.package, self.floor)

    def reward(self, agent):  # balance.py:223-241
        if agent is self.world.agents[0]:
            self.compute_on_the_ground()
            self.package_dist = torch.linalg.vector_norm(self.package.state.pos - self.goal.state.pos, dim=1)
            self.ground_rew = torch.where(self.on_the_ground, torch.full_like(self.package_dist, float(self.fall_reward)),
                                          torch.zeros_like(self.package_dist))
            global_shaping = self.package_dist * self.shaping_factor
            self.pos_rew = self.global_shaping - global_shaping
            keep(self, "global_shaping", global_shaping)
        return self.ground_rew + self.pos_rew

    def observation(self, agent):  # balance.py:243-258
        pkg, line = self.package.state, self.line.state
        a = agent.state
        return torch.cat(
            [a.pos, a.vel, a.pos - pkg.pos, a.pos - line.pos, pkg.pos - self.goal.state.pos, pkg.vel, line.vel,
             line.ang_vel, line.rot % torch.pi],
            dim=-1,
        )

    def done(self):  # balance.py:260-263
        return self.on_the_ground | self.world.is_overlapping(self.package, self.goal)

    def info(self, agent):
        return {"pos_rew": self.pos_rew, "ground_rew": self.ground_rew}

    def fused_reset_program(self):
        """``reset_world_at`` (balance.py:86-216) as a spawn program for the masked-reset kernel (fused.MaskedReset)."""
        w = self.world
        half, r_pkg, r = self.line_length / 2, self.package.shape.radius, self.agent_radius
        line_y = -w.y_semidim + r * 2
        spread = (half - r_pkg) if self.random_package_pos_on_line else 0.0
        span = self.line_length - r
        ops = [("uniform", self.goal, (-1.0, 1.0), (0.0, w.y_semidim), 0.0, 0),
               ("uniform", self.line, (-1.0 + half, 1.0 - half), (line_y, line_y), 0.0, 1),
               ("offset", self.package, self.line, (-spread, spread), r_pkg)]
        for i, agent in enumerate(w.agents):
            dx = -span / 2 + i * span / (self.n_agents - 1)
            ops.append(("offset", agent, self.line, (dx, dx), -r * 2))
        ops.append(("fixed", self.floor, 0.0, -w.y_semidim - self.floor.shape.width / 2 - r))
        return {"ops": ops, "terms": [(lambda: self.global_shaping, self.package, self.goal, self.shaping_factor)],
                "flags": [lambda: self.on_the_ground]}

    def make_fused_post(self, env):
        """reward + observation + done + info of every agent as one kernel (fused.BalancePost)."""
        from ..fused import BalancePost
        return BalancePost(env)
