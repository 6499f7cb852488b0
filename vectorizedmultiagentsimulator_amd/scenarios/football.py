"""`football` - BASELINE config 5 (reference: vmas/scenarios/football.py): two teams of sphere agents
and a ball on a walled pitch with two goals; substeps=2, drag 0.05, every agent speed-limited.

Ported: the learning-vs-learning game - ``ai_red_agents=False``, ``ai_blue_agents=False`` (all
agents are policy agents; red agents see and act in a mirrored frame), dense + sparse reward,
flat observations, random or formation spawning, the scripted ball (anti-stuck impulse near the
walls).  Not ported, and refused loudly: the heuristic ``AgentPolicy`` opponents (the reference's
default ``ai_red_agents=True``; football.py:1686-2360 is a spline planner, not part of the
simulator's hot path), ``enable_shooting``, ``physically_different``, ``dict_obs``.

Entity order (reference make_world, football.py:121-160): landmarks = 4 walls, 6 goal lines, 2 nets;
agents = blue 0..n-1, red 0..m-1, Ball.
"""
from __future__ import annotations

from typing import List, Optional

import torch

from ..core import Agent, Box, Landmark, Line, Sphere, World
from ..scenario import BaseScenario, check_kwargs_consumed, keep

X, Y = 0, 1


def ball_action_script(ball: Agent, world: World):
    """football.py:1620-1680: a small impulse away from a wall the ball is resting against (scaled down by
    its speed), none along x in front of a goal mouth.  (Both velocity factors use vel.y, as the reference
    does.)"""
    dist_thres = world.agent_size * 2
    vel_thres, impulse = 0.3, 0.05
    pos, vel = ball.state.pos, ball.state.vel

    def near(d):  # 1 at the wall, 0 from dist_thres away
        return 1 - d.clamp(max=dist_thres) / dist_thres

    upper = near(world.pitch_width / 2 - pos[:, Y])
    lower = near(world.pitch_width / 2 + pos[:, Y])
    right = near(world.pitch_length / 2 - pos[:, X])
    left = near(world.pitch_length / 2 + pos[:, X])
    slow = 1 - vel[:, Y].abs().clamp(max=vel_thres) / vel_thres
    dist_action = torch.stack([left - right, lower - upper], dim=1)
    vel_action = torch.stack([slow, slow], dim=1)
    actions = dist_action * vel_action * impulse
    in_goal_mouth = (pos[:, Y] < world.goal_size / 2) & (pos[:, Y] > -world.goal_size / 2)
    ax = torch.where(in_goal_mouth, torch.zeros_like(actions[:, X]), actions[:, X])
    ball.action.u = torch.stack([ax, actions[:, Y]], dim=1)


class Scenario(BaseScenario):
    def init_params(self, kwargs: dict):  # football.py:24-119 (same names and defaults); pops what it uses
        kwargs.pop("viewer_size", None)
        self.n_blue_agents = kwargs.pop("n_blue_agents", 3)
        self.n_red_agents = kwargs.pop("n_red_agents", 3)
        self.ai_red_agents = kwargs.pop("ai_red_agents", True)
        self.ai_blue_agents = kwargs.pop("ai_blue_agents", False)
        self.physically_different = kwargs.pop("physically_different", False)
        self.spawn_in_formation = kwargs.pop("spawn_in_formation", False)
        self.only_blue_formation = kwargs.pop("only_blue_formation", True)
        self.formation_agents_per_column = kwargs.pop("formation_agents_per_column", 2)
        self.randomise_formation_indices = kwargs.pop("randomise_formation_indices", False)
        self.formation_noise = kwargs.pop("formation_noise", 0.2)
        for ai_only in ("n_traj_points", "ai_strength", "ai_decision_strength", "ai_precision_strength", "disable_ai_red"):
            kwargs.pop(ai_only, None)
        self.agent_size = kwargs.pop("agent_size", 0.025)
        self.goal_size = kwargs.pop("goal_size", 0.35)
        self.goal_depth = kwargs.pop("goal_depth", 0.1)
        self.pitch_length = kwargs.pop("pitch_length", 3.0)
        self.pitch_width = kwargs.pop("pitch_width", 1.5)
        self.ball_mass = kwargs.pop("ball_mass", 0.25)
        self.ball_size = kwargs.pop("ball_size", 0.02)
        self.u_multiplier = kwargs.pop("u_multiplier", 0.1)
        self.enable_shooting = kwargs.pop("enable_shooting", False)
        for shooting_only in ("u_rot_multiplier", "u_shoot_multiplier", "shooting_radius", "shooting_angle"):
            kwargs.pop(shooting_only, None)
        self.max_speed = kwargs.pop("max_speed", 0.15)
        self.ball_max_speed = kwargs.pop("ball_max_speed", 0.3)
        self.dense_reward = kwargs.pop("dense_reward", True)
        self.pos_shaping_factor_ball_goal = kwargs.pop("pos_shaping_factor_ball_goal", 10.0)
        self.pos_shaping_factor_agent_ball = kwargs.pop("pos_shaping_factor_agent_ball", 0.1)
        self.distance_to_ball_trigger = kwargs.pop("distance_to_ball_trigger", 0.4)
        self.scoring_reward = kwargs.pop("scoring_reward", 100.0)
        self.observe_teammates = kwargs.pop("observe_teammates", True)
        self.observe_adversaries = kwargs.pop("observe_adversaries", True)
        self.dict_obs = kwargs.pop("dict_obs", False)
        if kwargs.pop("dense_reward_ratio", None) is not None:
            raise ValueError("dense_reward_ratio in football is deprecated, please use `dense_reward` "
                             "which is a bool that turns on/off the dense reward")
        for flag, what in ((self.ai_red_agents, "ai_red_agents=True (the heuristic AgentPolicy opponents): pass "
                                                "ai_red_agents=False"),
                           (self.ai_blue_agents, "ai_blue_agents=True"), (self.enable_shooting, "enable_shooting=True"),
                           (self.physically_different, "physically_different=True"), (self.dict_obs, "dict_obs=True")):
            if flag:
                raise NotImplementedError(f"football: {what} is not available natively; attach() the reference's "
                                          "scenario instead (adapter.py)")

    def make_world(self, batch_dim: int, device, **kwargs) -> World:
        world_kwargs = {k: kwargs.pop(k) for k in ("exact_broad_phase", "lanes_per_env") if k in kwargs}
        self.init_params(kwargs)
        check_kwargs_consumed(kwargs)
        world = World(batch_dim, device, dt=0.1, drag=0.05, substeps=2,
                      x_semidim=self.pitch_length / 2 + self.goal_depth - self.agent_size,
                      y_semidim=self.pitch_width / 2 - self.agent_size, **world_kwargs)
        world.agent_size, world.pitch_width, world.pitch_length = self.agent_size, self.pitch_width, self.pitch_length
        world.goal_size, world.goal_depth = self.goal_size, self.goal_depth

        def player(name):
            return Agent(name=name, shape=Sphere(radius=self.agent_size), u_multiplier=[self.u_multiplier] * 2,
                         max_speed=self.max_speed, alpha=1)

        self.blue_agents = [player(f"agent_blue_{i}") for i in range(self.n_blue_agents)]
        self.red_agents = [player(f"agent_red_{i}") for i in range(self.n_red_agents)]
        for a in self.blue_agents + self.red_agents:
            world.add_agent(a)
        world.blue_agents, world.red_agents = self.blue_agents, self.red_agents
        self.ball = Agent(name="Ball", shape=Sphere(radius=self.ball_size), action_script=ball_action_script,
                          max_speed=self.ball_max_speed, mass=self.ball_mass, alpha=1)
        world.add_agent(self.ball)
        world.ball = self.ball

        wall_len = self.pitch_width / 2 - self.agent_size - self.goal_size / 2
        hl, hw, gs, gd, r = self.pitch_length / 2, self.pitch_width / 4, self.goal_size, self.goal_depth, self.agent_size
        half_pi = torch.pi / 2
        # name, shape, collide, position, rotation (football.py:686-1020)
        self._static = [
            ("Right Top Wall", Line(length=wall_len), True, (hl, hw + gs / 4), half_pi),
            ("Left Top Wall", Line(length=wall_len), True, (-hl, hw + gs / 4), half_pi),
            ("Right Bottom Wall", Line(length=wall_len), True, (hl, -hw - gs / 4), half_pi),
            ("Left Bottom Wall", Line(length=wall_len), True, (-hl, -hw - gs / 4), half_pi),
            ("Right Goal Back", Line(length=gs), True, (hl + gd - r, 0.0), half_pi),
            ("Left Goal Back", Line(length=gs), True, (-hl - gd + r, 0.0), half_pi),
            ("Right Goal Top", Line(length=gd), True, (hl + gd / 2 - r, gs / 2), None),
            ("Left Goal Top", Line(length=gd), True, (-hl - gd / 2 + r, gs / 2), None),
            ("Right Goal Bottom", Line(length=gd), True, (hl + gd / 2 - r, -gs / 2), None),
            ("Left Goal Bottom", Line(length=gd), True, (-hl - gd / 2 + r, -gs / 2), None),
            ("Blue Net", Box(length=gd, width=gs), False, (-hl - gd / 2 + r / 2, 0.0), None),
            ("Red Net", Box(length=gd, width=gs), False, (hl + gd / 2 - r / 2, 0.0), None),
        ]
        self._static_landmarks = []
        for name, shape, collide, _, _ in self._static:
            lm = Landmark(name=name, collide=collide, movable=False, shape=shape)
            world.add_landmark(lm)
            self._static_landmarks.append(lm)
        self.blue_net, self.red_net = self._static_landmarks[10], self._static_landmarks[11]
        world.blue_net, world.red_net = self.blue_net, self.red_net

        f32 = dict(device=device, dtype=torch.float32)
        self.left_goal_pos = torch.tensor([-self.pitch_length / 2 - self.ball_size / 2, 0], **f32)
        self.right_goal_pos = -self.left_goal_pos
        self._mirror = (torch.tensor([1.0, 1.0], **f32), torch.tensor([-1.0, 1.0], **f32))
        self._reset_agent_range = torch.tensor([self.pitch_length / 2, self.pitch_width], **f32)
        self._reset_agent_offset_blue = torch.tensor([-self.pitch_length / 2 + self.agent_size, -self.pitch_width / 2], **f32)
        self._reset_agent_offset_red = torch.tensor([-self.agent_size, -self.pitch_width / 2], **f32)
        zeros = torch.zeros(batch_dim, **f32)
        self._done = torch.zeros(batch_dim, device=device, dtype=torch.bool)
        self._sparse_reward_blue, self._sparse_reward_red = zeros.clone(), zeros.clone()
        self._dense_reward_blue, self._dense_reward_red = zeros.clone(), zeros.clone()
        for name in ("pos_rew_blue", "pos_rew_red", "pos_rew_agent_blue", "pos_rew_agent_red"):
            setattr(self.ball, name, zeros.clone())
        return world

    # ------------------------------------------------------------------ reset (football.py:162-171, 388-600)
    def reset_world_at(self, env_index: Optional[int] = None):
        self._reset_agents(env_index)
        self._reset_ball(env_index)
        w = self.world
        for lm, (_, _, _, pos, rot) in zip(self._static_landmarks, self._static):
            lm.set_pos(torch.tensor(pos, device=w.device, dtype=torch.float32), batch_index=env_index)
            if rot is not None:
                lm.set_rot(torch.tensor([rot], device=w.device, dtype=torch.float32), batch_index=env_index)
        if env_index is None:
            self._done.zero_()
        else:
            self._done[env_index] = False

    def _random_spawn(self, blue: bool, env_index):
        w = self.world
        shape = (1, 2) if env_index is not None else (w.batch_dim, 2)
        offset = self._reset_agent_offset_blue if blue else self._reset_agent_offset_red
        return torch.rand(shape, device=w.device) * self._reset_agent_range + offset

    def _spawn_formation(self, agents: List[Agent], blue: bool, env_index):  # football.py:417-463
        w = self.world
        if self.randomise_formation_indices:
            agents = [agents[i] for i in torch.randperm(len(agents)).tolist()]
        k = 0
        endpoint = -(self.pitch_length / 2 + self.goal_depth) * (1 if blue else -1)
        for x in torch.linspace(0, endpoint, len(agents) // self.formation_agents_per_column + 3).tolist()[1:-1]:
            if k >= len(agents):
                break
            column = agents[k : k + self.formation_agents_per_column]
            for y in torch.linspace(self.pitch_width / 2, -self.pitch_width / 2, len(column) + 2).tolist()[1:-1]:
                shape = (2,) if env_index is not None else (w.batch_dim, 2)
                noise = (torch.rand(shape, device=w.device) - 0.5) * self.formation_noise
                agents[k].set_pos(torch.tensor([x, y], device=w.device, dtype=torch.float32) + noise, batch_index=env_index)
                k += 1

    def _reset_agents(self, env_index):
        w = self.world
        if self.spawn_in_formation:
            self._spawn_formation(self.blue_agents, True, env_index)
            if not self.only_blue_formation:
                self._spawn_formation(self.red_agents, False, env_index)
        else:
            for a in self.blue_agents:
                a.set_pos(self._random_spawn(True, env_index), batch_index=env_index)
        if not self.spawn_in_formation or self.only_blue_formation:
            for a in self.red_agents:
                a.set_pos(self._random_spawn(False, env_index), batch_index=env_index)
                a.set_rot(torch.tensor([torch.pi], device=w.device, dtype=torch.float32), batch_index=env_index)

    def _closest_agent_to_ball(self, team: List[Agent]) -> torch.Tensor:  # football.py:586-600 -> [B]
        ball = self.ball.state.pos
        return torch.stack([torch.linalg.vector_norm(a.state.pos - ball, dim=-1) for a in team], dim=-1).min(dim=-1)[0]

    def _reset_ball(self, env_index):
        ball = self.ball
        terms = {
            "min_agent_dist_to_ball_blue": (self, self._closest_agent_to_ball(self.blue_agents)),
            "min_agent_dist_to_ball_red": (self, self._closest_agent_to_ball(self.red_agents)),
        }
        terms["pos_shaping_blue"] = (ball, torch.linalg.vector_norm(ball.state.pos - self.right_goal_pos, dim=-1)
                                     * self.pos_shaping_factor_ball_goal)
        terms["pos_shaping_red"] = (ball, torch.linalg.vector_norm(ball.state.pos - self.left_goal_pos, dim=-1)
                                    * self.pos_shaping_factor_ball_goal)
        terms["pos_shaping_agent_blue"] = (ball, terms["min_agent_dist_to_ball_blue"][1] * self.pos_shaping_factor_agent_ball)
        terms["pos_shaping_agent_red"] = (ball, terms["min_agent_dist_to_ball_red"][1] * self.pos_shaping_factor_agent_ball)
        for name, (obj, value) in terms.items():
            if env_index is None:
                keep(obj, name, value)
            else:
                getattr(obj, name)[env_index] = value[env_index]

    # ------------------------------------------------------------------ step hooks
    def process_action(self, agent: Agent):  # football.py:1050-1057
        if agent in self.red_agents:  # red agents act in a mirrored frame: x flipped
            u = agent.action.u
            agent.action.u = torch.stack([-u[:, X], u[:, Y]], dim=1)

    def reward(self, agent: Agent):  # football.py:1121-1161
        if agent is self.world.agents[0]:
            ball = self.ball
            bx, by = ball.state.pos[:, X], ball.state.pos[:, Y]
            over_right = bx > self.pitch_length / 2 + self.ball_size / 2
            over_left = bx < -self.pitch_length / 2 - self.ball_size / 2
            in_mouth = (by <= self.goal_size / 2) & (by >= -self.goal_size / 2)
            blue_score, red_score = over_right & in_mouth, over_left & in_mouth
            self._sparse_reward_blue = self.scoring_reward * blue_score - self.scoring_reward * red_score
            self._sparse_reward_red = -self._sparse_reward_blue
            keep(self, "_done", blue_score | red_score)
            if self.dense_reward:
                self._dense_reward_blue = self._reward_ball_to_goal(True) + self._reward_all_agent_to_ball(True)
                self._dense_reward_red = self._reward_ball_to_goal(False) + self._reward_all_agent_to_ball(False)
        if agent in self.blue_agents:
            return self._sparse_reward_blue + self._dense_reward_blue
        return self._sparse_reward_red + self._dense_reward_red

    def _reward_ball_to_goal(self, blue: bool):  # football.py:1163-1187
        ball, side = self.ball, "blue" if blue else "red"
        dist = torch.linalg.vector_norm(ball.state.pos - (self.right_goal_pos if blue else self.left_goal_pos), dim=-1)
        setattr(ball, f"distance_to_goal_{side}", dist)
        shaping = dist * self.pos_shaping_factor_ball_goal
        rew = getattr(ball, f"pos_shaping_{side}") - shaping
        setattr(ball, f"pos_rew_{side}", rew)
        keep(ball, f"pos_shaping_{side}", shaping)
        return rew

    def _reward_all_agent_to_ball(self, blue: bool):  # football.py:1189-1219
        ball, side = self.ball, "blue" if blue else "red"
        min_dist = self._closest_agent_to_ball(self.blue_agents if blue else self.red_agents)
        keep(self, f"min_agent_dist_to_ball_{side}", min_dist)
        shaping = min_dist * self.pos_shaping_factor_agent_ball
        ball_moving = torch.linalg.vector_norm(ball.state.vel, dim=-1) > 1e-6
        close = min_dist < self.distance_to_ball_trigger
        rew = torch.where(close | ball_moving, torch.zeros_like(shaping), getattr(ball, f"pos_shaping_agent_{side}") - shaping)
        setattr(ball, f"pos_rew_agent_{side}", rew)
        keep(ball, f"pos_shaping_agent_{side}", shaping)
        return rew

    def observation(self, agent: Agent):  # football.py:1221-1460, flat layout
        blue = agent in self.blue_agents
        my_team, other_team = (self.blue_agents, self.red_agents) if blue else (self.red_agents, self.blue_agents)
        goal_pos = (self.right_goal_pos if blue else self.left_goal_pos).expand(self.world.batch_dim, 2)
        flip = self._mirror[0 if blue else 1]  # red: x mirrored

        def m(t):
            return t * flip

        a, ball = agent.state, self.ball.state
        pos, vel, force = m(a.pos), m(a.vel), m(a.force)
        bpos, bvel, bforce, gpos = m(ball.pos), m(ball.vel), m(ball.force), m(goal_pos)
        parts = [force, pos - bpos, vel - bvel, bpos - gpos, bvel, bforce, pos - gpos, vel]
        others = (other_team if self.observe_adversaries else []), ([t for t in my_team if t is not agent]
                                                                    if self.observe_teammates else [])
        for group in others:
            for o in group:
                opos, ovel, oforce = m(o.state.pos), m(o.state.vel), m(o.state.force)
                parts += [pos - opos, vel - ovel, ovel, oforce]
        return torch.cat(parts, dim=-1)

    def done(self):
        return self._done

    def info(self, agent: Agent):  # football.py:1483-1515
        blue = agent in self.blue_agents
        side = "blue" if blue else "red"
        ball = self.ball
        min_dist = getattr(self, f"min_agent_dist_to_ball_{side}")
        return {
            "sparse_reward": self._sparse_reward_blue if blue else self._sparse_reward_red,
            "ball_goal_pos_rew": getattr(ball, f"pos_rew_{side}"),
            "all_agent_ball_pos_rew": getattr(ball, f"pos_rew_agent_{side}"),
            "ball_pos": ball.state.pos.clone(),  # (a state VIEW would change under the caller at the next step)
            "dist_ball_to_goal": getattr(ball, f"pos_shaping_{side}") / self.pos_shaping_factor_ball_goal,
            "min_agent_dist_to_ball": min_dist,
            "touching_ball": min_dist <= self.agent_size + self.ball_size + 1e-2,
        }

    # ------------------------------------------------------------------ fused stages (fused.py)
    def fused_action_factors(self, agent: Agent):
        """What ``process_action`` does, as per-dimension factors for the fused action ingest."""
        return [-1.0, 1.0] if agent in self.red_agents else None

    def fused_agent_scripts(self):
        """The scripted ball, for the device-side script (VMAS_SCRIPT_FOOTBALL_BALL)."""
        from .. import _abi
        return [dict(kind=_abi.SCRIPT_FOOTBALL_BALL, agent=self.ball,
                     params=[self.agent_size * 2, self.pitch_width / 2, self.pitch_length / 2, self.goal_size / 2])]

    def fused_reset_program(self):
        """``reset_world_at`` (football.py:162-171: reset_agents 388-414, reset_ball 512-584, reset_walls / reset_goals
        686-1020) as a spawn program for ``vmas_env_reset_where``: blue agents uniform in the left half, red agents uniform
        in the right half and turned by pi, the ball at the centre (World.reset zeroed it), every wall and goal line at its
        fixed pose; then the ball's four shaping terms and the two closest-agent distances.  Formation spawning draws a
        permutation on the host: tensor path."""
        if self.spawn_in_formation:
            return None
        import math
        L, Wd, r = self.pitch_length, self.pitch_width, self.agent_size
        # torch.rand * [L / 2, W] + offset (football.py:465-490): U([offset, offset + range))
        blue_x, red_x, ys = (-L / 2 + r, r), (-r, L / 2 - r), (-Wd / 2, Wd / 2)
        ops = [("uniform", a, blue_x, ys, 0.0, i) for i, a in enumerate(self.blue_agents)]
        ops += [("uniform", a, red_x, ys, 0.0, len(self.blue_agents) + i, math.pi) for i, a in enumerate(self.red_agents)]
        ops += [("fixed", lm, pos[0], pos[1], rot) for lm, (_, _, _, pos, rot) in zip(self._static_landmarks, self._static)]
        ball = self.ball
        right, left = self.right_goal_pos.tolist(), self.left_goal_pos.tolist()
        terms = [
            (lambda: self.min_agent_dist_to_ball_blue, "min", self.blue_agents, ball, 1.0),
            (lambda: self.min_agent_dist_to_ball_red, "min", self.red_agents, ball, 1.0),
            (lambda: ball.pos_shaping_blue, "point", ball, right, self.pos_shaping_factor_ball_goal),
            (lambda: ball.pos_shaping_red, "point", ball, left, self.pos_shaping_factor_ball_goal),
            (lambda: ball.pos_shaping_agent_blue, "min", self.blue_agents, ball, self.pos_shaping_factor_agent_ball),
            (lambda: ball.pos_shaping_agent_red, "min", self.red_agents, ball, self.pos_shaping_factor_agent_ball),
        ]
        return {"ops": ops, "terms": terms, "flags": [lambda: self._done]}

    def make_fused_post(self, env):
        """reward + observation + done + info of every agent as one kernel (fused.FootballPost)."""
        from .. import _abi
        from ..fused import FootballPost
        n = len(self.blue_agents) + len(self.red_agents)
        same_dims = self.n_blue_agents == self.n_red_agents or not (self.observe_teammates or self.observe_adversaries) \
            or (self.observe_teammates and self.observe_adversaries)
        if not self.dense_reward or n + 1 > _abi.ENV_MAX_AGENTS or not same_dims:
            return None
        return FootballPost(env)
