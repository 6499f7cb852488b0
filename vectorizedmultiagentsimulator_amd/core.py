"""Host-side object model of the MI355X-native simulator.

Mirrors the names, argument meaning and error behaviour of the reference's data model
(vmas/simulator/core.py:48-1232: ``Sphere/Box/Line``, ``EntityState/AgentState``,
``Action``, ``Entity/Landmark/Agent``, ``World``; vmas/simulator/joints.py) so that
scenarios written against the reference port over unchanged - but it is designed the
other way round: the WORLD owns one packed structure-of-arrays buffer
(``state[E,6,ld]``, ``agent_ft[A,3,ld]``, environment index fastest; see
include/vmas_hip.h) and every ``entity.state.pos`` is a strided *view* into it.
Property setters ``copy_`` into the view instead of rebinding tensors
(reference: core.py:222-284), so the HIP kernels and the Python side always see the
same memory and ``World.step()`` is a single kernel launch with no gather/scatter.

``World.step()`` has NO CPU implementation: it calls libvmas_hip.so and raises if the
library or a GPU is missing.  Construction, reset and the state setters work on any
torch device, which is what the CPU-side host-logic tests exercise.
"""
from __future__ import annotations

import math
from typing import Callable, Dict, List, Optional, Sequence, Tuple, Union

import torch
from torch import Tensor

from . import _abi as A
from .spec import WorldSpec, spec_from_world

X, Y = 0, 1
LINE_MIN_DIST = 4 / 6e2  # utils.py:28
COLLISION_FORCE = 100  # utils.py:29
JOINT_FORCE = 130  # utils.py:30
TORQUE_CONSTRAINT_FORCE = 1  # utils.py:31
DRAG = 0.25  # utils.py:33
LINEAR_FRICTION = 0.0
ANGULAR_FRICTION = 0.0


# ----------------------------------------------------------------------------------
# shapes (core.py:84-203)
# ----------------------------------------------------------------------------------
def _positive(what: str, value) -> float:
    """Shape dimensions must be positive numbers (the reference asserts the same, core.py:133-135, 143, 173)."""
    if not value > 0:
        raise AssertionError(f"a shape's {what} must be a positive number, got {value!r}")
    return value


class Shape:
    def moment_of_inertia(self, mass: float) -> float:
        raise NotImplementedError

    def circumscribed_radius(self) -> float:
        raise NotImplementedError

    def get_delta_from_anchor(self, anchor: Tuple[float, float]) -> Tuple[float, float]:
        raise NotImplementedError


class Box(Shape):
    def __init__(self, length: float = 0.3, width: float = 0.1, hollow: bool = False):
        self._length, self._width, self.hollow = _positive("length", length), _positive("width", width), hollow

    length = property(lambda self: self._length)
    width = property(lambda self: self._width)

    def get_delta_from_anchor(self, anchor):
        return anchor[X] * self.length / 2, anchor[Y] * self.width / 2

    def moment_of_inertia(self, mass):
        return (1 / 12) * mass * (self.length**2 + self.width**2)

    def circumscribed_radius(self):
        return math.sqrt((self.length / 2) ** 2 + (self.width / 2) ** 2)


class Sphere(Shape):
    def __init__(self, radius: float = 0.05):
        self._radius = _positive("radius", radius)

    radius = property(lambda self: self._radius)

    def get_delta_from_anchor(self, anchor):
        # fp32 arithmetic and the normalisation quirk of core.py:151-158 kept on purpose
        delta = torch.tensor([anchor[X] * self.radius, anchor[Y] * self.radius]).to(torch.float32)
        delta_norm = torch.linalg.vector_norm(delta)
        if delta_norm > self.radius:
            delta /= delta_norm * self.radius
        return tuple(delta.tolist())

    def moment_of_inertia(self, mass):
        return (1 / 2) * mass * self.radius**2

    def circumscribed_radius(self):
        return self.radius


class Line(Shape):
    def __init__(self, length: float = 0.5):
        self._length, self._width = _positive("length", length), 2

    length = property(lambda self: self._length)
    width = property(lambda self: self._width)

    def moment_of_inertia(self, mass):
        return (1 / 12) * mass * (self.length**2)

    def circumscribed_radius(self):
        return self.length / 2

    def get_delta_from_anchor(self, anchor):
        return anchor[X] * self.length / 2, 0.0


# ----------------------------------------------------------------------------------
# state: views into the world's packed buffer
# ----------------------------------------------------------------------------------
class EntityState:
    """pos/vel [B,2], rot/ang_vel [B,1] - views; assignment copies into the view."""

    _FIELDS = {"pos": (0, 2), "vel": (2, 4), "rot": (4, 5), "ang_vel": (5, 6)}

    def __init__(self, entity: "Entity"):
        self._entity = entity

    def _view(self, name: str) -> Tensor:
        w = self._entity._world
        assert w is not None, "First add an entity to the world before setting its state"
        lo, hi = self._FIELDS[name]
        return w._packed_state()[self._entity._index, lo:hi, : w.batch_dim].T

    def _assign(self, name: str, value: Tensor):
        v = self._view(name)
        assert (
            value.shape[0] == v.shape[0]
        ), f"Internal state must match batch dim, got {value.shape[0]}, expected {v.shape[0]}"
        v.copy_(value.to(v.device).reshape(v.shape))
        self._entity._world._query_cache = None

    pos = property(lambda s: s._view("pos"), lambda s, v: s._assign("pos", v))
    vel = property(lambda s: s._view("vel"), lambda s, v: s._assign("vel", v))
    rot = property(lambda s: s._view("rot"), lambda s, v: s._assign("rot", v))
    ang_vel = property(lambda s: s._view("ang_vel"), lambda s, v: s._assign("ang_vel", v))

    def _reset(self, env_index: Optional[int]):  # core.py:286-296
        for name in self._FIELDS:
            v = self._view(name)
            if env_index is None:
                v.zero_()
            else:
                v[env_index] = 0.0


class AgentState(EntityState):
    """+ force [B,2], torque [B,1] (views into agent_ft) and comm state c."""

    def __init__(self, entity: "Entity"):
        super().__init__(entity)
        self._c: Optional[Tensor] = None

    def _ft_view(self, lo: int, hi: int) -> Tensor:
        w = self._entity._world
        assert w is not None, "First add an entity to the world before setting its state"
        return w._packed_agent_ft()[self._entity._agent_index, lo:hi, : w.batch_dim].T

    def _ft_assign(self, lo: int, hi: int, value: Tensor):
        v = self._ft_view(lo, hi)
        assert (
            value.shape[0] == v.shape[0]
        ), f"Internal state must match batch dim, got {value.shape[0]}, expected {v.shape[0]}"
        v.copy_(value.to(v.device).reshape(v.shape))

    force = property(lambda s: s._ft_view(0, 2), lambda s, v: s._ft_assign(0, 2, v))
    torque = property(lambda s: s._ft_view(2, 3), lambda s, v: s._ft_assign(2, 3, v))

    @property
    def c(self):
        return self._c

    @c.setter
    def c(self, c: Tensor):
        self._c = c

    def _reset(self, env_index: Optional[int]):  # core.py:377-388
        super()._reset(env_index)
        for lo, hi in ((0, 2), (2, 3)):
            v = self._ft_view(lo, hi)
            if env_index is None:
                v.zero_()
            else:
                v[env_index] = 0.0
        if self._c is not None:
            if env_index is None:
                self._c = torch.zeros_like(self._c)
            else:
                self._c[env_index] = 0.0


class Action:
    """Physical action ``u`` and communication ``c`` of an agent (core.py:414-517)."""

    def __init__(self, u_range, u_multiplier, u_noise, action_size: int):
        self.action_size = action_size
        self.u_range = u_range
        self.u_multiplier = u_multiplier
        self.u_noise = u_noise
        for name in ("u_range", "u_multiplier", "u_noise"):
            v = getattr(self, name)
            if isinstance(v, Sequence):
                assert len(v) == action_size, f"Action attribute {name} has wrong length {len(v)} != {action_size}"
        self.u: Optional[Tensor] = None
        self.c: Optional[Tensor] = None
        self._tensors: Dict[Tuple[str, str], Tensor] = {}

    def _as_tensor(self, name: str, device) -> Tensor:
        key = (name, str(device))
        if key not in self._tensors:
            v = getattr(self, name)
            v = list(v) if isinstance(v, Sequence) else [v] * self.action_size
            self._tensors[key] = torch.tensor(v, dtype=torch.float32, device=device)
        return self._tensors[key]

    def u_range_tensor_on(self, device) -> Tensor:
        return self._as_tensor("u_range", device)

    def u_multiplier_tensor_on(self, device) -> Tensor:
        return self._as_tensor("u_multiplier", device)


# ----------------------------------------------------------------------------------
# dynamics (vmas/simulator/dynamics/*): action.u -> state.force / state.torque
# ----------------------------------------------------------------------------------
class Dynamics:
    needed_action_size = 0
    agent: "Agent" = None

    def process_action(self):
        raise NotImplementedError


class Holonomic(Dynamics):  # dynamics/holonomic.py:9-15
    needed_action_size = 2

    def process_action(self):
        self.agent.state.force = self.agent.action.u[:, :2]


class HolonomicWithRotation(Dynamics):  # dynamics/holonomic_with_rot.py
    needed_action_size = 3

    def process_action(self):
        self.agent.state.force = self.agent.action.u[:, :2]
        self.agent.state.torque = self.agent.action.u[:, 2].unsqueeze(-1)


# ----------------------------------------------------------------------------------
# entities
# ----------------------------------------------------------------------------------
class Entity:
    def __init__(
        self,
        name: str,
        movable: bool = False,
        rotatable: bool = False,
        collide: bool = True,
        density: float = 25.0,
        mass: float = 1.0,
        shape: Shape = None,
        v_range: float = None,
        max_speed: float = None,
        color=None,
        is_joint: bool = False,
        drag: float = None,
        linear_friction: float = None,
        angular_friction: float = None,
        gravity: Union[float, Tensor, Sequence[float]] = None,
        collision_filter: Callable[["Entity"], bool] = lambda _: True,
    ):
        self._name = name
        self._movable, self._rotatable, self._collide = movable, rotatable, collide
        self._density, self._mass = density, mass
        self._shape = shape if shape is not None else Sphere()
        self._v_range, self._max_speed = v_range, max_speed
        self._color = color
        self._is_joint = is_joint
        self._collision_filter = collision_filter
        self._drag = drag
        self._linear_friction, self._angular_friction = linear_friction, angular_friction
        if gravity is not None and not isinstance(gravity, Tensor):
            gravity = torch.tensor(gravity, dtype=torch.float32)
        self._gravity = gravity
        self._goal = None
        self._state = EntityState(self)
        self._world: Optional["World"] = None
        self._index = -1
        self._observers: List = []

    # observable (utils.py Observable): joints re-place their link when an end moves
    def subscribe(self, observer):
        self._observers.append(observer)

    def notify_observers(self, *args, **kwargs):
        for o in self._observers:
            o.notify(self, *args, **kwargs)

    name = property(lambda s: s._name)
    movable = property(lambda s: s._movable)
    rotatable = property(lambda s: s._rotatable)
    collide = property(lambda s: s._collide)
    shape = property(lambda s: s._shape)
    max_speed = property(lambda s: s._max_speed)
    v_range = property(lambda s: s._v_range)
    is_joint = property(lambda s: s._is_joint)
    drag = property(lambda s: s._drag)
    state = property(lambda s: s._state)
    angular_friction = property(lambda s: s._angular_friction)
    color = property(lambda s: s._color)

    @property
    def batch_dim(self):
        return None if self._world is None else self._world.batch_dim

    @property
    def device(self):
        return None if self._world is None else self._world.device

    def _static_changed(self):
        if self._world is not None:
            self._world._invalidate_backend()

    @property
    def mass(self):
        return self._mass

    @mass.setter
    def mass(self, mass: float):  # core.py:634-636 (debug/het_mass changes it at reset)
        self._mass = mass
        self._static_changed()

    @property
    def moment_of_inertia(self):
        return self.shape.moment_of_inertia(self.mass)

    @property
    def linear_friction(self):
        return self._linear_friction

    @linear_friction.setter
    def linear_friction(self, value):
        self._linear_friction = value
        self._static_changed()

    @property
    def gravity(self):
        return self._gravity

    @gravity.setter
    def gravity(self, value):
        if value is not None and not isinstance(value, Tensor):
            value = torch.tensor(value, dtype=torch.float32)
        self._gravity = value
        self._static_changed()

    @property
    def collision_filter(self):
        return self._collision_filter

    @collision_filter.setter
    def collision_filter(self, f):
        self._collision_filter = f
        self._static_changed()

    @property
    def goal(self):
        return self._goal

    @goal.setter
    def goal(self, goal: "Entity"):
        self._goal = goal

    def collides(self, entity: "Entity") -> bool:  # core.py:621-624
        if not self.collide:
            return False
        return self._collision_filter(entity)

    def _reset(self, env_index: Optional[int]):
        self.state._reset(env_index)

    # core.py:733-761
    def set_pos(self, pos: Tensor, batch_index: Optional[int]):
        self._set_state_property("pos", pos, batch_index)

    def set_vel(self, vel: Tensor, batch_index: Optional[int]):
        self._set_state_property("vel", vel, batch_index)

    def set_rot(self, rot: Tensor, batch_index: Optional[int]):
        self._set_state_property("rot", rot, batch_index)

    def set_ang_vel(self, ang_vel: Tensor, batch_index: Optional[int]):
        self._set_state_property("ang_vel", ang_vel, batch_index)

    def _set_state_property(self, name: str, new: Tensor, batch_index: Optional[int]):
        assert self._world is not None, f"Tried to set property of {self.name} without adding it to the world"
        if batch_index is not None:
            assert 0 <= batch_index < self.batch_dim, f"Index must be between 0 and {self.batch_dim}, got {batch_index}"
        view = self.state._view(name)
        new = new.to(view.device)
        if batch_index is None:
            if len(new.shape) > 1 and new.shape[0] == self.batch_dim:
                view.copy_(new)
            else:
                view.copy_(new.reshape(1, -1).expand(view.shape))
        else:
            view[batch_index] = new.reshape(-1)
        self._world._query_cache = None
        self.notify_observers()


class Landmark(Entity):
    pass


class Agent(Entity):
    def __init__(
        self,
        name: str,
        shape: Shape = None,
        movable: bool = True,
        rotatable: bool = True,
        collide: bool = True,
        density: float = 25.0,
        mass: float = 1.0,
        f_range: float = None,
        max_f: float = None,
        t_range: float = None,
        max_t: float = None,
        v_range: float = None,
        max_speed: float = None,
        color=None,
        alpha: float = 0.5,
        obs_range: float = None,
        obs_noise: float = None,
        u_noise: Union[float, Sequence[float]] = 0.0,
        u_range: Union[float, Sequence[float]] = 1.0,
        u_multiplier: Union[float, Sequence[float]] = 1.0,
        action_script: Callable[["Agent", "World"], None] = None,
        sensors: List = None,
        c_noise: float = 0.0,
        silent: bool = True,
        adversary: bool = False,
        drag: float = None,
        linear_friction: float = None,
        angular_friction: float = None,
        gravity=None,
        collision_filter: Callable[[Entity], bool] = lambda _: True,
        render_action: bool = False,
        dynamics: Dynamics = None,
        action_size: int = None,
        discrete_action_nvec: List[int] = None,
    ):
        super().__init__(
            name, movable, rotatable, collide, density, mass, shape, v_range, max_speed, color,
            is_joint=False, drag=drag, linear_friction=linear_friction, angular_friction=angular_friction,
            gravity=gravity, collision_filter=collision_filter,
        )
        if obs_range == 0.0:
            assert sensors is None, f"Blind agent cannot have sensors, got {sensors}"
        if action_size is not None and discrete_action_nvec is not None and action_size != len(discrete_action_nvec):
            raise ValueError(
                f"action_size {action_size} is inconsistent with discrete_action_nvec {discrete_action_nvec}"
            )
        if discrete_action_nvec is not None and not all(n > 1 for n in discrete_action_nvec):
            raise ValueError(f"All values in discrete_action_nvec must be greater than 1, got {discrete_action_nvec}")
        self._obs_range, self._obs_noise = obs_range, obs_noise
        self._f_range, self._max_f, self._t_range, self._max_t = f_range, max_f, t_range, max_t
        self._action_script = action_script
        self._sensors: List = []
        for s in sensors or []:
            self.add_sensor(s)
        self._c_noise, self._silent, self._adversary = c_noise, silent, adversary
        self._render_action, self._alpha = render_action, alpha
        self.dynamics = dynamics if dynamics is not None else Holonomic()
        if action_size is not None:
            self.action_size = action_size
        elif discrete_action_nvec is not None:
            self.action_size = len(discrete_action_nvec)
        else:
            self.action_size = self.dynamics.needed_action_size
        self.discrete_action_nvec = discrete_action_nvec if discrete_action_nvec is not None else [3] * self.action_size
        self.dynamics.agent = self
        self._action = Action(u_range=u_range, u_multiplier=u_multiplier, u_noise=u_noise, action_size=self.action_size)
        self._state = AgentState(self)
        self._agent_index = -1

    def add_sensor(self, sensor):
        sensor.agent = self
        self._sensors.append(sensor)

    action_script = property(lambda s: s._action_script)
    sensors = property(lambda s: s._sensors)
    action = property(lambda s: s._action)
    u_range = property(lambda s: s._action.u_range)
    u_multiplier = property(lambda s: s._action.u_multiplier)
    max_f = property(lambda s: s._max_f)
    f_range = property(lambda s: s._f_range)
    max_t = property(lambda s: s._max_t)
    t_range = property(lambda s: s._t_range)
    silent = property(lambda s: s._silent)
    adversary = property(lambda s: s._adversary)
    obs_noise = property(lambda s: s._obs_noise if s._obs_noise is not None else 0)
    c_noise = property(lambda s: s._c_noise)

    def action_callback(self, world: "World"):  # core.py:966-982
        self._action_script(self, world)
        if self._silent or world.dim_c == 0:
            assert self._action.c is None, f"Agent {self.name} should not communicate but action script communicates"
        assert self._action.u is not None, f"Action script of {self.name} should set u action"
        assert self._action.u.shape[1] == self.action_size, f"Scripted action of agent {self.name} has wrong shape"


# ----------------------------------------------------------------------------------
# joints (vmas/simulator/joints.py)
# ----------------------------------------------------------------------------------
class JointConstraint:
    """Two anchor points held ``dist`` apart (joints.py:148-216)."""

    def __init__(self, entity_a, entity_b, anchor_a=(0.0, 0.0), anchor_b=(0.0, 0.0), dist: float = 0.0,
                 rotate: bool = True, fixed_rotation: Optional[float] = None):
        assert entity_a != entity_b, "Cannot join same entity"
        for anchor in (anchor_a, anchor_b):
            assert max(anchor) <= 1 and min(anchor) >= -1, f"Joint anchor points should be between -1 and 1, got {anchor}"
        assert dist >= 0, f"Joint dist must be >= 0, got {dist}"
        if fixed_rotation is not None:
            assert not rotate, "If fixed rotation is provided, rotate should be False"
        if rotate:
            assert fixed_rotation is None, "If you provide a fixed rotation, rotate should be False"
            fixed_rotation = 0.0
        self.entity_a, self.entity_b = entity_a, entity_b
        self.anchor_a, self.anchor_b = anchor_a, anchor_b
        self.dist, self.fixed_rotation, self.rotate = dist, fixed_rotation, rotate

    def pos_point(self, entity) -> Tensor:  # joints.py:209-216
        anchor = self.anchor_a if entity is self.entity_a else self.anchor_b
        d = entity.shape.get_delta_from_anchor(anchor)
        rot = entity.state.rot.squeeze(-1)
        c, s = torch.cos(rot), torch.sin(rot)
        dx = torch.tensor(d[0], dtype=torch.float32, device=rot.device)
        dy = torch.tensor(d[1], dtype=torch.float32, device=rot.device)
        return entity.state.pos + torch.stack([dx * c - dy * s, dx * s + dy * c], dim=-1)


class Joint:
    """joints.py:21-144.  ``dist > 0`` inserts a movable, rotatable link landmark."""

    def __init__(self, entity_a, entity_b, anchor_a=(0.0, 0.0), anchor_b=(0.0, 0.0), rotate_a: bool = True,
                 rotate_b: bool = True, dist: float = 0.0, collidable: bool = False, width: float = 0.0,
                 mass: float = 1.0, fixed_rotation_a: Optional[float] = None, fixed_rotation_b: Optional[float] = None):
        assert entity_a != entity_b, "Cannot join same entity"
        for anchor in (anchor_a, anchor_b):
            assert max(anchor) <= 1 and min(anchor) >= -1, f"Joint anchor points should be between -1 and 1, got {anchor}"
        assert dist >= 0, f"Joint dist must be >= 0, got {dist}"
        if dist == 0:
            assert not collidable, "Cannot have collidable joint with dist 0"
            assert width == 0, "Cannot have width for joint with dist 0"
            assert fixed_rotation_a == fixed_rotation_b, "If dist is 0, fixed_rotation_a and fixed_rotation_b should be the same"
        if fixed_rotation_a is not None:
            assert not rotate_a, "If you provide a fixed rotation for a, rotate_a should be False"
        if fixed_rotation_b is not None:
            assert not rotate_b, "If you provide a fixed rotation for b, rotate_b should be False"
        if width > 0:
            assert collidable
        self.entity_a, self.entity_b = entity_a, entity_b
        self.rotate_a, self.rotate_b = rotate_a, rotate_b
        self.fixed_rotation_a, self.fixed_rotation_b = fixed_rotation_a, fixed_rotation_b
        self.landmark = None
        self.joint_constraints: List[JointConstraint] = []
        if dist == 0:
            self.joint_constraints.append(
                JointConstraint(entity_a, entity_b, anchor_a, anchor_b, dist, rotate_a and rotate_b, fixed_rotation_a)
            )
        else:
            entity_a.subscribe(self)
            entity_b.subscribe(self)
            self.landmark = Landmark(
                name=f"joint {entity_a.name} {entity_b.name}", collide=collidable, movable=True, rotatable=True,
                mass=mass, shape=(Box(length=dist, width=width) if width != 0 else Line(length=dist)), is_joint=True,
            )
            self.joint_constraints += [
                JointConstraint(self.landmark, entity_a, (-1, 0), anchor_a, 0.0, rotate_a, fixed_rotation_a),
                JointConstraint(self.landmark, entity_b, (1, 0), anchor_b, 0.0, rotate_b, fixed_rotation_b),
            ]

    def notify(self, observable, *args, **kwargs):  # joints.py:121-144
        pos_a = self.joint_constraints[0].pos_point(self.entity_a)
        pos_b = self.joint_constraints[1].pos_point(self.entity_b)
        self.landmark.set_pos((pos_a + pos_b) / 2, batch_index=None)
        angle = torch.atan2(pos_b[:, Y] - pos_a[:, Y], pos_b[:, X] - pos_a[:, X]).unsqueeze(-1)
        self.landmark.set_rot(angle, batch_index=None)
        if not self.rotate_a and self.fixed_rotation_a is None:
            self.joint_constraints[0].fixed_rotation = angle - self.entity_a.state.rot
        if not self.rotate_b and self.fixed_rotation_b is None:
            self.joint_constraints[1].fixed_rotation = angle - self.entity_b.state.rot


# ----------------------------------------------------------------------------------
# world
# ----------------------------------------------------------------------------------
#: batches below this many environments use the reference's exact batch-global broad phase by default.  Since round 6 that is
#: EVERY batch: the lazy form of the rule (csrc/vmas_env_device.h) runs inside the step launch at any size and costs a tile
#: a wait only when one of its environments sits in a pair's band, so there is no size from which the per-environment form
#: has to stand in for the reference's (it did not: VERDICT round 5 - football 8 192: 20 of 30 states differ).
EXACT_AUTO_BELOW = 1 << 62

class World:
    """core.py:1088-1232 + step 1972-2015, re-homed on one packed buffer and one kernel."""

    def __init__(
        self,
        batch_dim: int,
        device: Union[torch.device, str],
        dt: float = 0.1,
        substeps: int = 1,
        drag: float = DRAG,
        linear_friction: float = LINEAR_FRICTION,
        angular_friction: float = ANGULAR_FRICTION,
        x_semidim: float = None,
        y_semidim: float = None,
        dim_c: int = 0,
        collision_force: float = COLLISION_FORCE,
        joint_force: float = JOINT_FORCE,
        torque_constraint_force: float = TORQUE_CONSTRAINT_FORCE,
        contact_margin: float = 1e-3,
        gravity: Tuple[float, float] = (0.0, 0.0),
        exact_broad_phase: Optional[bool] = None,
        lanes_per_env: int = 0,
    ):
        assert batch_dim > 0, f"Batch dim must be greater than 0, got {batch_dim}"
        self._batch_dim = batch_dim
        self._device = torch.device(device)
        self._agents: List[Agent] = []
        self._landmarks: List[Landmark] = []
        self._x_semidim, self._y_semidim = x_semidim, y_semidim
        self._dim_p, self._dim_c = 2, dim_c
        self._dt, self._substeps = dt, substeps
        self._sub_dt = self._dt / self._substeps
        self._drag = drag
        self._gravity = torch.tensor(gravity, dtype=torch.float32)
        self._linear_friction, self._angular_friction = linear_friction, angular_friction
        self._collision_force, self._joint_force = collision_force, joint_force
        self._contact_margin, self._torque_constraint_force = contact_margin, torque_constraint_force
        self._joints: Dict[frozenset, JointConstraint] = {}
        # packed storage + backend (built lazily once the entity list is known)
        self._ld = A.leading_dim(batch_dim)
        self._state: Optional[Tensor] = None
        self._agent_ft: Optional[Tensor] = None
        self._backend = None
        self._spec: Optional[WorldSpec] = None
        #: True (None: the default) = the reference's batch-global ``.any()`` broad phase exactly (core.py:2797-2801; inside
        #: the step launch at every batch size: the lazy form); False = every static pair evaluated per
        #: environment - the same result only while SOME environment of the batch has the pair's bounding circles
        #: overlapping whenever another one sits in its band.
        self.exact_broad_phase = (batch_dim < EXACT_AUTO_BELOW) if exact_broad_phase is None else bool(exact_broad_phase)
        self._lanes_per_env = lanes_per_env
        # geometric queries answered by ONE kernel launch per state version (GPU worlds)
        self._query_list: List[Tuple[str, int, int]] = []
        self._query_index: Dict[Tuple[str, int, int], int] = {}
        self._query_cache: Optional[Tensor] = None

    # ---- construction ---------------------------------------------------------
    def _register(self, e: Entity):
        assert self._state is None, (
            "entities cannot be added once the packed state exists (after the first state access/reset/step)"
        )
        e._world = self

    def add_agent(self, agent: Agent):
        self._register(agent)
        self._agents.append(agent)
        self._reindex()

    def add_landmark(self, landmark: Landmark):
        self._register(landmark)
        self._landmarks.append(landmark)
        self._reindex()

    def add_joint(self, joint: Joint):
        assert self._substeps > 1, "For joints, world substeps needs to be more than 1"
        if joint.landmark is not None:
            self.add_landmark(joint.landmark)
        for c in joint.joint_constraints:
            self._joints[frozenset({c.entity_a.name, c.entity_b.name})] = c
        self._invalidate_backend()

    def _reindex(self):
        for i, e in enumerate(self.entities):  # landmarks first, then agents (core.py:1220-1222)
            e._index = i
        for i, a in enumerate(self._agents):
            a._agent_index = i

    # ---- packed storage -------------------------------------------------------------
    def _packed_state(self) -> Tensor:
        if self._state is None:
            self._state = torch.zeros(len(self.entities), A.STATE_FIELDS, self._ld, dtype=torch.float32, device=self._device)
            self._agent_ft = torch.zeros(max(len(self._agents), 1), A.AGENT_FIELDS, self._ld, dtype=torch.float32,
                                         device=self._device)
            if self._dim_c > 0:
                for a in self._agents:
                    a.state.c = torch.zeros(self._batch_dim, self._dim_c, dtype=torch.float32, device=self._device)
        return self._state

    def _packed_agent_ft(self) -> Tensor:
        self._packed_state()
        return self._agent_ft

    def _invalidate_backend(self):
        if self._backend is not None:
            self._backend.close()
        self._backend, self._spec = None, None
        self._query_cache = None
        self._queries_registered = False

    def invalidate_queries(self):
        """Call after writing entity state in place through a view (``e.state.pos[i] = ...``); the
        setters, ``reset`` and ``step`` do it themselves."""
        self._query_cache = None

    def _query(self, kind: str, a: Entity, b: Entity) -> Optional[Tensor]:
        """Row of the fused query kernel for (kind, a, b), or None on non-GPU worlds."""
        if self._device.type != "cuda":
            return None
        key = (kind, a._index, b._index)
        be = self._get_backend()
        if key not in self._query_index or not getattr(self, "_queries_registered", False):
            if key not in self._query_index:
                self._query_index[key] = len(self._query_list)
                self._query_list.append(key)
            be.set_queries(self._query_list)
            self._queries_registered = True
            self._query_cache = None
        if self._query_cache is None:
            self._query_cache = be.run_queries()
        return self._query_cache[self._query_index[key], : self._batch_dim]

    @property
    def spec(self) -> WorldSpec:
        if self._spec is None:
            self._spec = spec_from_world(self)
        return self._spec

    def _get_backend(self):
        if self._backend is None:
            from .backend import HipWorld  # raises loudly when the HIP library is missing

            self._packed_state()
            self._backend = HipWorld(self.spec, self._batch_dim, self._device, lanes_per_env=self._lanes_per_env,
                                     state=self._state, agent_ft=self._agent_ft)
            hint = getattr(self, "epilogue_hint", None)  # (post_kind, n_packages), set by scenarios with a fused post-step
            if hint is not None and not self._lanes_per_env:
                self._backend.reserve_epilogue(*hint)
            req = getattr(self, "_specialize_request", None)
            if req is not None:
                # a static change (mass setter, add_joint, sensors ...) rebuilt the backend: the specialisation asked for
                # belongs to the world, not to one backend object - re-applied to the new schedule here (from the cache, or
                # compiled if the request allowed compiling); `backend.specialized` tells whether it took
                self._apply_specialize(*req)
        return self._backend

    def _apply_specialize(self, cache_dir, cached_only: bool, strict: bool) -> bool:
        import warnings

        from .specialize import SpecializeError

        be = self._backend
        hint = getattr(self, "epilogue_hint", None)  # (epilogues the planner reserves LDS for; else the scenario's fused_post)
        try:
            return be.specialize(post=hint[0] if hint is not None else getattr(self, "fused_post", 0), cache_dir=cache_dir,
                                 cached_only=cached_only)
        except (SpecializeError, OSError) as e:
            if strict:
                raise
            # the specialisation is an optional fast path: a cache entry the library refuses, an unreadable header, a failing
            # compiler ... must not break make_env - the world keeps the interpreter (same results bit for bit)
            warnings.warn(f"world-specialised kernel not used, the world runs the schedule interpreter: {e}", RuntimeWarning)
            return False

    def specialize(self, cache_dir: Optional[str] = None, cached_only: bool = False, strict: Optional[bool] = None) -> bool:
        """A world-specialised step kernel for this world as it is now (specialize.py): compiled once, cached on disk.  The
        request stays with the world: a backend rebuilt after a static change is specialised again.  ``strict`` (default:
        whenever compiling is allowed): errors raise; otherwise they warn and the world keeps the interpreter."""
        strict = (not cached_only) if strict is None else strict
        self._specialize_request = None  # (applied explicitly below; _get_backend must not apply it a second time)
        self._get_backend()
        self._specialize_request = (cache_dir, cached_only, strict)
        return self._apply_specialize(cache_dir, cached_only, strict)

    # ---- reference API ---------------------------------------------------------------
    batch_dim = property(lambda s: s._batch_dim)
    device = property(lambda s: s._device)
    agents = property(lambda s: s._agents)
    landmarks = property(lambda s: s._landmarks)
    x_semidim = property(lambda s: s._x_semidim)
    y_semidim = property(lambda s: s._y_semidim)
    dt = property(lambda s: s._dt)
    dim_p = property(lambda s: s._dim_p)
    dim_c = property(lambda s: s._dim_c)
    substeps = property(lambda s: s._substeps)

    @property
    def joints(self):
        return self._joints.values()

    @property
    def entities(self) -> List[Entity]:
        return self._landmarks + self._agents

    @property
    def policy_agents(self) -> List[Agent]:
        return [a for a in self._agents if a.action_script is None]

    @property
    def scripted_agents(self) -> List[Agent]:
        return [a for a in self._agents if a.action_script is not None]

    def reset(self, env_index: Optional[int]):
        self._query_cache = None
        self._packed_state()
        for e in self.entities:
            e._reset(env_index)

    def _per_env_inputs(self):
        """Optional per-env step inputs: inferred joint rotations, tensor gravities."""
        spec = self.spec
        jfr = eg = None
        be = self._get_backend()
        if any(j.per_env_fixed_rotation for j in spec.joints):
            jfr = torch.zeros(len(spec.joints), self._ld, dtype=torch.float32, device=self._device)
            k = 0
            ents = self.entities
            for ia, ea in enumerate(ents):
                for ib in range(ia + 1, len(ents)):
                    c = self._joints.get(frozenset({ea.name, ents[ib].name}))
                    if c is None:
                        continue
                    fr = c.fixed_rotation
                    if isinstance(fr, Tensor):
                        jfr[k, : self._batch_dim] = fr.reshape(-1)
                    elif fr is not None:
                        jfr[k, : self._batch_dim] = float(fr)
                    k += 1
        if any(e.per_env_gravity for e in spec.entities):
            eg = torch.zeros(len(spec.entities), 2, self._ld, dtype=torch.float32, device=self._device)
            for i, e in enumerate(self.entities):
                if spec.entities[i].per_env_gravity:
                    eg[i, :, : self._batch_dim] = e.gravity.T
        del be
        return jfr, eg

    def step(self):
        """core.py:1972-2015 as ONE fused kernel launch (or 2 per substep in exact mode)."""
        be = self._get_backend()
        self._query_cache = None
        jfr, eg = self._per_env_inputs()
        if self.exact_broad_phase:
            be.step_exact(joint_fixed_rot=jfr, entity_gravity=eg)
        else:
            be.step(joint_fixed_rot=jfr, entity_gravity=eg)
        if self._dim_c > 0:  # _update_comm_state core.py:2910-2913
            for agent in self._agents:
                if not agent.silent:
                    agent.state.c = agent.action.c

    def step_env(self, ingest_args, err_flags, post_kind: int, post_desc, post_buffers):
        """``step()`` with the environment's action ingest and the scenario's post-step fused into the
        same launch (fused.py / ``vmas_world_step_env``)."""
        assert self._dim_c == 0
        be = self._get_backend()
        self._query_cache = None
        jfr, eg = self._per_env_inputs()
        be.step_env(ingest_args, err_flags, post_kind, post_desc, post_buffers, joint_fixed_rot=jfr, entity_gravity=eg,
                    exact=self.exact_broad_phase)

    # ---- scenario-side geometric queries (core.py:1788-1969, 2788-2803): batched torch ops
    def collides(self, a: Entity, b: Entity) -> Tensor:
        """Per-environment version of the broad-phase test of core.py:2788-2803 ([B] bool; the
        reference reduces it with ``.any()`` over the batch, a host sync)."""
        if (not a.collides(b)) or (not b.collides(a)) or a is b:
            return torch.zeros(self._batch_dim, dtype=torch.bool, device=self._device)
        if not a.movable and not a.rotatable and not b.movable and not b.rotatable:
            return torch.zeros(self._batch_dim, dtype=torch.bool, device=self._device)
        d = torch.linalg.vector_norm(a.state.pos - b.state.pos, dim=-1)
        return d <= a.shape.circumscribed_radius() + b.shape.circumscribed_radius()

    def get_distance_from_point(self, entity: Entity, test_point_pos: Tensor, env_index: int = None) -> Tensor:
        from . import geometry

        d = geometry.get_distance_from_point(entity, test_point_pos)
        return d if env_index is None else d[env_index]

    def get_distance(self, entity_a: Entity, entity_b: Entity, env_index: int = None) -> Tensor:
        from . import geometry

        d = self._query("distance", entity_a, entity_b)
        if d is None:
            d = geometry.get_distance(entity_a, entity_b)
        else:
            d = d.clone()  # (the query kernel's output buffer is re-used by the next launch: the reference returns fresh tensors)
        return d if env_index is None else d[env_index]

    def is_overlapping(self, entity_a: Entity, entity_b: Entity, env_index: int = None) -> Tensor:
        from . import geometry

        q = self._query("overlap", entity_a, entity_b)
        o = geometry.is_overlapping(entity_a, entity_b) if q is None else q > 0.5
        return o if env_index is None else o[env_index]

    def cast_rays_all(self) -> Tensor:
        """All Lidar sensors of all agents at once: [n_lidars, max_rays, ld]."""
        return self._get_backend().cast_rays()
