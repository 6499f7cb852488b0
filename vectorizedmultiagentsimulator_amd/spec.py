"""Static world description ("WorldSpec") - the constant block the kernels run on.

``spec_from_world`` walks a live ``World`` object by duck typing (attribute names of
vmas/simulator/core.py; works on the reference's classes and on this package's own
``core`` classes alike) and evaluates, once, everything in ``World.step`` that does
not depend on the per-environment state:

* entity order and flags                         core.py:1220-1222, 1995-2004
* the static part of ``World.collides``          core.py:2788-2796
* pair bucketing by shape and its ordering       core.py:2112-2189
* the joint list in discovery order              core.py:2116-2122
* python-double -> fp32 roundings of all scalars exactly where torch performs them

The spec is plain data (JSON round-trippable) so golden fixtures can carry it to a
machine that does not have the reference installed.
"""
from __future__ import annotations

import ctypes as C
import json
import math
from dataclasses import asdict, dataclass, field
from typing import Any, Dict, List, Optional

from . import _abi as A

_SHAPE_CODES = {"Sphere": A.SHAPE_SPHERE, "Box": A.SHAPE_BOX, "Line": A.SHAPE_LINE}


@dataclass
class EntitySpec:
    name: str
    flags: int
    shape: int
    agent_index: int
    mass: float
    inertia: float
    length: float = 0.0
    width: float = 0.0
    radius: float = 0.0
    bound_radius: float = 0.0
    one_minus_drag: float = 1.0
    max_speed: float = 0.0
    v_range: float = 0.0
    lin_friction: float = 0.0
    ang_friction: float = 0.0
    gravity: List[float] = field(default_factory=lambda: [0.0, 0.0])
    max_f: float = 0.0
    f_range: float = 0.0
    max_t: float = 0.0
    t_range: float = 0.0
    per_env_gravity: bool = False  # entity.gravity is a [B,2] tensor -> StepArgs.entity_gravity


@dataclass
class PairSpec:
    a: int
    b: int
    type: int
    bound_sum: float


@dataclass
class JointSpec:
    a: int
    b: int
    delta_a: List[float]
    delta_b: List[float]
    dist: float
    rotate: bool
    fixed_rotation: float = 0.0
    per_env_fixed_rotation: bool = False  # fixed_rotation is a [B,1] tensor (joints.py:141-144)


@dataclass
class LidarSpec:
    entity: int
    n_rays: int
    max_range: float
    targets: List[int]
    angles: List[float]


@dataclass
class WorldSpec:
    entities: List[EntitySpec]
    pairs: List[PairSpec]
    joints: List[JointSpec]
    n_agents: int
    substeps: int
    sub_dt: float
    gravity: List[float]
    has_gravity: bool
    x_semidim: Optional[float]
    y_semidim: Optional[float]
    collision_force: float
    joint_force: float
    contact_margin: float
    torque_constraint_force: float
    lidars: List[LidarSpec] = field(default_factory=list)

    # ---- sizes -----------------------------------------------------------
    @property
    def n_entities(self) -> int:
        return len(self.entities)

    @property
    def n_dynamic(self) -> int:
        return sum(1 for e in self.entities if e.flags & (A.F_MOVABLE | A.F_ROTATABLE))

    def step_bytes_per_env(self) -> int:
        """Algorithmic HBM bytes per env per World.step:
        read all entity state + agent force/torque, write back dynamic entities."""
        return 24 * self.n_entities + 12 * self.n_agents + 24 * self.n_dynamic

    # ---- (de)serialisation -------------------------------------------------
    def to_json(self) -> str:
        return json.dumps(asdict(self))

    @staticmethod
    def from_json(s: str) -> "WorldSpec":
        d = json.loads(s)
        return WorldSpec(
            entities=[EntitySpec(**e) for e in d.pop("entities")],
            pairs=[PairSpec(**p) for p in d.pop("pairs")],
            joints=[JointSpec(**j) for j in d.pop("joints")],
            lidars=[LidarSpec(**l) for l in d.pop("lidars", [])],
            **d,
        )

    # ---- ctypes ------------------------------------------------------------
    def to_ctypes(self) -> "CWorldDesc":
        return CWorldDesc(self)


class CWorldDesc:
    """Owns the ctypes arrays a ``VmasWorldDesc`` points into (keep it alive)."""

    def __init__(self, spec: WorldSpec):
        self.spec = spec
        nE, nP, nJ = len(spec.entities), len(spec.pairs), len(spec.joints)
        self.entities = (A.EntityDesc * max(nE, 1))()
        for i, e in enumerate(spec.entities):
            d = self.entities[i]
            d.flags, d.shape, d.agent_index = e.flags, e.shape, e.agent_index
            d.mass, d.inertia = e.mass, e.inertia
            d.length, d.width, d.radius, d.bound_radius = e.length, e.width, e.radius, e.bound_radius
            d.one_minus_drag = e.one_minus_drag
            d.max_speed, d.v_range = e.max_speed, e.v_range
            d.lin_friction, d.ang_friction = e.lin_friction, e.ang_friction
            d.gravity[0], d.gravity[1] = e.gravity
            d.max_f, d.f_range, d.max_t, d.t_range = e.max_f, e.f_range, e.max_t, e.t_range
        self.pairs = (A.PairDesc * max(nP, 1))()
        for i, p in enumerate(spec.pairs):
            self.pairs[i].a, self.pairs[i].b, self.pairs[i].type = p.a, p.b, p.type
            self.pairs[i].bound_sum = p.bound_sum
        self.joints = (A.JointDesc * max(nJ, 1))()
        for i, j in enumerate(spec.joints):
            d = self.joints[i]
            d.a, d.b = j.a, j.b
            d.delta_a[0], d.delta_a[1] = j.delta_a
            d.delta_b[0], d.delta_b[1] = j.delta_b
            d.dist, d.rotate, d.fixed_rotation = j.dist, int(bool(j.rotate)), j.fixed_rotation
        w = A.WorldDesc()
        w.abi_version = A.ABI_VERSION
        w.n_entities, w.n_agents, w.n_pairs, w.n_joints = nE, spec.n_agents, nP, nJ
        w.substeps, w.sub_dt = spec.substeps, spec.sub_dt
        w.gravity[0], w.gravity[1] = spec.gravity
        w.has_gravity = int(bool(spec.has_gravity))
        w.x_semidim = float("nan") if spec.x_semidim is None else spec.x_semidim
        w.y_semidim = float("nan") if spec.y_semidim is None else spec.y_semidim
        w.collision_force, w.joint_force = spec.collision_force, spec.joint_force
        w.contact_margin, w.torque_constraint_force = spec.contact_margin, spec.torque_constraint_force
        w.entities = C.cast(self.entities, C.POINTER(A.EntityDesc))
        w.pairs = C.cast(self.pairs, C.POINTER(A.PairDesc))
        w.joints = C.cast(self.joints, C.POINTER(A.JointDesc))
        self.world = w
        # lidars
        self._lidar_keep: List[Any] = []
        self.lidars = (A.LidarDesc * max(len(spec.lidars), 1))()
        for i, l in enumerate(spec.lidars):
            t = (C.c_int32 * max(len(l.targets), 1))(*l.targets)
            a = (C.c_float * max(len(l.angles), 1))(*l.angles)
            self._lidar_keep += [t, a]
            d = self.lidars[i]
            d.entity, d.n_rays, d.max_range, d.n_targets = l.entity, l.n_rays, l.max_range, len(l.targets)
            d.targets = C.cast(t, C.POINTER(C.c_int32))
            d.angles = C.cast(a, C.POINTER(C.c_float))

    @property
    def max_rays(self) -> int:
        return max([l.n_rays for l in self.spec.lidars], default=0)


def _is_tensor(x) -> bool:
    return hasattr(x, "shape") and hasattr(x, "dtype") and not isinstance(x, (int, float))


def _pair_type(sa: str, sb: str):
    """(type, swap) following the bucketing of core.py:2125-2174."""
    if sa == "Sphere" and sb == "Sphere":
        return A.PAIR_SS, False
    if {sa, sb} == {"Line", "Sphere"}:
        return A.PAIR_LS, sa == "Sphere"  # (line, sphere)
    if sa == "Line" and sb == "Line":
        return A.PAIR_LL, False
    if {sa, sb} == {"Box", "Sphere"}:
        return A.PAIR_BS, sa == "Sphere"  # (box, sphere)
    if {sa, sb} == {"Box", "Line"}:
        return A.PAIR_BL, sa == "Line"  # (box, line)
    if sa == "Box" and sb == "Box":
        return A.PAIR_BB, False
    raise AssertionError(f"unsupported shape pair {sa}/{sb}")  # core.py:2174


def spec_from_world(world, with_lidars: bool = True) -> WorldSpec:
    """Extract the static description of ``world`` (duck typed, see module doc)."""
    entities = list(world.entities)
    agents = list(world.agents)
    agent_ids = {id(a): i for i, a in enumerate(agents)}
    index = {id(e): i for i, e in enumerate(entities)}
    world_drag = float(world._drag)
    world_lin = float(world._linear_friction)
    world_ang = float(world._angular_friction)

    especs: List[EntitySpec] = []
    for e in entities:
        sname = type(e.shape).__name__
        if sname not in _SHAPE_CODES:
            raise NotImplementedError(f"shape {sname} of entity {e.name}")
        flags = 0
        if e.movable:
            flags |= A.F_MOVABLE
        if e.rotatable:
            flags |= A.F_ROTATABLE
        is_agent = id(e) in agent_ids
        if is_agent:
            flags |= A.F_AGENT
        es = EntitySpec(
            name=e.name,
            flags=0,
            shape=_SHAPE_CODES[sname],
            agent_index=agent_ids.get(id(e), -1),
            mass=float(e.mass),
            inertia=float(e.moment_of_inertia),
            bound_radius=float(e.shape.circumscribed_radius()),
        )
        if sname == "Sphere":
            es.radius = float(e.shape.radius)
        elif sname == "Box":
            es.length, es.width = float(e.shape.length), float(e.shape.width)
            if e.shape.hollow:
                flags |= A.F_HOLLOW
        else:
            es.length = float(e.shape.length)
        drag = e.drag if e.drag is not None else world_drag
        es.one_minus_drag = 1 - float(drag)  # python double, rounded to fp32 by the multiply
        if e.max_speed is not None:
            flags |= A.F_MAX_SPEED
            es.max_speed = float(e.max_speed)
        if e.v_range is not None:
            flags |= A.F_V_RANGE
            es.v_range = float(e.v_range)
        lf = e.linear_friction
        if lf is not None:
            if _is_tensor(lf):
                raise NotImplementedError("tensor-valued linear_friction")
            flags |= A.F_LIN_FRICTION
            es.lin_friction = float(lf)
        elif world_lin > 0:
            flags |= A.F_LIN_FRICTION
            es.lin_friction = world_lin
        af = e.angular_friction
        if af is not None:
            if _is_tensor(af):
                raise NotImplementedError("tensor-valued angular_friction")
            flags |= A.F_ANG_FRICTION
            es.ang_friction = float(af)
        elif world_ang > 0:
            flags |= A.F_ANG_FRICTION
            es.ang_friction = world_ang
        g = e.gravity
        if g is not None:
            flags |= A.F_GRAVITY
            if _is_tensor(g) and g.dim() == 2:
                es.per_env_gravity = True
            else:
                gl = [float(v) for v in (g.tolist() if _is_tensor(g) else g)]
                es.gravity = gl
        if is_agent:
            if e.max_f is not None:
                flags |= A.F_MAX_F
                es.max_f = float(e.max_f)
            if e.f_range is not None:
                flags |= A.F_F_RANGE
                es.f_range = float(e.f_range)
            if e.max_t is not None:
                flags |= A.F_MAX_T
                es.max_t = float(e.max_t)
            if e.t_range is not None:
                flags |= A.F_T_RANGE
                es.t_range = float(e.t_range)
        es.flags = flags
        especs.append(es)

    # pair enumeration core.py:2112-2174 + static part of collides core.py:2788-2796
    buckets: Dict[int, List[PairSpec]] = {t: [] for t in range(6)}
    jspecs: List[JointSpec] = []
    joints_map = getattr(world, "_joints", {})
    for ia, ea in enumerate(entities):
        for ib, eb in enumerate(entities):
            if ib <= ia:
                continue
            joint = joints_map.get(frozenset({ea.name, eb.name}), None)
            if joint is not None:
                ja, jb = index[id(joint.entity_a)], index[id(joint.entity_b)]
                fr = joint.fixed_rotation
                # None = "to be inferred by Joint.notify" (joints.py:141-144) -> per-env tensor
                per_env = fr is None or _is_tensor(fr)
                jspecs.append(
                    JointSpec(
                        a=ja,
                        b=jb,
                        delta_a=[float(v) for v in joint.entity_a.shape.get_delta_from_anchor(joint.anchor_a)],
                        delta_b=[float(v) for v in joint.entity_b.shape.get_delta_from_anchor(joint.anchor_b)],
                        dist=float(joint.dist),
                        rotate=bool(joint.rotate),
                        fixed_rotation=0.0 if per_env else float(fr),
                        per_env_fixed_rotation=per_env,
                    )
                )
                if joint.dist == 0:
                    continue
            if (not ea.collides(eb)) or (not eb.collides(ea)) or ea is eb:
                continue
            if not ea.movable and not ea.rotatable and not eb.movable and not eb.rotatable:
                continue
            t, swap = _pair_type(type(ea.shape).__name__, type(eb.shape).__name__)
            a, b = (ib, ia) if swap else (ia, ib)
            buckets[t].append(
                PairSpec(a=a, b=b, type=t, bound_sum=ea.shape.circumscribed_radius() + eb.shape.circumscribed_radius())
            )
    pairs = [p for t in range(6) for p in buckets[t]]

    grav = world._gravity
    gl = [float(v) for v in (grav.tolist() if _is_tensor(grav) else grav)]
    spec = WorldSpec(
        entities=especs,
        pairs=pairs,
        joints=jspecs,
        n_agents=len(agents),
        substeps=int(world._substeps),
        sub_dt=float(world._sub_dt),
        gravity=gl,
        has_gravity=any(v != 0.0 for v in gl),
        x_semidim=None if world._x_semidim is None else float(world._x_semidim),
        y_semidim=None if world._y_semidim is None else float(world._y_semidim),
        collision_force=float(world._collision_force),
        joint_force=float(world._joint_force),
        contact_margin=float(world._contact_margin),
        torque_constraint_force=float(world._torque_constraint_force),
    )
    if with_lidars:
        spec.lidars = lidars_from_world(world)
    return spec


def lidars_from_world(world) -> List[LidarSpec]:
    """One LidarSpec per (agent, Lidar sensor): sensors.py:47-123, core.py:1676-1691."""
    entities = list(world.entities)
    out: List[LidarSpec] = []
    for agent in world.agents:
        for sensor in getattr(agent, "sensors", []) or []:
            if not hasattr(sensor, "_angles") or not hasattr(sensor, "_max_range"):
                continue
            filt = sensor.entity_filter
            targets = []
            for i, e in enumerate(entities):
                if e is agent or not filt(e):
                    continue
                assert e.collides(agent) and agent.collides(e), "Rays are only casted among collidables"
                targets.append(i)
            ang = sensor._angles
            ang0 = ang[0] if ang.dim() == 2 else ang
            out.append(
                LidarSpec(
                    entity=entities.index(agent),
                    n_rays=int(ang0.shape[0]),
                    max_range=float(sensor._max_range),
                    targets=targets,
                    angles=[float(v) for v in ang0.tolist()],
                )
            )
    return out


def nan_to_none(x: float) -> Optional[float]:
    return None if (isinstance(x, float) and math.isnan(x)) else x
