"""Drop-in for a LIVE reference ``World``: ``attach(env_or_world)`` replaces its
``step`` (vmas/simulator/core.py:1972) - and the ``cast_rays`` made by its Lidar sensors -
with the MI355X-native path, leaving ``vmas.make_env()`` / ``Environment.step()`` / the
Scenario untouched (the seam of SURVEY.md section 8b).

How the state is shared
    The reference keeps every state tensor as a separate ``[B, k]`` attribute and REBINDS it
    on every write (core.py:222-284, 2871-2908).  ``attach`` allocates the packed buffers of
    include/vmas_hip.h, copies the current state in, and then re-homes each
    ``entity.state._pos/_vel/_rot/_ang_vel`` (agents: ``_force/_torque``) as a strided VIEW
    into the packed buffer.  The seven property setters every Python-side write funnels
    through (SURVEY.md 8b "Adapter notes") are patched on the *instances' classes* to
    ``copy_`` into the view instead of rebinding, so scenarios, ``reset_at`` and dynamics keep
    working and the kernel always sees current data - no gather/scatter per step.

What stays dynamic (SURVEY.md 8b "inputs that are mutable between steps")
    ``JointConstraint.fixed_rotation`` tensors and per-env entity gravity are re-read
    every step.  The STATIC description (mass - ``debug/het_mass.py:50-53`` redraws it at every reset -,
    drag / friction / gravity / speed limits, ``collision_filter`` - ``joint_passage.py:614-622`` filters on
    data its reset rewrites -, movable / rotatable / collide flags, shape dimensions) is watched:
    writes to those attributes (setters ``core.py:634-636, 696-706, 720-722`` and the private names behind them) set a
    dirty flag on the handle through a ``__setattr__`` hook on the classes of the attached world, its entities and
    their shapes; a step then costs two integer comparisons unless something was written (round 3 rebuilt a
    fingerprint of ~25 attributes per entity on every step: 26 us for ``balance``, 67 us for ``football`` - several
    times the kernel).  After a ``World.reset`` (``core.py:1234``) the whole spec - collidability matrix included - is
    re-extracted iff the flag is set or some ``collision_filter`` is a closure / callable object, i.e. can depend on
    data the scenario's ``reset_world_at`` rewrites; the native world is rebuilt iff the spec differs.
    Not seen: a closure's captured data changing WITHOUT a ``World.reset`` - call ``handle.refresh()`` after such a write.

``grad_enabled`` worlds are refused (the kernels have no backward), as is a CPU device:
there is no fallback path.
"""
from __future__ import annotations

from typing import Callable, Optional

import torch

from . import _abi as A
from .spec import WorldSpec, spec_from_world


def _default_backend(spec: WorldSpec, batch: int, device, state, agent_ft):
    from .backend import HipWorld

    return HipWorld(spec, batch, device, state=state, agent_ft=agent_ft)


_FIELDS = ("pos", "vel", "rot", "ang_vel", "force", "torque")
_MARK = "_vmas_amd_attached"
_OWNER = "_vmas_amd_owner"
# private names behind everything spec_from_world reads from a world, an entity or a shape (one set for the three kinds)
_WATCH = frozenset((
    "_drag", "_linear_friction", "_angular_friction", "_gravity", "_substeps", "_sub_dt", "_x_semidim", "_y_semidim",
    "_collision_force", "_joint_force", "_contact_margin", "_torque_constraint_force", "_joints",
    "_mass", "_max_speed", "_v_range", "_movable", "_rotatable", "_collide", "_collision_filter", "_shape",
    "_max_f", "_f_range", "_max_t", "_t_range", "_radius", "_length", "_width", "hollow", "_hollow", "_dt",
))


def _patch_setattr(cls) -> None:
    """A write to a watched attribute of an object that belongs to an attached world marks that world's handle dirty.  The
    hook goes on the object's class (once; subclasses of a hooked class inherit it) and does nothing for unattached
    instances beyond one dictionary probe on watched names."""
    cur = cls.__setattr__
    if getattr(cur, "_vmas_amd", False):
        return

    def hook(self, name, value, _orig=cur):
        _orig(self, name, value)
        if name in _WATCH:
            owner = self.__dict__.get(_OWNER)
            if owner is not None:
                owner._dirty = True

    hook._vmas_amd = True
    cls.__setattr__ = hook


def _patch_state_class(cls) -> None:
    """Make the setters of the reference's EntityState/AgentState write THROUGH when the
    instance is attached, and behave exactly as before otherwise.

    The class-level property must be replaced (not a subclass): whole-batch ``set_pos`` calls
    ``EntityState.pos.fset(state, new)`` on the base class explicitly (core.py:745-755)."""
    for klass in cls.__mro__:
        for name in _FIELDS:
            prop = klass.__dict__.get(name)
            if not isinstance(prop, property) or getattr(prop.fset, "_vmas_amd", False):
                continue

            def fset(self, value, _orig=prop.fset, _priv="_" + name):
                if self.__dict__.get(_MARK):
                    cur = self.__dict__[_priv]
                    assert (
                        value.shape[0] == cur.shape[0]
                    ), f"Internal state must match batch dim, got {value.shape[0]}, expected {cur.shape[0]}"
                    cur.copy_(value.to(cur.device).reshape(cur.shape))
                else:
                    _orig(self, value)

            fset._vmas_amd = True
            setattr(klass, name, property(prop.fget, fset, prop.fdel, prop.__doc__))


class AttachedWorld:
    """Handle returned by ``attach``; ``detach()`` restores the reference behaviour."""

    def __init__(self, world, backend_factory: Callable = _default_backend, exact_broad_phase: Optional[bool] = None,
                 specialize: Optional[bool] = False, epilogue=None, spec_post: int = 0):
        from .core import EXACT_AUTO_BELOW

        self.world = world
        self._fast_step = None
        # None: the reference's exact batch-global broad phase below EXACT_AUTO_BELOW environments (core.World)
        self._exact = (int(world.batch_dim) < EXACT_AUTO_BELOW) if exact_broad_phase is None else bool(exact_broad_phase)
        self._factory = backend_factory
        # the fused Environment.step (attached_env.py): the post-step epilogue this world's steps carry - (kind, n_packages)
        # the library sizes its kernel geometry for, and the kind a run-time specialisation is compiled with
        self._epilogue, self._spec_post = epilogue, int(spec_post)
        self.fused = None          # attached_env.FusedEnvStep when env.step itself is the one-launch kernel
        self.fused_reason = None   # ... and why not, when it is not
        self._orig_step = world.step
        self._orig_classes = {}
        self._orig_measures = []
        self.batch = int(world.batch_dim)
        self.ld = A.leading_dim(self.batch)
        self.device = torch.device(world.device)
        self.spec = spec_from_world(world)
        nE, nA = self.spec.n_entities, self.spec.n_agents
        self.state = torch.zeros(nE, A.STATE_FIELDS, self.ld, dtype=torch.float32, device=self.device)
        self.agent_ft = torch.zeros(max(nA, 1), A.AGENT_FIELDS, self.ld, dtype=torch.float32, device=self.device)
        self._rehome()
        # a step kernel compiled for this world (specialize.py), also after a refresh(): True = compile it if the on-disk
        # cache does not have it, None = take it from the cache if it is there (never compile, never fail), False = never
        self.specialize = specialize
        self._new_backend()
        self._check_spec = False  # set by World.reset: the scenario's reset_world_at that follows may change statics
        self.refreshes = 0        # how many times the static description was found changed (tests, diagnostics)
        self._watch()
        world.step = self.step
        self._orig_reset = world.reset

        def reset(env_index=None, _orig=world.reset):
            self._check_spec = True
            return _orig(env_index)

        world.reset = reset
        self._patch_lidars()

    @property
    def exact_broad_phase(self) -> bool:
        return self._exact

    @exact_broad_phase.setter
    def exact_broad_phase(self, value):
        """Takes effect at the next step: the pre-marshalled ``world.step`` stepper is rebuilt for the new mode, and so is
        the one-launch ``env.step`` (its launch form, gating and cached flag all depend on it: attached_env.FusedEnvStep)."""
        self._exact = bool(value)
        if self._fast_step is not None:
            self._fast_step = self.backend.make_stepper(self._exact)
        if self.fused is not None:
            self.fused.build()

    def _new_backend(self):
        self.backend = self._factory(self.spec, self.batch, self.device, self.state, self.agent_ft)
        if self._epilogue is not None and hasattr(self.backend, "reserve_epilogue"):
            self.backend.reserve_epilogue(*self._epilogue)
        self._apply_specialize()

    def _apply_specialize(self):
        if self.specialize is False or not hasattr(self.backend, "specialize"):
            return
        if self.specialize:
            self.backend.specialize(post=self._spec_post)
            return
        try:  # (None: an optional fast path - whatever goes wrong with a cache entry, the world keeps the interpreter)
            self.backend.specialize(post=self._spec_post, cached_only=True)
        except Exception as e:  # noqa: BLE001
            import warnings

            warnings.warn(f"attach(): cached world-specialised kernel not used ({e}); the schedule interpreter runs", RuntimeWarning)

    # ---- state re-homing ---------------------------------------------------------
    def _rehome(self):
        B = self.batch
        for i, e in enumerate(self.world.entities):
            st = e.state
            views = {
                "_pos": self.state[i, 0:2, :B].T,
                "_vel": self.state[i, 2:4, :B].T,
                "_rot": self.state[i, 4:5, :B].T,
                "_ang_vel": self.state[i, 5:6, :B].T,
            }
            for k, v in views.items():
                v.copy_(getattr(st, k))
                st.__dict__[k] = v
            _patch_state_class(type(st))
            st.__dict__[_MARK] = True
            self._orig_classes[id(st)] = (st, type(st))
        for a_i, a in enumerate(self.world.agents):
            st = a.state
            fv = self.agent_ft[a_i, 0:2, :B].T
            tv = self.agent_ft[a_i, 2:3, :B].T
            if st._force is not None:
                fv.copy_(st._force)
            if st._torque is not None:
                tv.copy_(st._torque)
            st.__dict__["_force"], st.__dict__["_torque"] = fv, tv

    # ---- per-step dynamic inputs ------------------------------------------------------
    def _per_env_inputs(self):
        w, spec = self.world, self.spec
        jfr = eg = None
        if any(j.per_env_fixed_rotation for j in spec.joints):
            jfr = torch.zeros(len(spec.joints), self.ld, dtype=torch.float32, device=self.device)
            ents = list(w.entities)
            k = 0
            for ia, ea in enumerate(ents):
                for ib in range(ia + 1, len(ents)):
                    c = w._joints.get(frozenset({ea.name, ents[ib].name}))
                    if c is None:
                        continue
                    fr = c.fixed_rotation
                    if isinstance(fr, torch.Tensor):
                        jfr[k, : self.batch] = fr.reshape(-1)
                    elif fr is not None:
                        jfr[k, : self.batch] = float(fr)
                    k += 1
        if any(e.per_env_gravity for e in spec.entities):
            eg = torch.zeros(len(spec.entities), 2, self.ld, dtype=torch.float32, device=self.device)
            for i, e in enumerate(w.entities):
                if spec.entities[i].per_env_gravity:
                    eg[i, :, : self.batch] = e.gravity.T
        return jfr, eg

    # ---- mutable static inputs ---------------------------------------------------------
    def _watch(self):
        """(Re)arm the change detection on the world as it is now: owner marks on the world, its entities and their shapes
        (their classes hooked once), the tensor-valued attributes identified by object and version, and whether any
        collision filter can depend on data a reset rewrites."""
        w = self.world
        for o in getattr(self, "_marked", ()):  # (a shape object replaced since the last call must not keep marking this handle)
            o.__dict__.pop(_OWNER, None)
        objs = [w] + list(w.entities) + [e.shape for e in w.entities]
        for o in objs:
            _patch_setattr(type(o))
            o.__dict__[_OWNER] = self
        self._marked = objs

        def volatile(f):  # a closure, a bound method or a callable object: may read data reset_world_at rewrites
            return getattr(f, "__closure__", None) is not None or not hasattr(f, "__code__")

        self._volatile_filters = any(volatile(e.collision_filter) for e in w.entities)
        # the step itself: with no per-environment inputs (joint rotations, tensor gravities) it is one pre-marshalled
        # foreign call (backend.HipWorld.make_stepper)
        spec = self.spec
        self._per_env = any(j.per_env_fixed_rotation for j in spec.joints) or any(e.per_env_gravity for e in spec.entities)
        mk = getattr(self.backend, "make_stepper", None)
        self._fast_step = mk(self._exact) if (mk is not None and not self._per_env) else None
        # tensor-valued statics (the world's gravity; an entity's gravity / speed range given as a tensor): an in-place write
        # to one (``e.gravity[1] = ...``) goes through no setter - they are identified by object and version counter
        ts = [getattr(w, "_gravity", None)]
        for e in w.entities:
            ts += [v for v in (e.__dict__.get("_gravity"), e.__dict__.get("_v_range"), e.__dict__.get("_mass")) if isinstance(v, torch.Tensor)]
        self._watched_tensors = [t for t in ts if isinstance(t, torch.Tensor)]
        # (what a step compares - integers and identities, no container is built per step)
        self._seen_versions = [t._version for t in self._watched_tensors]
        self._seen_counts = (len(getattr(w, "_joints", ())), len(getattr(w, "_landmarks", ())) + len(getattr(w, "_agents", ())))
        self._seen_gravity = getattr(w, "_gravity", None)
        self._dirty = False

    def _unchanged(self) -> bool:
        """What a ``__setattr__`` hook cannot see: in-place writes to tensor-valued statics, joints added to the dict,
        entities added to the world."""
        w = self.world
        nj, ne = self._seen_counts
        if (len(getattr(w, "_joints", ())) != nj or len(getattr(w, "_landmarks", ())) + len(getattr(w, "_agents", ())) != ne
                or getattr(w, "_gravity", None) is not self._seen_gravity):
            return False
        for t, v in zip(self._watched_tensors, self._seen_versions):
            if t._version != v:
                return False
        return True

    def _sync_static(self):
        """Rebuild the native world iff the live world's static description is no longer the one it was built from."""
        check, self._check_spec = self._check_spec, False
        if not self._dirty and not (check and self._volatile_filters) and self._unchanged():
            return
        spec = spec_from_world(self.world)
        if spec != self.spec:
            self.refresh(spec)
        else:
            self._watch()  # (a shape object may have been replaced by an equal one: mark the new one)

    # ---- the replaced seam ----------------------------------------------------------
    def step(self):
        """World.step() (core.py:1972-2015) on the native path."""
        w = self.world
        self._sync_static()
        if self._fast_step is not None:
            self._fast_step()
        else:
            jfr, eg = self._per_env_inputs() if self._per_env else (None, None)
            if self._exact:
                self.backend.step_exact(joint_fixed_rot=jfr, entity_gravity=eg)
            else:
                self.backend.step(joint_fixed_rot=jfr, entity_gravity=eg)
        if w._dim_c > 0:  # _update_comm_state core.py:2910-2913
            for agent in w._agents:
                if not agent.silent:
                    agent.state.c = agent.action.c

    def refresh(self, spec: Optional[WorldSpec] = None):
        """Re-extract the static description and rebuild the native world on the same packed buffers (called by ``step``
        itself when masses, filters, ... changed; the entity list must be the one ``attach`` saw)."""
        spec = spec_from_world(self.world) if spec is None else spec
        assert spec.n_entities == self.spec.n_entities and spec.n_agents == self.spec.n_agents, (
            "entities were added to or removed from an attached world: detach() and attach() again")
        self.spec = spec
        self.backend.close()
        self._new_backend()
        self._watch()
        self.refreshes += 1

    # ---- sensors -----------------------------------------------------------------------
    def _patch_lidars(self):
        if not self.spec.lidars:
            return
        k = 0
        for agent in self.world.agents:
            for sensor in agent.sensors:
                if not hasattr(sensor, "_angles"):
                    continue
                idx, n_rays = k, self.spec.lidars[k].n_rays
                orig = sensor.measure
                self._orig_measures.append((sensor, orig))

                def measure(vectorized: bool = True, _idx=idx, _n=n_rays, _sensor=sensor, _orig=orig):
                    if not vectorized:  # the reference's scalar walk (World.cast_ray, a debug mode) stays what it is
                        return _orig(vectorized=False)
                    m = self.backend.cast_rays()[_idx, :_n, : self.batch].T
                    _sensor._last_measurement = m
                    return m

                sensor.measure = measure
                k += 1

    def detach(self):
        if self.fused is not None:
            self.fused.detach()
            self.fused = None
        self.world.step = self._orig_step
        self.world.reset = self._orig_reset
        for st, cls in self._orig_classes.values():
            for k in ("_pos", "_vel", "_rot", "_ang_vel", "_force", "_torque"):
                if k in st.__dict__ and st.__dict__[k] is not None:
                    st.__dict__[k] = st.__dict__[k].clone()
            st.__dict__[_MARK] = False
        for sensor, orig in self._orig_measures:
            sensor.measure = orig
        for o in self._marked:  # (the class hooks stay: they do nothing for objects without an owner)
            o.__dict__.pop(_OWNER, None)
        self.backend.close()


def attach(env_or_world, backend_factory: Callable = _default_backend,
           exact_broad_phase: Optional[bool] = None, specialize: Optional[bool] = False,
           fused: Optional[bool] = None, validate_actions=True) -> AttachedWorld:
    """Put a reference ``Environment`` (or ``World``) on the MI355X-native physics step.  ``exact_broad_phase``: the
    reference's batch-global ``.any()`` broad phase (core.py:2797-2801) exactly - None = True (the lazy form inside the step
    launch, any batch size; round 6); False = every static pair per environment.  ``specialize``: True = compile (once,
    cached on disk) a step kernel for this very world - any scenario the reference ships then runs at the speed of the
    built-in BASELINE specialisations instead of the schedule interpreter's (specialize.py); None = use it if the cache has
    it, never compile; False = the interpreter.

    ``fused``: ``Environment.step`` ITSELF as one kernel launch (attached_env.py) - the reference's action ingest
    (environment.py:616-749) as the step kernel's prologue, the scenario's reward / observation / done / info as its
    epilogue - for the reference's balance, transport, navigation and football scenarios in the configurations the post-step
    kernels cover.  None (default) = wherever that holds on a GPU environment, silently the reference's own
    ``Environment.step`` around the native ``World.step`` otherwise (``handle.fused_reason`` says why); True = required;
    False = never.  ``validate_actions``: the reference's NaN / range asserts (environment.py:621,651-653) - one small
    kernel and ONE host sync in front of the step launch instead of two syncs per agent, same behaviour (a bad action
    raises before the world is touched); "deferred": the step launch itself flags a bad action and the NEXT ``env.step``
    (or ``handle.fused.check_actions()``) raises - no synchronisation at all, the world has stepped once with the bad
    action by then; False drops the check."""
    world = getattr(env_or_world, "world", env_or_world)
    if getattr(env_or_world, "grad_enabled", False):
        raise NotImplementedError("grad_enabled=True needs the reference's autograd path; the HIP step has no backward")
    profile, reason = None, "fused=False"
    if fused is not False:
        from .attached_env import FusedEnvStep, plan_fuse

        default_backend = backend_factory is _default_backend
        profile, reason = plan_fuse(env_or_world) if default_backend else (None, "a custom backend_factory (no HIP library behind it)")
        if profile is None and fused:
            raise NotImplementedError(f"attach(fused=True): the one-launch Environment.step is not available - {reason}")
    if profile is None:
        h = AttachedWorld(world, backend_factory, exact_broad_phase, specialize)
        h.fused_reason = reason
        return h
    h = AttachedWorld(world, backend_factory, exact_broad_phase, specialize, epilogue=profile.reserve(env_or_world),
                      spec_post=profile.post_kind)
    try:
        h.fused = FusedEnvStep(env_or_world, h, profile, validate_actions)
    except Exception as e:  # noqa: BLE001
        # a world layout the post-step class refuses (AssertionError: entity order ...), or anything else the views / the
        # library raise on a reference version this layer was not written against: with fused=None the environment keeps the
        # reference's own Environment.step around the native World.step (the handle is returned, detach() works); fused=True
        # puts everything back and raises
        h.fused = None
        h.fused_reason = f"the post-step kernel does not cover this world: {type(e).__name__}: {e}"
        if fused:
            h.detach()
            if isinstance(e, AssertionError):
                raise NotImplementedError(f"attach(fused=True): {h.fused_reason}") from e
            raise
    return h
