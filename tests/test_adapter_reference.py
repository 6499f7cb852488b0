"""attach() on the REFERENCE's own Environment (the reference: /root/reference, or oracle/_ref on the GPU box).

No GPU here, so the backend injected into the adapter is the CPU oracle (test
infrastructure): what is under test is the drop-in plumbing - packing, the write-through
state setters, reset/reset_at, the replaced ``world.step`` and Lidar ``measure`` - by running
two reference environments side by side, one untouched and one attached."""
import os
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.reference

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


@pytest.fixture(scope="module")
def vmas():
    from oracle import ref  # /root/reference here, its byte-compiled build under oracle/_ref on the GPU box

    ref.import_vmas()
    return ref  # ref.make_env == vmas.make_env with scenario names resolved by import


from ref_backend import OracleBackend  # noqa: E402


def _actions(env, g):
    return [(torch.rand(env.num_envs, a.action_size, generator=g) * 2 - 1) * a.action.u_range_tensor.cpu() for a in env.agents]


CASES = [("balance", dict(n_agents=3), 60), ("transport", {}, 40), ("navigation", dict(n_agents=4), 30),
         ("waterfall", {}, 20), ("football", dict(n_blue_agents=2, n_red_agents=2, ai_red_agents=False), 30)]


def _side_by_side(vmas, scenario, kw, steps, device, attach_kw):
    """Two reference environments from the same seed: one untouched on the CPU, one attached (on ``device``); same
    actions; states, rewards and dones must track each other, through a partial reset and a detach."""
    from vectorizedmultiagentsimulator_amd.adapter import attach

    B = 6
    ref = vmas.make_env(scenario, num_envs=B, device="cpu", seed=0, **kw)
    att = vmas.make_env(scenario, num_envs=B, device=device, seed=0, **kw)
    if device != "cpu":  # the two devices draw different reset states: start the attached one from the CPU one's
        for ea, eb in zip(ref.world.entities, att.world.entities):
            eb.set_pos(ea.state.pos.to(device), batch_index=None)
            eb.set_vel(ea.state.vel.to(device), batch_index=None)
            eb.set_rot(ea.state.rot.to(device), batch_index=None)
            eb.set_ang_vel(ea.state.ang_vel.to(device), batch_index=None)
    h = attach(att, exact_broad_phase=True, **attach_kw)
    g1, g2 = torch.Generator().manual_seed(7), torch.Generator().manual_seed(7)
    for t in range(steps):
        o1, r1, d1, _ = ref.step(_actions(ref, g1))
        o2, r2, d2, _ = att.step([a.to(device) for a in _actions(att, g2)])
        # free-running comparison: chaos amplifies last-bit libm differences (SURVEY.md App. C-2:
        # the reference against itself reaches 2e-4 after 50 balance steps), so this is a
        # plumbing check with a loose bound; per-step parity is pinned by the golden tests
        for ea, eb in zip(ref.world.entities, att.world.entities):
            for name in ("pos", "vel", "rot", "ang_vel"):
                a, b = getattr(ea.state, name), getattr(eb.state, name).cpu()
                assert torch.allclose(a, b, atol=5e-3, rtol=1e-2), (
                    f"{scenario} {ea.name}.{name} diverged at step {t}: {(a - b).abs().max()}")
        keep = torch.ones(B, dtype=torch.bool)
        keep[2] = t <= steps // 2  # scenario-side caches of the re-drawn env differ after the reset
        if device == "cpu":  # (scenario-side shaping caches were initialised from different reset states on a GPU)
            for a, b in zip(r1, r2):
                assert torch.allclose(a[keep], b.cpu()[keep], atol=0.5, rtol=1e-2), f"{scenario} reward diverged at step {t}"
            assert (d1 == d2.cpu())[keep].float().mean() > 0.8
        if t == steps // 2:  # partial reset goes through the write-through setters
            ref.reset_at(2)
            att.reset_at(2)
            # both environments draw from ONE class-level RNG stream (environment.py:59-63), so
            # the two resets differ: copy env 2 across through the per-index setters
            for ea, eb in zip(ref.world.entities, att.world.entities):
                eb.set_pos(ea.state.pos[2].to(device), batch_index=2)
                eb.set_vel(ea.state.vel[2].to(device), batch_index=2)
                eb.set_rot(ea.state.rot[2].to(device), batch_index=2)
                eb.set_ang_vel(ea.state.ang_vel[2].to(device), batch_index=2)
    # state objects are views of the packed buffer
    e0 = att.world.entities[-1]
    assert e0.state.pos.data_ptr() == h.state[len(att.world.entities) - 1, 0:2, :B].T.data_ptr()
    h.detach()
    att.step([a.to(device) for a in _actions(att, g2)])  # reference path works again after detach
    return h


@pytest.mark.parametrize("scenario,kw,steps", CASES)
def test_attached_reference_env_tracks_the_untouched_one(vmas, scenario, kw, steps):
    _side_by_side(vmas, scenario, kw, steps, "cpu", dict(backend_factory=OracleBackend))


@pytest.mark.gpu
@pytest.mark.parametrize("scenario,kw,steps", CASES)
def test_attached_reference_env_on_the_hip_step(vmas, scenario, kw, steps):
    """The drop-in as shipped: the reference's own Environment on cuda:0, ``attach()`` with the default backend
    (``HipWorld`` = libvmas_hip.so) - World.step and Lidar.measure run on the kernels - beside an untouched reference
    environment on the CPU."""
    from vectorizedmultiagentsimulator_amd.backend import HipWorld

    assert torch.cuda.is_available()
    h = _side_by_side(vmas, scenario, kw, steps, "cuda:0", {})
    assert isinstance(h.backend, HipWorld)


def test_attach_refuses_grad(vmas):
    from vectorizedmultiagentsimulator_amd.adapter import attach

    env = vmas.make_env("balance", num_envs=2, device="cpu", seed=0, grad_enabled=True)
    with pytest.raises(NotImplementedError):
        attach(env, backend_factory=OracleBackend)


# ---- mutable static inputs: the step must use what the world holds NOW ---------------------------------
def _copy_state(src, dst, device):
    for ea, eb in zip(src.world.entities, dst.world.entities):
        eb.set_pos(ea.state.pos.to(device), batch_index=None)
        eb.set_vel(ea.state.vel.to(device), batch_index=None)
        eb.set_rot(ea.state.rot.to(device), batch_index=None)
        eb.set_ang_vel(ea.state.ang_vel.to(device), batch_index=None)


def _track(ref, att, device, steps, g1, g2, tol=1e-4):
    for t in range(steps):
        ref.step(_actions(ref, g1))
        att.step([a.to(device) for a in _actions(att, g2)])
        for ea, eb in zip(ref.world.entities, att.world.entities):
            for name in ("pos", "vel"):
                a, b = getattr(ea.state, name), getattr(eb.state, name).cpu()
                assert torch.allclose(a, b, atol=tol, rtol=1e-4), f"{ea.name}.{name} step {t}: {(a - b).abs().max()}"


def _het_mass_reset_is_tracked(vmas, device, attach_kw):
    """debug/het_mass.py:50-53 redraws both agents' masses in reset_world_at (setter core.py:634-636): the attached step
    must integrate with the NEW masses (round 2 stepped with the ones extracted at attach time)."""
    from vectorizedmultiagentsimulator_amd.adapter import attach

    ref = vmas.make_env("het_mass", num_envs=5, device="cpu", seed=0)
    att = vmas.make_env("het_mass", num_envs=5, device=device, seed=0)
    h = attach(att, exact_broad_phase=True, **attach_kw)
    g1, g2 = torch.Generator().manual_seed(3), torch.Generator().manual_seed(3)
    for round_ in range(3):
        ref.reset()
        att.reset()  # the scenario's reset_world_at redraws the attached world's masses
        _copy_state(ref, att, device)
        masses = [a.mass for a in att.world.agents]
        assert round_ == 0 or masses != prev
        prev = masses
        for a, m in zip(ref.world.agents, masses):  # (the two resets drew different noise: the untouched reference
            a.mass = m                              #  environment is given the attached one's masses, not vice versa)
        _track(ref, att, device, 8, g1, g2)
        assert [h.spec.entities[i].mass for i in range(len(masses))] == [float(m) for m in masses]
    assert h.refreshes >= 2  # every reset changed the masses
    h.detach()


def _filter_and_setters_are_tracked(vmas, device, attach_kw):
    """collision_filter reassigned after construction (joint_passage.py:622, setter core.py:720-722), mass and
    linear_friction through their public setters (core.py:634-636, 696-701) - mid-episode, no reset."""
    from vectorizedmultiagentsimulator_amd.adapter import attach

    kw = dict(n_agents=3)
    ref = vmas.make_env("balance", num_envs=4, device="cpu", seed=0, **kw)
    att = vmas.make_env("balance", num_envs=4, device=device, seed=0, **kw)
    _copy_state(ref, att, device)
    h = attach(att, exact_broad_phase=True, **attach_kw)
    g1, g2 = torch.Generator().manual_seed(5), torch.Generator().manual_seed(5)
    _track(ref, att, device, 5, g1, g2)
    n0 = len(h.spec.pairs)
    for env in (ref, att):  # the package stops colliding with the agents: it falls through them
        pkg = env.scenario.package
        agents = list(env.world.agents)
        pkg.collision_filter = lambda e, _agents=agents: e not in _agents
        env.world.agents[0].mass = 2.5
        env.scenario.line.linear_friction = 0.05
    _track(ref, att, device, 8, g1, g2)
    assert h.refreshes == 1 and len(h.spec.pairs) < n0
    assert h.spec.entities[att.world.entities.index(att.world.agents[0])].mass == 2.5
    h.detach()


def test_attach_tracks_het_mass_resets(vmas):
    _het_mass_reset_is_tracked(vmas, "cpu", dict(backend_factory=OracleBackend))


def test_attach_tracks_filter_mass_friction_setters(vmas):
    _filter_and_setters_are_tracked(vmas, "cpu", dict(backend_factory=OracleBackend))


@pytest.mark.gpu
def test_attach_tracks_het_mass_resets_on_the_hip_step(vmas):
    _het_mass_reset_is_tracked(vmas, "cuda:0", {})


@pytest.mark.gpu
def test_attach_tracks_filter_mass_friction_setters_on_the_hip_step(vmas):
    _filter_and_setters_are_tracked(vmas, "cuda:0", {})
