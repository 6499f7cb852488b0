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
